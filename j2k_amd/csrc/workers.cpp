// workers.cpp -- see common.h (class Workers)
#include "common.h"

#include <condition_variable>
#include <exception>
#include <mutex>
#include <thread>

namespace j2k_hip {

struct Workers::Impl {
    std::mutex mu;
    std::condition_variable go, done;
    std::vector<std::thread> threads;
    const std::function<void(unsigned)> *fn = nullptr;
    unsigned count = 0, pending = 0;
    unsigned long long generation = 0;
    bool quit = false;
    std::exception_ptr failure; // first exception thrown by a slice; rethrown by run() on the calling thread

    void loop(unsigned id)
    {
        unsigned long long seen = 0;
        for (;;) {
            const std::function<void(unsigned)> *f;
            {
                std::unique_lock<std::mutex> lk(mu);
                go.wait(lk, [&] { return quit || generation != seen; });
                if (quit) return;
                seen = generation;
                if (id >= count) continue;
                f = fn;
            }
            std::exception_ptr err;
            try { (*f)(id); } catch (...) { err = std::current_exception(); }
            std::lock_guard<std::mutex> lk(mu);
            if (err && !failure) failure = err;
            if (--pending == 0) done.notify_one();
        }
    }
};

Workers::Workers(unsigned n) : impl_(new Impl), n_(n ? n : 1)
{
    for (unsigned i = 1; i < n_; ++i) impl_->threads.emplace_back([this, i] { impl_->loop(i); });
}

Workers::~Workers()
{
    {
        std::lock_guard<std::mutex> lk(impl_->mu);
        impl_->quit = true;
    }
    impl_->go.notify_all();
    for (auto &t : impl_->threads) t.join();
    delete impl_;
}

void Workers::run(unsigned count, const std::function<void(unsigned)> &fn)
{
    if (count > n_) count = n_;
    if (count <= 1) { if (count) fn(0); return; }
    {
        std::lock_guard<std::mutex> lk(impl_->mu);
        impl_->fn = &fn; impl_->count = count; impl_->pending = count - 1;
        ++impl_->generation;
    }
    impl_->go.notify_all();
    std::exception_ptr mine;
    try { fn(0); } catch (...) { mine = std::current_exception(); }
    std::exception_ptr err;
    {
        std::unique_lock<std::mutex> lk(impl_->mu);
        impl_->done.wait(lk, [&] { return impl_->pending == 0; });
        err = mine ? mine : impl_->failure;
        impl_->failure = nullptr;
    }
    if (err) std::rethrow_exception(err);
}

} // namespace j2k_hip

"""j2k_amd -- MI355X-native JPEG 2000 encode path behind the fnordware/j2k plug-in's encode entry.

Layout:
  csrc/   hand-written HIP kernels for gfx950 + the C ABI (include/j2k_hip.h) -> libj2k_hip.so
  host/   C++ mirror of the reference's codec interface (HipCodec : j2k::Codec) -> libj2k_host.so
  api.py  ctypes harness over the C ABI (tests / bench plumbing)
  synth.py seeded synthetic frames in the After Effects buffer layout
"""

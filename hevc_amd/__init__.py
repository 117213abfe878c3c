"""hevc_amd — MI355X-native HEVC (HM 9.1) picture-reconstruction engine.

The product is the HIP library hevc_amd/libhevcrecon.so behind the C ABI of
include/hevcrecon.h.  The Python modules here are plumbing for tests and the
benchmark: abi (ctypes mirror of the header), syntax (dump/fixture I/O) and
engine (thin object wrapper over the C ABI).
"""

"""arm-spmv_amd — MI355X-native SpMV engine behind the API of ChuheHong/arm-spmv.

The directory name carries a hyphen (it mirrors the reference repository's name), so it is imported
through `__graft_entry__.load_package()`, which registers it as `arm_spmv_amd`.

  csrc/   hand-written HIP kernels + the extern "C" ABI (include/spmv_abi.h) -> lib/libspmv_hip.so
  host/   C++ source-compatible shim of the reference's classes / functions + harness
  capi    ctypes binding of the ABI (used by tests/ and bench.py)
  synth   numpy twin of the device-side synthetic matrix generators
  dist    row-range sharding over torch.distributed (one process per GPU, RCCL)
"""
from . import capi, synth  # noqa: F401

__all__ = ["capi", "synth"]

"""plan_amd — MI355X-native execution backend for daviszhen/plan's pkg/compute hot path.

Layout:
  plan_amd/csrc/      HIP kernels (gfx950) + the C-ABI (include/planhip.h) -> libplanhip.so
  plan_amd/hip.py     ctypes binding of that C-ABI (raises if the library is missing: there is
                      no CPU fallback anywhere in this package)
  plan_amd/chunk.py   host-side mirror of pkg/chunk (Chunk / Vector / SelectVector)
  plan_amd/exec.py    host-side mirror of the OperatorExec executors that drive the C-ABI
  plan_amd/tpchgen.py synthetic TPC-H data (libtpchgen.so)
"""
__all__ = ["tpchgen"]

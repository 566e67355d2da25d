"""plan_amd — MI355X-native execution backend for daviszhen/plan's pkg/compute hot path.

Layout:
  plan_amd/csrc/        HIP kernels (gfx950) + the C-ABI (include/planhip.h) -> libplanhip.so
  plan_amd/csrc/host/   C++ mirror of pkg/chunk and of the OperatorExec executors (-> libplanhost.so,
                        host_tester): the host side above the C-ABI, as the Go shim would be
  plan_amd/hip.py       ctypes binding of the C-ABI (raises if the library is missing: there is no CPU
                        fallback anywhere in this package)
  plan_amd/pipelines.py Q3 / Q9 assembled from the operator-granular calls (single GPU and N ranks)
  plan_amd/dist.py      the exchange protocol over ph_comm_* (RCCL), with two host-memory test doubles
  plan_amd/queries.py   plan descriptors of the fused Q1 / Q6 scans
  plan_amd/tpch.py      the 22 TPC-H queries as ph_plan operator trees (what the reference's planner would hand over) + result text
  plan_amd/loader.py    Arrow / parquet columns -> resident tables
  plan_amd/tpchgen.py   synthetic TPC-H data (libtpchgen.so)
"""
__all__ = ["tpchgen"]

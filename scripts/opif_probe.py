"""Q3 / Q1 behind the C++ operator layer from Python: does the process around the library matter? usage: python scripts/opif_probe.py [torch]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "torch":
    import torch  # noqa: F401
    torch.cuda.init()
from plan_amd import hip
ctx = hip.Ctx(0)
lib = ctypes.CDLL(os.path.join(ROOT, "plan_amd", "libplantpch.so"))
lib.planhost_last_error.restype = ctypes.c_char_p
db = ctypes.c_void_p()
assert lib.planhost_tpch_load(ctx.h, ctypes.c_int64(10), ctypes.c_int64(1), ctypes.byref(db)) == 0
for q in (3, 1, 9):
    avg, mn = ctypes.c_double(), ctypes.c_double()
    text, explain = ctypes.create_string_buffer(1 << 16), ctypes.create_string_buffer(1 << 14)
    rc = lib.planhost_tpch_run(db, ctypes.c_int32(q), ctypes.c_int32(50), ctypes.c_int32(8), ctypes.byref(avg), ctypes.byref(mn), text, ctypes.c_int64(len(text)), explain, ctypes.c_int64(len(explain)))
    print(f"Q{q}: rc {rc} avg {avg.value:.4f} ms min {mn.value:.4f} ms", "(torch loaded)" if "torch" in sys.modules else "")
lib.planhost_tpch_free(db)

"""A/B of the gemm8p tail split (debug bit 128 switches it off) on the bench shapes.  python scripts/g8_split_ab.py [M]"""
import os, sys, runpy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from idvs.morec_amd import _lib
_lib.lib().morec_tuning_set(b"gemm8p_tail_split", 1)
M = sys.argv[1] if len(sys.argv) > 1 else "51200"
for dbg, bias, name in ((128, 6, "tail split off"), (0, 2, "on, bias 2"), (0, 4, "on, bias 4"), (0, 6, "on, bias 6"), (0, 8, "on, bias 8"), (0, 10, "on, bias 10")):
    print("====", name, flush=True)
    _lib.lib().morec_tuning_set(b"gemm8p_debug", dbg)
    _lib.lib().morec_tuning_set(b"gemm8p_tail_bias", bias)
    sys.argv = ["gemm8p_check.py", M, "--no-check"]
    runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "gemm8p_check.py"), run_name="__main__")

"""Which torch-side (non-library) kernels run inside one bench step?  torch.profiler around a few steps of bench.py's loop,
listing aten ops with device time.   python scripts/torch_glue_profile.py [bench.py args]"""
import os, sys, runpy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity
from idvs.morec_amd import train_step

orig = train_step.TrainStep.step
state = {"n": 0, "prof": None}


def step(self, *a, **k):
    state["n"] += 1
    if state["n"] == 4:
        state["prof"] = profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=False)
        state["prof"].__enter__()
    r = orig(self, *a, **k)
    if state["n"] == 5 and state["prof"] is not None:
        torch.cuda.synchronize()
        state["prof"].__exit__(None, None, None)
        print(state["prof"].key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total", row_limit=45, max_name_column_width=60), file=sys.stderr)
        state["prof"] = None
    return r


train_step.TrainStep.step = step
sys.argv = [os.path.join(ROOT, "bench.py")] + (sys.argv[1:] or ["--steps", "4", "--warmup", "3", "--no-cpu-baseline", "--no-secondary"])
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")

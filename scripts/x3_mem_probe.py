import sys, os, runpy, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from idvs.morec_amd import train_step
orig = train_step.TrainStep.step
n = {"i": 0}
import time
def step(self, *a, **k):
    t0 = time.perf_counter()
    r = orig(self, *a, **k)
    torch.cuda.synchronize()
    n["i"] += 1
    print(f"step {n['i']}: {1e3*(time.perf_counter()-t0):.1f} ms, allocated {torch.cuda.memory_allocated()/2**30:.1f} GiB, reserved {torch.cuda.memory_reserved()/2**30:.1f} GiB, max alloc {torch.cuda.max_memory_allocated()/2**30:.1f}", file=sys.stderr)
    return r
train_step.TrainStep.step = step
sys.argv = [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'bench.py'), '--dtype', 'fp32x3', '--steps', '5', '--warmup', '3', '--no-cpu-baseline', '--no-secondary']
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'bench.py'), run_name='__main__')

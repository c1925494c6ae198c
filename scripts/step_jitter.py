"""Per-step wall times of bench.py's loop (host clock, one synchronisation per step) + caching-allocator counters: is a slow run a few
stalled steps or uniformly slow ones?   python scripts/step_jitter.py [bench.py args]"""
import os, sys, runpy, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from idvs.morec_amd import train_step

orig = train_step.TrainStep.step
times = []


def step(self, *a, **k):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = orig(self, *a, **k)
    t1 = time.perf_counter()           # host time to ISSUE the step
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    st = torch.cuda.memory_stats()
    times.append((1e3 * (t1 - t0), 1e3 * (t2 - t0), st["num_device_alloc"], st["num_device_free"], st["reserved_bytes.all.current"] >> 20))
    return r


train_step.TrainStep.step = step
sys.argv = [os.path.join(ROOT, "bench.py")] + (sys.argv[1:] or ["--steps", "30", "--warmup", "5", "--no-cpu-baseline", "--no-secondary"])
try:
    runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
finally:
    for i, t in enumerate(times):
        print(f"step {i:3d}: issue {t[0]:8.2f} ms  total {t[1]:8.2f} ms  device allocs {t[2]} frees {t[3]} reserved {t[4]} MiB", file=sys.stderr)

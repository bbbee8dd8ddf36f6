"""Where one bench step spends its wall time: sampler loops vs metric kernels vs host-side metric assembly."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
torch.cuda.set_device(0)
spec = bench.CONFIGS[1]
wl = bench.PairWorkload(spec, torch.device("cuda:0"), 0, spec["batch"])
wl.step(1, None); torch.cuda.synchronize()
import types
eng = wl.engine
orig_sums, orig_w1, orig_bsm = eng.device_metric_sums, eng.device_wasserstein, eng.batch_scalar_metrics
marks = {}
def wrap(name, fn, sync=True):
    def f(*a, **k):
        if sync: torch.cuda.synchronize()
        t0 = time.perf_counter(); r = fn(*a, **k)
        if sync: torch.cuda.synchronize()
        marks[name] = marks.get(name, 0.0) + time.perf_counter() - t0
        return r
    return f
eng.device_metric_sums = wrap("metric_sums", orig_sums)
eng.device_wasserstein = wrap("wasserstein", orig_w1)
eng.batch_scalar_metrics = wrap("host_scalar_metrics", orig_bsm, sync=False)
t0 = time.perf_counter()
for _ in range(3):
    wl.step(1, None)
torch.cuda.synchronize()
tot = time.perf_counter() - t0
print("step ms", tot / 3 * 1e3, {k: round(v / 3 * 1e3, 3) for k, v in marks.items()})

"""Wall time of the teacher loop alone, the student loop alone, and both on their own streams (bench workload)."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from distillation_trajectories_amd._hip import RULE_PSAMPLE
torch.cuda.set_device(0)
spec = bench.CONFIGS[1]
wl = bench.PairWorkload(spec, torch.device("cuda:0"), 0, spec["batch"])
def run(i):
    (h, (lo, hi)), traj = wl.parts[i][0], wl.part_traj[i][0]
    tb = h.time_bias_general(wl.tb_t, wl.tb_cond, wl.tb_present, 2 * wl.T)
    traj[0].copy_(wl.x_T[lo:hi])
    h.sample(RULE_PSAMPLE, traj, wl.H, wl.H, tb, 2, wl.coef, wl.has_noise, z=wl.z, z_shift=wl.z_shift, w_scalar=spec["guidance"])
def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
def both():
    def worker(i):
        with torch.cuda.stream(wl.streams[i][0]):
            run(i)
    ts = [threading.Thread(target=worker, args=(i,)) for i in (0, 1)]
    [t.start() for t in ts]; [t.join() for t in ts]
    torch.cuda.synchronize()
print(f"teacher alone {timed(lambda: run(0)):.1f} ms   student alone {timed(lambda: run(1)):.1f} ms   both streams {timed(both):.1f} ms")

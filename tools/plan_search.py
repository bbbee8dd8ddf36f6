"""Best-of-N launch plans, judged END TO END: the autotuner times layers alone on an idle chip and its picks between near-equal
candidates differ from run to run; here N fresh processes each tune the shapes of one bench configuration
(DT_AUTOTUNE=1, no table) and record their plans, every candidate table is then benchmarked twice more as a table (no tuning in
the process), and the candidates are listed by their median value next to the committed table.
    python tools/plan_search.py [config] [N]        (run on the GPU box; writes gpurun_out/plan_search_c<config>/)"""
import json, os, statistics, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
config = sys.argv[1] if len(sys.argv) > 1 else "1"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 6
out = os.path.join(ROOT, "gpurun_out", f"plan_search_c{config}")
os.makedirs(out, exist_ok=True)


def bench(env):
    e = dict(os.environ, **env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", config, "--no-cpu-baseline", "--no-profile"],
                       capture_output=True, text=True, env=e)
    try:
        return json.loads(r.stdout.strip().splitlines()[-1])["value"]
    except Exception:
        sys.stderr.write(r.stderr[-2000:])
        return 0.0


rows = []
base = [bench({}) for _ in range(3)]
print("committed table:", base, flush=True)
for i in range(N):
    cache = os.path.join(out, f"cache_{i}.json")
    if os.path.exists(cache):
        os.remove(cache)
    v0 = bench({"DT_PLAN_TABLE": "", "DT_AUTOTUNE": "1", "DT_TUNE_CACHE": cache})
    table = os.path.join(out, f"table_{i}.json")
    json.dump({"meta": {"made_by": "tools/plan_search.py"}, "plans": json.load(open(cache))}, open(table, "w"))
    vs = [bench({"DT_PLAN_TABLE": table}) for _ in range(2)]
    rows.append((statistics.median(vs + [v0]), v0, vs, table))
    print(f"candidate {i}: tuned run {v0:.0f}, as a table {vs}", flush=True)
rows.sort(reverse=True)
print("best:", rows[0], "committed median", statistics.median(base))

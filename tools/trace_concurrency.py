"""Concurrency analysis of a rocprofv3 --kernel-trace CSV of `bench.py --config 2` (DT_BENCH_MARKERS=1): between the two
profile markers, per hardware queue: kernels, busy time, idle gaps; overall: span, union busy time, average concurrency,
time with exactly k kernels in flight."""
import csv, glob, sys
from collections import defaultdict
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
marks = [i for i, r in enumerate(rows) if "profile_marker" in r[2]]
lo, hi = (marks[-2], marks[-1]) if len(marks) >= 2 else (0, len(rows) - 1)
sel = rows[lo + 1: hi]
t0, t1 = rows[lo][1], rows[hi][0]
span = t1 - t0
print(f"{len(sel)} dispatches, span {span / 1e6:.2f} ms")
per_q = defaultdict(list)
for s, e, n, q, st in sel:
    per_q[(q, st)].append((s, e, n))
for q, lst in sorted(per_q.items()):
    busy = sum(e - s for s, e, _ in lst)
    gaps = [lst[i + 1][0] - lst[i][1] for i in range(len(lst) - 1)]
    pos = [g for g in gaps if g > 0]
    big = sorted(pos)[-5:]
    print(f"queue/stream {q}: {len(lst)} kernels, first {(lst[0][0] - t0) / 1e6:.1f} ms last {(lst[-1][1] - t0) / 1e6:.1f} ms, busy {busy / 1e6:.1f} ms, "
          f"gaps>0: {len(pos)} sum {sum(pos) / 1e6:.1f} ms, median {sorted(pos)[len(pos) // 2] / 1e3 if pos else 0:.1f} us, largest {[round(g / 1e3) for g in big]} us")
ev = []
for s, e, *_ in sel:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
depth, last, hist = 0, t0, defaultdict(int)
for t, d in ev:
    hist[depth] += t - last
    last = t
    depth += d
hist[0] += t1 - last
tot = sum(e - s for s, e, *_ in sel)
print("time with k kernels in flight (ms):", {k: round(v / 1e6, 1) for k, v in sorted(hist.items())})
print(f"sum of durations {tot / 1e6:.1f} ms, average concurrency while busy {tot / max(1, span - hist[0]):.2f}")
by_name = defaultdict(lambda: [0, 0])
for s, e, n, *_ in sel:
    k = n.split("(")[0].replace("void dt::", "")[:60]
    by_name[k][0] += 1; by_name[k][1] += e - s
for k, (c, d) in sorted(by_name.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"  {d / 1e6:8.1f} ms {c:7d} x {d / c / 1e3:7.1f} us  {k}")

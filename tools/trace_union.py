"""Busy-time analysis of a rocprofv3 --kernel-trace CSV: wall span, union of kernel intervals, sum of durations."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
# keep the last third (timed steps), approx: use all sampler kernels after the last autotune burst
t_lo = rows[len(rows) // 2][0]
rows = [r for r in rows if r[0] >= t_lo]
span = rows[-1][1] - rows[0][0]
busy, cur_s, cur_e = 0, rows[0][0], rows[0][1]
for s, e, _ in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = sum(e - s for s, e, _ in rows)
print(f"kernels {len(rows)} span {span/1e6:.2f} ms, union busy {busy/1e6:.2f} ms ({busy/span:.1%}), sum of durations {tot/1e6:.2f} ms (avg concurrency {tot/busy:.2f})")

#!/usr/bin/env python3
"""Fold the rocprofv3 passes of profiles/collect_rNN.sh into the committed summaries (argument: the round tag, e.g. r02):
  profiles/<tag>_kernel_stats.csv  per-kernel launch statistics of the TIMED steps (from the --kernel-trace dispatches
                                   between bench.py's two profile_marker_kernel launches; rocprof's own --stats file
                                   would also count the autotuner's trial launches)
  profiles/<tag>_hbm_traffic.json  FETCH_SIZE / WRITE_SIZE passes, per kernel class, per launch.  Both are in KiB;
                                   FETCH_SIZE is doubled per the gfx950 note of MI355X_MICROARCH.md
  profiles/<tag>_sq_counters.json  SQ counters of the dominant kernel template (summed over its launches in the timed
                                   steps, per instantiation and in total) with the derived rates
"""
import collections, csv, glob, json, os, re, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "gpurun_out")
METHOD = ("rocprofv3 --kernel-trace [--pmc ...] in separate passes over `bench.py --steps 2 --warmup 1 --serial` (tags ending "
          "in c2: `--config 2 --steps 1 --warmup 1 --serial`) (profiles/collect_%s.sh), dispatches between the two "
          "profile_marker_kernel launches only" % tag.replace("c2", ""))


def klass(name):
    m = re.search(r"dt::(\w+)(?:<([^>]*)>)?", name)
    if not m:
        return None
    base, args = m.group(1), m.group(2)
    if args and base.startswith("conv_"):
        a = [x.strip() for x in args.split(",")]
        # conv_strip_bf16x6_kernel<BM, BN, ABL, KC, WK>: WK > 1 = K split across the waves (named <BM,BN,K{WK}> in the library)
        if base == "conv_strip_bf16x6_kernel" and len(a) >= 5 and a[4] not in ("1", "1u"):
            return f"{base}<{a[0]},{a[1]},K{a[4]}>"
        return f"{base}<{a[0]},{a[1]}>"
    return base


def find(pass_dir, suffix):
    files = glob.glob(os.path.join(out, pass_dir, "**", "*" + suffix), recursive=True)
    return files[0] if files else None


def timed_window(pass_dir):
    """(lo, hi) Dispatch_Id of the two markers in this pass's kernel trace (ids are per process, in launch order)."""
    f = find(pass_dir, "kernel_trace.csv")
    if not f:
        return None
    marks = [int(r["Dispatch_Id"]) for r in csv.DictReader(open(f)) if "profile_marker_kernel" in r["Kernel_Name"]]
    return (marks[0], marks[-1]) if len(marks) >= 2 else None


def kernel_stats(pass_dir):
    f, win = find(pass_dir, "kernel_trace.csv"), timed_window(pass_dir)
    if not f or not win:
        return None
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if win[0] < int(r["Dispatch_Id"]) < win[1]:
            acc[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    total = sum(sum(v) for v in acc.values())
    rows = [(k, len(v), sum(v), sum(v) / len(v), 100.0 * sum(v) / total, min(v), max(v)) for k, v in acc.items()]
    rows.sort(key=lambda r: -r[2])
    return rows


def counters(pass_dir):
    """{kernel class: {counter: [sum over launches, launches]}} within the timed window."""
    f, win = find(pass_dir, "counter_collection.csv"), timed_window(pass_dir)
    if not f or not win:
        return {}
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for r in csv.DictReader(open(f)):
        if not (win[0] < int(r["Dispatch_Id"]) < win[1]):
            continue
        k = klass(r["Kernel_Name"])
        if k:
            c = acc[k][r["Counter_Name"]]
            c[0] += float(r["Counter_Value"]); c[1] += 1
    return acc


def launch_gaps(pass_dir):
    """Idle time between consecutive dispatches of the timed window (one stream in --serial runs): start[i+1] - end[i]."""
    f, win = find(pass_dir, "kernel_trace.csv"), timed_window(pass_dir)
    if not f or not win:
        return None
    d = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), klass(r["Kernel_Name"]) or r["Kernel_Name"])
               for r in csv.DictReader(open(f)) if win[0] < int(r["Dispatch_Id"]) < win[1])
    gaps = sorted(max(0, d[i + 1][0] - d[i][1]) for i in range(len(d) - 1))
    busy = sum(e - s0 for s0, e, _ in d)
    span = d[-1][1] - d[0][0]
    q = lambda x: gaps[int(x * (len(gaps) - 1))]
    return {"dispatches": len(d), "span_ms": span / 1e6, "kernel_busy_ms": busy / 1e6, "idle_between_kernels_ms": (span - busy) / 1e6,
            "gap_ns": {"median": q(0.5), "p10": q(0.1), "p90": q(0.9), "mean": sum(gaps) / len(gaps)}, "method": METHOD}


def forward_timelines(pass_dir):
    """Launch-by-launch listing of one forward + update of each model in the serial loop: dispatches between two
    consecutive cfg_update launches, the 25th of the first loop (teacher) and the 25th of the second (student)."""
    f, win = find(pass_dir, "kernel_trace.csv"), timed_window(pass_dir)
    if not f or not win:
        return None
    d = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), klass(r["Kernel_Name"]) or r["Kernel_Name"].split("(")[0][-40:])
               for r in csv.DictReader(open(f)) if win[0] < int(r["Dispatch_Id"]) < win[1])
    ends = [i for i, x in enumerate(d) if x[2] == "cfg_update_kernel"]
    out = {}
    for name, k in (("teacher_forward_25", 24), ("student_forward_25", 74)):
        if k + 1 >= len(ends):
            continue
        seg = d[ends[k] + 1: ends[k + 1] + 1]
        out[name] = {"launches": len(seg), "span_us": (seg[-1][1] - seg[0][0]) / 1e3, "busy_us": sum(e - s0 for s0, e, _ in seg) / 1e3,
                     "sequence": [[n, round((e - s0) / 1e3, 1)] for s0, e, n in seg]}
    return out


tl = forward_timelines(tag + "_stats")
if tl:
    json.dump(tl, open(os.path.join(root, "profiles", f"{tag}_forward_timeline.json"), "w"), indent=1)
    print(f"{tag}_forward_timeline.json:", {k: (v["launches"], round(v["span_us"], 1)) for k, v in tl.items()})

gaps = launch_gaps(tag + "_stats")
if gaps:
    json.dump(gaps, open(os.path.join(root, "profiles", f"{tag}_launch_gaps.json"), "w"), indent=1)
    print(f"{tag}_launch_gaps.json:", gaps["gap_ns"], f"idle {gaps['idle_between_kernels_ms']:.2f} of {gaps['span_ms']:.2f} ms")

rows = kernel_stats(tag + "_stats")
if rows:
    with open(os.path.join(root, "profiles", f"{tag}_kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            w.writerow([r[0], r[1], r[2], f"{r[3]:.1f}", f"{r[4]:.2f}", r[5], r[6]])
    print(f"{tag}_kernel_stats.csv:", len(rows), "kernels;", rows[0][0][:60], f"{rows[0][4]:.1f} %")

fetch, write = counters(tag + "_fetch"), counters(tag + "_write")
res = {}
for k in sorted(set(fetch) & set(write)):
    fs, ws = fetch[k]["FETCH_SIZE"], write[k]["WRITE_SIZE"]
    f, w = fs[0] / fs[1] * 1024 * 2, ws[0] / ws[1] * 1024
    res[k] = {"hbm_bytes_per_launch": int(f + w), "fetch_bytes_per_launch_corrected_x2": int(f), "write_bytes_per_launch": int(w),
              "launches_sampled": fs[1], "method": METHOD + "; FETCH_SIZE (KiB) doubled per the gfx950 note, WRITE_SIZE (KiB) as is"}
if res:
    json.dump(res, open(os.path.join(root, "profiles", f"{tag}_hbm_traffic.json"), "w"), indent=1)
    print(f"{tag}_hbm_traffic.json:", len(res), "kernel classes")

sq = {}
for d in (tag + "_sq_a", tag + "_sq_b"):
    for k, cs in counters(d).items():
        for name, (v, n) in cs.items():
            sq.setdefault(k, {})[name] = {"sum": v, "launches": n}
if sq and rows:
    # dominant kernel TEMPLATE by time in the stats pass
    by_tpl = collections.defaultdict(float)
    for r in rows:
        k = klass(r[0])
        if k:
            by_tpl[k.split("<")[0]] += r[2]
    dom = max(by_tpl, key=by_tpl.get)
    members = {k: v for k, v in sq.items() if k.split("<")[0] == dom}
    total = collections.defaultdict(float)
    for cs in members.values():
        for name, c in cs.items():
            total[name] += c["sum"]
    def derived(c):
        g = lambda n: c.get(n, 0.0)
        d = {}
        if g("SQ_INSTS_MFMA"):
            d["mfma_busy_cycles_per_mfma"] = g("SQ_VALU_MFMA_BUSY_CYCLES") / g("SQ_INSTS_MFMA")
            d["valu_per_mfma"] = g("SQ_INSTS_VALU") / g("SQ_INSTS_MFMA")
            d["salu_per_mfma"] = g("SQ_INSTS_SALU") / g("SQ_INSTS_MFMA")
            if g("SQ_INSTS_LDS"):
                d["lds_insts_per_mfma"] = g("SQ_INSTS_LDS") / g("SQ_INSTS_MFMA")
        if g("SQ_BUSY_CYCLES"):
            # SQ_VALU_MFMA_BUSY_CYCLES counts cycles of the 1024 SIMD matrix pipes; SQ_BUSY_CYCLES is per SE-level SQ
            # (32 shader engines on this part): matrix-pipe utilisation = busy / (SQ busy cycles / 32 * 1024 SIMDs)
            d["mfma_pipe_utilisation"] = g("SQ_VALU_MFMA_BUSY_CYCLES") / (g("SQ_BUSY_CYCLES") / 32.0 * 1024.0)
        if g("SQ_WAVE_CYCLES"):
            d["wave_time_waiting_frac"] = g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES")
            d["wave_time_issue_stalled_frac"] = g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES")
        if g("SQ_LDS_IDX_ACTIVE"):
            d["lds_bank_conflict_frac"] = g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE")
        return d
    # the other hand-written kernels of the round (whole-forward small-model kernel, one-pass metrics): same derived rates
    others = {k: {"counters": {n: c["sum"] for n, c in cs.items()}, "launches": max(c["launches"] for c in cs.values()),
                  "derived": derived({n: c["sum"] for n, c in cs.items()})}
              for k, cs in sorted(sq.items()) if k in ("unet_fused_kernel", "pair_metrics_kernel", "cfg_update_kernel")}
    doc = {"kernel": dom + "<BM,BN>", "method": METHOD, "other_kernels": others,
           "total": {"counters": dict(total), "derived": derived(total)},
           "instantiations": {k: {"counters": {n: c["sum"] for n, c in cs.items()}, "launches": max(c["launches"] for c in cs.values()),
                                  "derived": derived({n: c["sum"] for n, c in cs.items()})} for k, cs in sorted(members.items())}}
    json.dump(doc, open(os.path.join(root, "profiles", f"{tag}_sq_counters.json"), "w"), indent=1)
    print(f"{tag}_sq_counters.json:", dom, {k: round(v, 3) for k, v in doc["total"]["derived"].items()})

#!/usr/bin/env python3
"""Fold the rocprofv3 passes of profiles/collect_r01.sh into the committed summaries:
  profiles/r01_kernel_stats.csv  (from --kernel-trace --stats)
  profiles/hbm_traffic.json      (FETCH_SIZE / WRITE_SIZE passes, per kernel class, per launch)
FETCH_SIZE and WRITE_SIZE are in KiB; FETCH_SIZE is doubled per the gfx950 note of MI355X_MICROARCH.md."""
import collections, csv, glob, json, os, re, shutil, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "gpurun_out")


def klass(name):
    m = re.search(r"dt::(\w+)(?:<([^>]*)>)?", name)
    if not m:
        return None
    base, args = m.group(1), m.group(2)
    if args and base.startswith("conv_"):
        a = [x.strip() for x in args.split(",")]
        return f"{base}<{a[0]},{a[1]}>"
    return base


def per_launch(pattern, counter):
    files = glob.glob(os.path.join(out, pattern, "**", "*counter_collection.csv"), recursive=True)
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = klass(r["Kernel_Name"])
            if k:
                acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
    return acc


fetch, write = per_launch("pmc_fetch", "FETCH_SIZE"), per_launch("pmc_write", "WRITE_SIZE")
res = {}
for k in sorted(set(fetch) & set(write)):
    f = fetch[k][0] / fetch[k][1] * 1024 * 2
    w = write[k][0] / write[k][1] * 1024
    res[k] = {"hbm_bytes_per_launch": int(f + w), "fetch_bytes_per_launch_corrected_x2": int(f), "write_bytes_per_launch": int(w),
              "launches_sampled": fetch[k][1],
              "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (KiB units), FETCH_SIZE doubled per "
                        "MI355X_MICROARCH.md gfx950 note; bench.py --steps 2 --warmup 1 --serial"}
if res:
    json.dump(res, open(os.path.join(root, "profiles", "hbm_traffic.json"), "w"), indent=1)
    print("hbm_traffic.json:", len(res), "kernel classes")
stats = glob.glob(os.path.join(out, "prof_r01c", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], os.path.join(root, "profiles", "r01_kernel_stats.csv"))
    print("copied", stats[0])

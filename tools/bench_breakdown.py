#!/usr/bin/env python3
"""Pretty-print the per-kernel table and tile choices of a bench.py JSON line (stdin or file)."""
import json, sys
src = open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read()
d = json.loads(src.strip().splitlines()[-1])
print(d["value"], d["unit"], d["ms_per_step"], "ms/step")
tot = sum(v["ms"] for v in d["kernels"].values())
for k, v in d["kernels"].items():
    print(f"{k:42s} {v['launches']:5d} {v['ms']:8.2f} ms {v['ms'] / tot:6.1%}  TF/s {v['tflops']}  GB/s {v['gbps']}")
print("serial kernel time per step", round(tot / d["steps"], 2), "ms")
for sf, rows in d["tile_choices"].items():
    print(sf, "; ".join(f"{r[0]}.{r[1]} {r[2]}x{r[3]}/s{r[4]} {r[5]}" for r in rows))

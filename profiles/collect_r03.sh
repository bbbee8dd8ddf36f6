# Round-3 profile collection (run on the GPU box from the repo root via gpurun: bash profiles/collect_r03.sh).
# Every pass of a configuration runs the SAME command; counters are collected in their own passes with --kernel-trace only (no
# other trace domain), FETCH_SIZE and WRITE_SIZE separately (TCC slots), per MI355X_MICROARCH.md.  DT_BENCH_MARKERS=1 makes
# bench.py launch `profile_marker_kernel` before and after the timed steps; summarize_pmc.py keeps only the dispatches in
# between.  --serial: one stream, so per-kernel durations are free of cross-stream contention (the headline `value` is
# measured without the profiler, on several streams).  Launch plans come from the committed table (plans/gfx950.json): every
# pass replays the same plan.  Tag r03 = configs[1] (the headline workload), r03c2 = configs[2] (the size sweep).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export DT_BENCH_MARKERS=1
run() { d=$1; shift; rocprofv3 --kernel-trace "$@" --output-format csv -d gpurun_out/$d -o p -- $B > gpurun_out/$d.log 2>&1; echo "$d rc=$?"; }
passes() { t=$1
  run ${t}_stats --stats &&
  run ${t}_fetch --pmc FETCH_SIZE &&
  run ${t}_write --pmc WRITE_SIZE &&
  run ${t}_sq_a --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE &&
  run ${t}_sq_b --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_WAVES &&
  python3 profiles/summarize_pmc.py $t
  rm -rf gpurun_out/${t}_stats gpurun_out/${t}_fetch gpurun_out/${t}_write gpurun_out/${t}_sq_a gpurun_out/${t}_sq_b
}
B="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --serial"
passes r03
B="python3 bench.py --config 2 --steps 1 --warmup 1 --no-cpu-baseline --no-profile --serial"
passes r03c2
# the box only returns gpurun_out/: park the summaries there too (copy them into profiles/ afterwards)
mkdir -p gpurun_out/profiles_r03 && cp profiles/r03_*.csv profiles/r03_*.json profiles/r03c2_*.csv profiles/r03c2_*.json gpurun_out/profiles_r03/ 2>/dev/null

# Round-2 profile collection (run on the GPU box from the repo root via gpurun: bash profiles/collect_r02.sh).
# Every pass runs the SAME command; counters are collected in their own passes with --kernel-trace only (no other trace
# domain), FETCH_SIZE and WRITE_SIZE separately (TCC slots), per MI355X_MICROARCH.md.  DT_BENCH_MARKERS=1 makes bench.py
# launch `profile_marker_kernel` before and after the timed steps; summarize_pmc.py keeps only the dispatches in between
# (the autotuner's trial launches at start-up are dropped).  --serial: one stream, so per-kernel durations are free of
# cross-stream contention (the headline `value` is measured without the profiler, on two streams).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export DT_BENCH_MARKERS=1
export DT_TUNE_CACHE=gpurun_out/r02_tune_cache.json     # the first pass tunes, the others replay its launch plan
rm -f $DT_TUNE_CACHE
B="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --serial"
run() { d=$1; shift; rocprofv3 --kernel-trace "$@" --output-format csv -d gpurun_out/$d -o p -- $B > gpurun_out/$d.log 2>&1; echo "$d rc=$?"; }
run r02_stats --stats &&
run r02_fetch --pmc FETCH_SIZE &&
run r02_write --pmc WRITE_SIZE &&
run r02_sq_a --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE &&
run r02_sq_b --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_WAVES
python3 profiles/summarize_pmc.py r02
# the box only returns gpurun_out/: park the summaries there too (copy them into profiles/ afterwards), drop the raw traces
mkdir -p gpurun_out/profiles_r02 && cp profiles/r02_*.csv profiles/r02_*.json gpurun_out/profiles_r02/ 2>/dev/null
rm -rf gpurun_out/r02_stats gpurun_out/r02_fetch gpurun_out/r02_write gpurun_out/r02_sq_a gpurun_out/r02_sq_b

# Round-1 profile collection (run on the GPU box from the repo root via gpurun).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --serial"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r01c -o r01c -- $B > gpurun_out/prof_r01c.log 2>&1 &&
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o f -- $B > gpurun_out/pmc_fetch.log 2>&1 &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o w -- $B > gpurun_out/pmc_write.log 2>&1
echo rc=$? && python3 profiles/summarize_pmc.py

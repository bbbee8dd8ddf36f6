#!/bin/bash
# Rehearse bench.py's multi-rank path on a box with fewer GPUs than ranks: N rank processes started by
# bench.py's own launcher (no torchrun), gloo instead of RCCL for the metric all-gather, all ranks on GPU 0.
# Usage: tools/rehearse_ranks.sh [N=2] [extra bench.py args]   (keep N <= 6: the GPU box admits 6 GPU processes)
set -euo pipefail
N=${1:-2}
shift || true
cd "$(dirname "$0")/.."
DT_BENCH_BACKEND=gloo python3 bench.py --gpus "$N" --steps 2 --warmup 1 --no-profile --batch 64 "$@"

#!/bin/bash
# Kernel-time statistics of a group of S loops (tools/group_loop.py) on the GPU box.
# Usage (through gpurun): bash tools/profile_group.sh <tag> <S> [frames]
set -u
TAG=${1:-g}
S=${2:-16}
FRAMES=${3:-60}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT" || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o group -- \
    python3 tools/group_loop.py --frames "$FRAMES" --seqs "$S" > "$OUT/group_stats.log" 2>&1 || exit 2
python3 tools/kernel_table.py "$OUT/stats"

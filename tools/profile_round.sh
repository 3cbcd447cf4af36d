#!/bin/bash
# Collects the rocprofv3 evidence of one round on the GPU box: kernel-time statistics of the default
# benchmark, then HBM traffic and SQ counters in separate counter passes (each its own run).
# Usage (through gpurun): bash tools/profile_round.sh <tag>
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT" || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o bench -- \
    python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline > "$OUT/bench_stats.log" 2>&1 || exit 2
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o bench -- \
    python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > "$OUT/pmc_fetch.log" 2>&1 || exit 3
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o bench -- \
    python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > "$OUT/pmc_write.log" 2>&1 || exit 4
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY \
    --output-format csv -d "$OUT/pmc_sq" -o bench -- \
    python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > "$OUT/pmc_sq.log" 2>&1 || exit 5
find "$OUT" -name "*.csv" | head -50

#!/bin/bash
# The rocprofv3 evidence of the bench lines on the GPU box, one command per profile directory:
#   prof_<tag>_headline   bench.py --only-headline   (the headline loop and the replays that price its tracker launches)
#   prof_<tag>_group16    bench.py --only-group 16   (... and a group of 16 loops in lockstep)
# each with kernel-time statistics + trace, then HBM traffic and SQ counters in separate counter passes.
# Usage (through gpurun): bash tools/profile_round3.sh <tag> [headline|group16|both]
set -u
TAG=${1:-r03a}
WHAT=${2:-both}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd "$ROOT" || exit 1
profile() {   # name, bench arguments...
    local NAME=$1; shift
    local OUT=$ROOT/gpurun_out/prof_${TAG}_${NAME}
    mkdir -p "$OUT"
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o bench -- \
        python3 bench.py "$@" --steps 200 --warmup 10 --no-cpu-baseline > "$OUT/bench_stats.log" 2>&1 || return 2
    timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o bench -- \
        python3 bench.py "$@" --steps 40 --warmup 5 --no-cpu-baseline > "$OUT/pmc_fetch.log" 2>&1 || return 3
    timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o bench -- \
        python3 bench.py "$@" --steps 40 --warmup 5 --no-cpu-baseline > "$OUT/pmc_write.log" 2>&1 || return 4
    timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY \
        --output-format csv -d "$OUT/pmc_sq" -o bench -- \
        python3 bench.py "$@" --steps 40 --warmup 5 --no-cpu-baseline > "$OUT/pmc_sq.log" 2>&1 || return 5
    grep -h "^{\"metric\"" "$OUT/bench_stats.log" | tail -1 > "$OUT/bench_line.json"
    python3 tools/kernel_table.py "$OUT/stats" 14
}
if [ "$WHAT" = headline ] || [ "$WHAT" = both ]; then profile headline --only-headline || exit $?; fi
if [ "$WHAT" = group16 ] || [ "$WHAT" = both ]; then profile group16 --only-group 16 || exit $?; fi

#!/bin/bash
# Profiling passes over `python3 bench.py` on the GPU box; each pass is its own run.
#   tools/profile_bench.sh TAG [bench args...]
# Outputs under gpurun_out/prof_TAG/{stats,fetch,write,sq}.
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 20 --warmup 5 --no-cpu-baseline $*"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/stats.log" 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/fetch.log" 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/write.log" 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d "$OUT/sq" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/sq.log" 2>&1 || exit 1
echo done

# kernel-stats pass only: tools/quick_stats.sh TAG [bench args...]; prints the top kernels
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/qs_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/bench.py" --steps 20 --warmup 5 --no-cpu-baseline "$@" > "$OUT/stats.log" 2>&1 || { tail -5 "$OUT/stats.log"; exit 1; }
python3 "$ROOT/tools/summarize_rocprof.py" "$(find "$OUT/stats" -name '*kernel_stats.csv' | head -1)" "$TAG" | head -${TOPN:-16}
grep '^{' "$OUT/stats.log" | cut -c1-200

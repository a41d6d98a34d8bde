#!/usr/bin/env bash
# One GPU-box session: parity tests -> bench -> rocprofv3 kernel trace of the
# same bench command.  Stops at the first step that times out or is killed.
# usage: scripts/gpu_check.sh [tag] [bench args...]
set -u -o pipefail
TAG="${1:-run}"; shift || true
OUT="$GRAFT_REPO_ROOT/gpurun_out/$TAG"
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
run() { # name timeout cmd...
  local name="$1" to="$2"; shift 2
  echo "=== $name: $*"
  timeout -k 10 "$to" "$@" > "$OUT/$name.log" 2>&1
  local rc=$?
  echo "=== $name rc=$rc"
  tail -n 25 "$OUT/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "!!! $name timed out/killed: stopping"; exit $rc; fi
  return 0
}
rocminfo | grep -m1 gfx || true
run pytest 900 python -m pytest tests -m gpu -q --timeout 300 ${PYTEST_ARGS:-}
run bench 600 python bench.py "$@"
grep -h '^{' "$OUT/bench.log" > "$OUT/bench.json" || true
export TMPDIR=/tmp
run rocprof 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -- python3 bench.py "$@" --profile
grep -h '^{' "$OUT/rocprof.log" > "$OUT/bench_profiled.json" || true
find "$OUT/prof" -name '*kernel_stats*.csv' | head -1 | xargs -r -I{} cp {} "$OUT/kernel_stats.csv"
[ -f "$OUT/kernel_stats.csv" ] && head -15 "$OUT/kernel_stats.csv"
find "$OUT/prof" -name '*kernel_trace*.csv' -size +20M -delete || true
echo "=== done"

#!/usr/bin/env bash
# One GPU-box session of named steps: scripts/gpu_session.sh TAG step [step ..]; a step is "name:timeout:command".
# Output of each step -> gpurun_out/TAG/name.log (tail echoed).  Stops at the first step that times out or is killed.
set -u -o pipefail
TAG="$1"; shift
OUT="$GRAFT_REPO_ROOT/gpurun_out/$TAG"
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
rocminfo | grep -m1 gfx || true
for step in "$@"; do
  name="${step%%:*}"; rest="${step#*:}"; to="${rest%%:*}"; cmd="${rest#*:}"
  echo "=== $name: $cmd"
  timeout -k 10 "$to" bash -c "$cmd" > "$OUT/$name.log" 2>&1
  rc=$?
  echo "=== $name rc=$rc"
  tail -n 30 "$OUT/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "!!! $name timed out/killed: stopping"; exit $rc; fi
done
echo "=== done"

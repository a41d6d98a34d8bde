#!/usr/bin/env bash
# Long randomised parity session: every fuzz script with several seeds; stops at the first failure.
# usage: scripts/gpu_fuzz_all.sh TAG [rounds]
set -u -o pipefail
TAG="${1:-fuzz}"; ROUNDS="${2:-3}"
OUT="$GRAFT_REPO_ROOT/gpurun_out/$TAG"
mkdir -p "$OUT"; cd "$GRAFT_REPO_ROOT"
for r in $(seq 1 "$ROUNDS"); do
  for f in fuzz_r2 fuzz_fast fuzz_blocks fuzz_chain fuzz_frame; do
    seed=$((100 * r + 7))
    echo "=== $f seed $seed"
    timeout -k 10 280 python3 scripts/$f.py 25 $seed > "$OUT/${f}_$seed.log" 2>&1
    rc=$?
    tail -n 1 "$OUT/${f}_$seed.log" | cut -c1-300
    if [ $rc -ne 0 ]; then echo "!!! $f seed $seed failed rc=$rc"; tail -n 15 "$OUT/${f}_$seed.log"; exit 1; fi
  done
done
echo "=== all fuzz rounds ok"

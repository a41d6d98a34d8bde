#!/usr/bin/env bash
# usage: gpu_pmc_sets.sh tag "SET1 counters" "SET2 counters" ...   (one rocprofv3 --pmc pass per set)
set -u -o pipefail
TAG="$1"; shift
OUT="$GRAFT_REPO_ROOT/gpurun_out/$TAG"
mkdir -p "$OUT"; cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
i=0
for set in "$@"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/p$i" -- python3 bench.py --steps 3 --warmup 3 --no-cpu-baseline > "$OUT/p$i.log" 2>&1 || { echo "pass $i failed"; tail -5 "$OUT/p$i.log"; }
  f=$(find "$OUT/p$i" -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python3 scripts/pmc_summary.py "$f" | grep -A12 "k_ols_decim4\|k_mix_fir\|k_ols_wave\|k_ols_frame" | grep -v synth
done
find "$OUT" -name '*.csv' -size +8M -delete

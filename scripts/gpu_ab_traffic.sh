#!/usr/bin/env bash
# One session: interleaved A/B of library variants for the chain (bench.py) and cfg5, then FETCH_SIZE per variant.
# usage: scripts/gpu_ab_traffic.sh TAG "chainlib1 chainlib2 .." "cfg5variant1 cfg5variant2 .." "pmclib1 pmclib2 .." [rounds]
#   chain libs: "-" or a path (ab_bench.py syntax); cfg5 variants: bench_cfg5_variants.py syntax; pmc libs: "-" or a path
set -u -o pipefail
TAG="$1"; CH="$2"; C5="$3"; PM="$4"; R="${5:-3}"
OUT="$GRAFT_REPO_ROOT/gpurun_out/$TAG"
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
step() { local name="$1" to="$2"; shift 2
  echo "=== $name"
  timeout -k 10 "$to" "$@" > "$OUT/$name.log" 2>&1
  local rc=$?
  echo "=== $name rc=$rc"; tail -n 12 "$OUT/$name.log" | cut -c1-400
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "!!! $name timed out/killed: stopping"; exit $rc; fi
}
[ -n "$CH" ] && step ab_chain 900 python3 scripts/ab_bench.py "$R" $CH
[ -n "$C5" ] && step ab_cfg5 600 python3 scripts/bench_cfg5_variants.py "$R" $C5
i=0
for lib in $PM; do
  i=$((i+1))
  for ctr in FETCH_SIZE WRITE_SIZE; do
    name="pmc_${i}_${ctr}"
    if [ "$lib" = "-" ]; then unset RR_LIB; else export RR_LIB="$GRAFT_REPO_ROOT/$lib"; fi
    step "$name" 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d "$OUT/$name" -- python3 bench.py --profile --steps 20 --warmup 3 --settle-ms 0
    f=$(find "$OUT/$name" -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && { echo "lib=$lib"; python3 scripts/pmc_summary.py "$f" | grep -A2 "k_ols_frame\|k_ols_wave"; } > "$OUT/$name.summary.txt"
    cat "$OUT/$name.summary.txt"
    rm -rf "$OUT/$name"
    name="pmc5_${i}_${ctr}"
    step "$name" 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d "$OUT/$name" -- python3 scripts/prof_cfg5.py 12
    f=$(find "$OUT/$name" -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && { echo "lib=$lib"; python3 scripts/pmc_summary.py "$f" | grep -A2 "k_filter_blk4096"; } > "$OUT/$name.summary.txt"
    cat "$OUT/$name.summary.txt"
    rm -rf "$OUT/$name"
  done
done
unset RR_LIB
echo "=== done"

#!/usr/bin/env bash
# One GPU-box session that produces every file profiles/ keeps for a round (scripts/collect_profiles.py copies them):
#   bench.json                      python bench.py (the driver's command, defaults)
#   bench_profiled.json + kernel_stats.csv   rocprofv3 --kernel-trace --stats -- python3 bench.py --profile
#   pmc_{sq1,sq2,fetch,write}.summary.txt    separate --pmc passes of `bench.py --profile` (chain launches only)
#   cfg5_kernel_stats.csv, cfg5_pmc_*.summary.txt   the same for BASELINE configs[4] (scripts/prof_cfg5.py)
#   blocks.txt, cfg5.txt, cfg3.txt   scripts/bench_blocks.py, bench_cfg5.py, bench_cfg3.py
#   extras.txt, decim_ab.txt, meter.txt, meter_kernel_stats.csv   bench_extras.py, bench_decim_ab.py, bench_meter.py, prof_meter.py
#   shapes_kernel_stats.csv         per-kernel averages of the two-kernel chain shapes (scripts/prof_shapes.py)
#   wave2k.txt, bluestein_big.txt   k_ols_wave2k against k_ols_wave<8>, k_bluestein_big against the routes it replaces
# usage: scripts/gpu_profiles.sh TAG
set -u -o pipefail
TAG="${1:-prof}"
OUT="$GRAFT_REPO_ROOT/gpurun_out/$TAG"
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
step() { # name timeout cmd...
  local name="$1" to="$2"; shift 2
  echo "=== $name"
  timeout -k 10 "$to" "$@" > "$OUT/$name.log" 2>&1
  local rc=$?
  echo "=== $name rc=$rc"; tail -n 4 "$OUT/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "!!! $name timed out/killed: stopping"; exit $rc; fi
}
pmc() { # name script-and-args... ; counters in $CTRS
  local name="$1"; shift
  step "$name" 300 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d "$OUT/$name" -- python3 "$@"
  local f; f=$(find "$OUT/$name" -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python3 scripts/pmc_summary.py "$f" > "$OUT/$name.summary.txt"
  find "$OUT/$name" -name '*.csv' -size +8M -delete
}
SQ1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
SQ2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE"
step bench 600 python3 bench.py
grep -h '^{' "$OUT/bench.log" > "$OUT/bench.json" || true
step rocprof 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -- python3 bench.py --profile
grep -h '^{' "$OUT/rocprof.log" > "$OUT/bench_profiled.json" || true
find "$OUT/prof" -name '*kernel_stats*.csv' | head -1 | xargs -r -I{} cp {} "$OUT/kernel_stats.csv"
find "$OUT/prof" -name '*kernel_trace*.csv' -size +20M -delete || true
B="bench.py --profile --steps 20 --warmup 3 --settle-ms 0"
CTRS="$SQ1" pmc pmc_sq1 $B
CTRS="$SQ2" pmc pmc_sq2 $B
CTRS="FETCH_SIZE" pmc pmc_fetch $B
CTRS="WRITE_SIZE" pmc pmc_write $B
# (400 launches: the power controller's first 50 ms after an idle gap - faster, then slower than the steady state - are
#  diluted; 40 launches from idle measured its transient, not the kernel)
step cfg5prof 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/cfg5prof" -- python3 scripts/prof_cfg5.py 400
find "$OUT/cfg5prof" -name '*kernel_stats*.csv' | head -1 | xargs -r -I{} cp {} "$OUT/cfg5_kernel_stats.csv"
CTRS="$SQ1" pmc cfg5_pmc_sq1 scripts/prof_cfg5.py
CTRS="$SQ2" pmc cfg5_pmc_sq2 scripts/prof_cfg5.py
CTRS="FETCH_SIZE" pmc cfg5_pmc_fetch scripts/prof_cfg5.py
CTRS="WRITE_SIZE" pmc cfg5_pmc_write scripts/prof_cfg5.py
step blocks 400 python3 scripts/bench_blocks.py
step fftsizes 300 python3 scripts/fft_sizes_probe.py
step cfg5 300 python3 scripts/bench_cfg5.py
step cfg3 300 python3 scripts/bench_cfg3.py
step extras 400 python3 scripts/bench_extras.py
step decim_ab 300 python3 scripts/bench_decim_ab.py
step meter 200 python3 scripts/bench_meter.py
step meterprof 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/meterprof" -- python3 scripts/prof_meter.py
find "$OUT/meterprof" -name '*kernel_stats*.csv' | head -1 | xargs -r -I{} cp {} "$OUT/meter_kernel_stats.csv"
step shapes 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/shapesprof" -- python3 scripts/prof_shapes.py
find "$OUT/shapesprof" -name '*kernel_stats*.csv' | head -1 | xargs -r -I{} cp {} "$OUT/shapes_kernel_stats.csv"
step clock 300 python3 scripts/clock_probe.py --seconds 4 idle cfg5 chain copy cfg5 chain
grep -h '^CLOCK_PROBE_JSON' "$OUT/clock.log" | python3 scripts/clock_summary.py > "$OUT/clock_power.txt" || true
step bank 300 python3 scripts/callsize_probe.py bank 64 10 12 14 16 18
step callsize 300 python3 scripts/callsize_probe.py 10 12 14 16 18 20 22 24 26
step wave2k 300 python3 scripts/wave2k_probe.py
step bsbig 300 python3 scripts/bs_big_probe.py
step smoke 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
echo "=== done"

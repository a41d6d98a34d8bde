#!/usr/bin/env bash
# PMC passes over the bench command (kernel trace + counters only; each pass its
# own run, as the gfx950 guide prescribes).  usage: scripts/gpu_pmc.sh tag [bench args]
set -u -o pipefail
TAG="${1:-pmc}"; shift || true
OUT="$GRAFT_REPO_ROOT/gpurun_out/$TAG"
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
pass() { # name counters...
  local name="$1"; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 bench.py --steps 3 --warmup 3 --no-cpu-baseline ${BENCH_ARGS:-} > "$OUT/$name.log" 2>&1
  local rc=$?
  echo "pass $name rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
  f=$(find "$OUT/$name" -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python3 scripts/pmc_summary.py "$f" > "$OUT/$name.summary.txt" && cat "$OUT/$name.summary.txt"
  find "$OUT/$name" -name '*.csv' -size +8M -delete
}
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pass sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE
pass fetch FETCH_SIZE
pass write WRITE_SIZE
echo done

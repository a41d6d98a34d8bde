#!/usr/bin/env bash
# two SQ counter passes over any script; usage: gpu_pmc_cmd.sh tag kernel-pattern script.py [args..]
set -u -o pipefail
TAG="$1"; PAT="$2"; shift 2
OUT="$GRAFT_REPO_ROOT/gpurun_out/$TAG"
mkdir -p "$OUT"; cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$OUT/sq1" -- python3 "$@" > "$OUT/sq1.log" 2>&1 || { tail -5 "$OUT/sq1.log"; exit 1; }
f=$(find "$OUT/sq1" -name '*counter_collection.csv' | head -1)
python3 scripts/pmc_summary.py "$f" | grep -A9 "$PAT" | tee "$OUT/sq1.summary.txt"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d "$OUT/sq2" -- python3 "$@" > "$OUT/sq2.log" 2>&1 || { tail -5 "$OUT/sq2.log"; exit 1; }
f=$(find "$OUT/sq2" -name '*counter_collection.csv' | head -1)
python3 scripts/pmc_summary.py "$f" | grep -A9 "$PAT" | tee "$OUT/sq2.summary.txt"
find "$OUT" -name '*.csv' -size +8M -delete

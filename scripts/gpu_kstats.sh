#!/usr/bin/env bash
# rocprofv3 kernel statistics of one script: scripts/gpu_kstats.sh TAG script.py [args..]
# -> gpurun_out/TAG/kernel_stats.csv and a per-kernel table (calls, average ns, share) on stdout.
set -u -o pipefail
TAG="$1"; shift
OUT="$GRAFT_REPO_ROOT/gpurun_out/$TAG"
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -- python3 "$@" > "$OUT/run.log" 2>&1
echo "rc=$?"
f=$(find "$OUT/prof" -name '*kernel_stats*.csv' | head -1)
if [ -z "$f" ]; then echo "no kernel_stats.csv"; tail -n 5 "$OUT/run.log"; exit 1; fi
cp "$f" "$OUT/kernel_stats.csv"
find "$OUT/prof" -name '*kernel_trace*.csv' -size +20M -delete || true
python3 - "$OUT/kernel_stats.csv" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(r["Name"][:70].ljust(70), r["Calls"].rjust(6), f'{float(r["AverageNs"]) / 1e3:10.1f} us', r["Percentage"].rjust(7), "%")
PY

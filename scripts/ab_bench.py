#!/usr/bin/env python3
"""A/B of library variants in ONE GPU session: interleaved rounds of bench.py,
one subprocess per run (usage: ab_bench.py rounds lib1 lib2 ...; '-' = product;
'lib@kernel' also sets RR_FUSED_KERNEL=kernel, e.g. '-@olsw')."""
import json
import os
import statistics
import subprocess
import sys

rounds = int(sys.argv[1])
libs = sys.argv[2:]
res = {l: {"fir": [], "fft": [], "step": [], "rms": []} for l in libs}
for r in range(rounds):
    for l in libs:
        env = dict(os.environ)
        lib, _, kern = l.partition("@")
        if kern:
            env["RR_FUSED_KERNEL"] = kern
        if lib != "-":
            env["RR_LIB"] = os.path.abspath(lib)
        else:
            env.pop("RR_LIB", None)
        out = subprocess.run([sys.executable, "bench.py", "--steps", os.environ.get("AB_STEPS", "100"), "--warmup", "3", "--no-cpu-baseline", "--no-general-nco", "--no-host-fed"],
                             env=env, capture_output=True, text=True, timeout=300)
        line = [x for x in out.stdout.splitlines() if x.startswith("{")]
        if not line:
            print(l, "FAILED", out.stderr[-500:])
            sys.exit(1)
        d = json.loads(line[0])
        k = d["kernels"]
        res[l]["fir"].append(k.get("fused_mix_fir_decim", {}).get("avg_ms", float("nan")))
        res[l]["fft"].append(k.get("fourier", {}).get("avg_ms", 0.0))
        res[l]["step"].append(d["ms_per_step"])
        res[l]["rms"].append(d.get("parity_fused_vs_block_by_block_last_step_rms") or 0.0)
for l in libs:
    f = res[l]
    print(f"{l:40s} fir med={statistics.median(f['fir']):.4f} min={min(f['fir']):.4f}  fft med={statistics.median(f['fft']):.4f}  step med={statistics.median(f['step']):.4f} min={min(f['step']):.4f}  rms vs block-by-block {max(f['rms']):.2e}")

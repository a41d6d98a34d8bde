#!/usr/bin/env python3
"""CLOCK_PROBE_JSON line of scripts/clock_probe.py (stdin) -> a small table: per load, ms per call, the shader clock of the
eight XCDs (amd-smi, median of the samples) and the socket power (median / max)."""
import json, sys
for line in sys.stdin:
    if not line.startswith("CLOCK_PROBE_JSON "):
        continue
    d = json.loads(line[len("CLOCK_PROBE_JSON "):])
    print(f"{'load':10s} {'ms/call':>9s} {'gfx clk MHz (median per XCD: min .. max)':>44s} {'socket W med':>13s} {'max':>6s} {'samples':>8s}")
    for k, v in d.items():
        clk = sorted(v[x]["med"] for x in v if x.startswith("amd-smi") and ".clock.gfx_" in x)
        pw = v.get("amd-smi.gpu_data[0].power.socket_power") or v.get("rocm-smi.card0.Current Socket Graphics Package Power (W)") or {}
        cl = f"{clk[0]:.0f} .. {clk[-1]:.0f}" if clk else "n/a"
        print(f"{k.split('@')[0]:10s} {v['ms_per_call']:9.4f} {cl:>44s} {pw.get('med', float('nan')):13.0f} {pw.get('max', float('nan')):6.0f} {pw.get('n', 0):8d}")

#!/usr/bin/env python3
"""Copies what scripts/gpu_profiles.sh left under gpurun_out/TAG into profiles/ under the round's name and
regenerates profiles/traffic_fused_fir.json from the FETCH_SIZE / WRITE_SIZE passes.
usage: scripts/collect_profiles.py TAG r02"""
import json, os, re, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, rnd = sys.argv[1], sys.argv[2]
src, dst = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles")
names = {"bench.json": "bench.json", "bench_profiled.json": "bench_profiled.json", "kernel_stats.csv": "bench_kernel_stats.csv",
         "cfg5_kernel_stats.csv": "cfg5_kernel_stats.csv", "blocks.log": "blocks.txt", "fftsizes.log": "fft_sizes.txt",
         "cfg5.log": "cfg5.txt", "cfg3.log": "cfg3.txt", "extras.log": "extras.txt", "decim_ab.log": "decim_ab.txt",
         "meter.log": "meter.txt", "meter_kernel_stats.csv": "meter_kernel_stats.csv", "clock_power.txt": "clock_power.txt",
         "bank.log": "bank.txt", "callsize.log": "callsize.txt", "shapes_kernel_stats.csv": "shapes_kernel_stats.csv"}
for p in ("pmc_sq1", "pmc_sq2", "pmc_fetch", "pmc_write", "cfg5_pmc_sq1", "cfg5_pmc_sq2", "cfg5_pmc_fetch", "cfg5_pmc_write"):
    names[p + ".summary.txt"] = p + ".summary.txt"
for a, b in names.items():
    f = os.path.join(src, a)
    if os.path.exists(f):
        text = open(f, errors="replace").read()
        text = "\n".join(l for l in text.splitlines() if "amdgpu.ids" not in l) + "\n"
        open(os.path.join(dst, f"{rnd}_{b}"), "w").write(text)
        print("copied", a, "->", f"{rnd}_{b}")
    else:
        print("missing", a)


def counter(path, kernel, name):
    txt = open(path).read()
    m = re.search(re.escape(kernel) + r"[^\n]*\n((?:    [^\n]*\n)+)", txt)
    if not m:
        return None
    mm = re.search(name + r"\s+avg=([0-9.e+]+)", m.group(1))
    return float(mm.group(1)) if mm else None


f, w = os.path.join(src, "pmc_fetch.summary.txt"), os.path.join(src, "pmc_write.summary.txt")
if os.path.exists(f) and os.path.exists(w):
    # the chain's dominant kernel: the fused frame kernel where it runs, else the wave-per-block FIR stage
    kern = ("k_ols_frame<true" if counter(f, "k_ols_frame<true", "FETCH_SIZE") else
            "k_ols_frame" if counter(f, "k_ols_frame", "FETCH_SIZE") else "k_ols_wave")
    fetch, write = counter(f, kern, "FETCH_SIZE"), counter(w, kern, "WRITE_SIZE")
    if fetch and write:
        n = 1 << 26
        hbm = (2 * fetch + write) * 1024.0
        json.dump({"kernel": "k_ols_frame<true, SW>" if kern == "k_ols_frame<true" else kern, "samples_per_launch": n, "FETCH_SIZE_KB_raw": fetch, "WRITE_SIZE_KB": write,
                   "correction": "FETCH_SIZE x2 (gfx950 reports 1/2 of wide coalesced reads), WRITE_SIZE as is; separate --pmc passes (scripts/gpu_profiles.sh)",
                   "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": 10.0 * n, "ratio": hbm / (10.0 * n),
                   "source": f"profiles/{rnd}_pmc_fetch.summary.txt, profiles/{rnd}_pmc_write.summary.txt (bench.py --profile: chain launches only)",
                   "session": f"gpurun_out/{tag} -> profiles/{rnd}_* (scripts/gpu_profiles.sh, one box for every file of the set)",
                   "box": "a 1-GPU MI355X box of the pool (fresh per gpurun call; hostnames are not stable)"},
                  open(os.path.join(dst, "traffic_fused_fir.json"), "w"), indent=1)
        print("traffic_fused_fir.json:", hbm, "bytes per launch =", hbm / (10.0 * n), "x algorithmic")

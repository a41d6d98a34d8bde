#!/usr/bin/env python3
"""Copies what scripts/gpu_profiles.sh left under gpurun_out/TAG into profiles/ under the round's name and
regenerates profiles/traffic_fused_fir.json from the FETCH_SIZE / WRITE_SIZE passes.
Also writes profiles/RND_summary.txt: the headline figures of the set as text lines (read from bench.json, the rocprofv3 kernel
statistics and the traffic file) - DESIGN.md quotes that file instead of carrying figures of its own.
usage: scripts/collect_profiles.py TAG r02        (scripts/collect_profiles.py --summary r03a: the summary of files already there)"""
import csv, json, os, re, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def write_summary(dst, rnd):
    """RND_summary.txt from RND_bench.json, RND_bench_kernel_stats.csv, RND_cfg5_kernel_stats.csv and traffic_fused_fir.json."""
    out = []
    bj = os.path.join(dst, f"{rnd}_bench.json")
    if os.path.exists(bj):
        d = json.loads([l for l in open(bj) if l.startswith("{")][0])
        r = d["roofline"]
        out.append(f"bench.py: value {d['value']} MSamples/s, {d['ms_per_step']} ms per step of {d['config']['samples_per_step_per_gpu']} samples, "
                   f"{d['steps']} steps, whole chain {d['pct_hbm_roofline_whole_chain']} % of the 10 B/sample roofline")
        out.append(f"bench.py: roofline kernel {r['kernel']}, hipEvent average over the timed launches {r['avg_launch_ms']} ms, achieved {r['achieved']} GB/s, "
                   f"frac {r['frac']}, device copy measured in the run {r.get('measured_copy_GBs')} GB/s")
        g = d.get("general_nco")
        if g:
            out.append(f"bench.py: general_nco value {g['value']} MSamples/s, {g['ms_per_step']} ms per step, kernel {g['avg_launch_ms']} ms, frac {g['frac']}, "
                       f"parity against the block-by-block kernels {g.get('parity_vs_block_by_block', float('nan')):.2e}")
        c = d.get("cpu_baseline")
        if c:
            out.append(f"bench.py: cpu_baseline {c['value']} MSamples/s on {c['cores']} threads, {c['single_thread']['value']} on one ({c.get('cpu_model', '')})")
        if d.get("host_fed"):
            out.append(f"bench.py: host_fed {d['host_fed']['value']} MSamples/s over PCIe (never `value`)")
        if "parity_fused_vs_block_by_block_last_step_rms" in d:
            out.append(f"bench.py: last step against the block-by-block kernels, relative rms {d['parity_fused_vs_block_by_block_last_step_rms']:.2e}")
    for name, label in ((f"{rnd}_bench_kernel_stats.csv", "rocprofv3 (bench.py --profile)"), (f"{rnd}_cfg5_kernel_stats.csv", "rocprofv3 (prof_cfg5.py 400)")):
        f = os.path.join(dst, name)
        if not os.path.exists(f):
            continue
        for row in csv.DictReader(open(f)):
            if ("k_ols_frame" in row["Name"] or "k_filter_blk4096" in row["Name"]) and int(row["Calls"]) >= 20:
                kn = re.sub(r"\(.*", "", row["Name"].replace("void rr::", "").replace("(anonymous namespace)::", ""))
                avg = float(row["AverageNs"])
                bps = 10.0 if "k_ols_frame" in kn else 16.0
                out.append(f"{label}: {kn} {row['Calls']} launches, average {avg:.0f} ns (min {row['MinNs']}, max {row['MaxNs']}) = "
                           f"{bps * (1 << 26) / (avg * 1e-9) / 8e12:.3f} of {bps:.0f} B/sample x 2^26 samples at 8 TB/s")
    tj = os.path.join(dst, "traffic_fused_fir.json")
    if os.path.exists(tj) and rnd[-1].isdigit():
        t = json.load(open(tj))
        out.append(f"PMC passes: {t['kernel']} FETCH_SIZE {t['FETCH_SIZE_KB_raw']:.0f} KB x 2 + WRITE_SIZE {t['WRITE_SIZE_KB']:.0f} KB = {t['hbm_bytes_per_launch']:.0f} bytes "
                   f"per launch = {t['ratio']:.3f} x the algorithmic {t['algorithmic_bytes_per_launch']:.0f}")
    open(os.path.join(dst, f"{rnd}_summary.txt"), "w").write("\n".join(out) + "\n")
    print("wrote", f"{rnd}_summary.txt")


if sys.argv[1] == "--summary":
    write_summary(os.path.join(ROOT, "profiles"), sys.argv[2])
    sys.exit(0)
tag, rnd = sys.argv[1], sys.argv[2]
src, dst = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles")
names = {"bench.json": "bench.json", "bench_profiled.json": "bench_profiled.json", "kernel_stats.csv": "bench_kernel_stats.csv",
         "cfg5_kernel_stats.csv": "cfg5_kernel_stats.csv", "blocks.log": "blocks.txt", "fftsizes.log": "fft_sizes.txt",
         "cfg5.log": "cfg5.txt", "cfg3.log": "cfg3.txt", "extras.log": "extras.txt", "decim_ab.log": "decim_ab.txt",
         "meter.log": "meter.txt", "meter_kernel_stats.csv": "meter_kernel_stats.csv", "clock_power.txt": "clock_power.txt",
         "bank.log": "bank.txt", "callsize.log": "callsize.txt", "shapes_kernel_stats.csv": "shapes_kernel_stats.csv",
         "wave2k.log": "wave2k.txt", "bsbig.log": "bluestein_big.txt"}
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
write_summary(dst, rnd)

#!/usr/bin/env python3
"""Shader clock and socket power WHILE a kernel runs (prove or retire "the card clocks
k_filter_blk4096 at 1.5 GHz").

A sampler thread polls whatever the box offers — the amdgpu sysfs/hwmon files (freq1_input, power1_average /
power1_input, pp_dpm_sclk) every 20 ms, and `amd-smi metric` / `rocm-smi` every ~0.5 s as a cross-check — while
the main thread keeps ONE workload on the card for `--seconds`:
   cfg5   the 1024-tap Filter (k_filter_blk4096), 2^26 samples per call
   chain  the cfg2 chain (k_ols_frame), 2^26 samples per call
   copy   a device-to-device copy of 512 MiB (the memory system alone)
   idle   nothing
For every workload: calls per second (= ms per call), and min / median / max of each sampled quantity.
usage: python scripts/clock_probe.py [--seconds 4] [cfg5 chain copy idle ...]
"""
import glob
import json
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def sysfs_sources():
    src = {}
    for dev in sorted(glob.glob("/sys/class/drm/card*/device")):
        for pat, name in (("hwmon/hwmon*/freq1_input", "sclk_hz"), ("hwmon/hwmon*/freq2_input", "mclk_hz"),
                          ("hwmon/hwmon*/power1_average", "power_uW"), ("hwmon/hwmon*/power1_input", "power_in_uW"),
                          ("hwmon/hwmon*/temp1_input", "temp_mC"), ("hwmon/hwmon*/temp2_input", "temp2_mC")):
            for f in glob.glob(os.path.join(dev, pat)):
                try:
                    open(f).read()
                    src.setdefault(f"{os.path.basename(os.path.dirname(dev))}:{name}", f)
                except OSError:
                    pass
        f = os.path.join(dev, "pp_dpm_sclk")
        if os.path.exists(f):
            src[f"{os.path.basename(os.path.dirname(dev))}:pp_dpm_sclk"] = f
    return src


def read_src(name, path):
    try:
        s = open(path).read()
    except OSError:
        return None
    if name.endswith("pp_dpm_sclk"):
        for line in s.splitlines():
            if line.rstrip().endswith("*"):
                return float("".join(c for c in line.split(":")[1] if c.isdigit() or c == "."))  # MHz
        return None
    try:
        return float(s.strip())
    except ValueError:
        return None


def smi_sample():
    """One amd-smi / rocm-smi reading as a flat dict of numbers (best effort)."""
    out = {}
    for cmd in (["amd-smi", "metric", "-g", "0", "--clock", "--power", "--usage", "--json"],
                ["rocm-smi", "-d", "0", "--showclocks", "--showpower", "--showuse", "--json"]):
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=20)
        except (OSError, subprocess.TimeoutExpired):
            continue
        if r.returncode != 0 or not r.stdout.strip():
            continue
        try:
            doc = json.loads(r.stdout[r.stdout.index("[") if r.stdout.lstrip().startswith("[") else r.stdout.index("{"):])
        except ValueError:
            continue

        def walk(prefix, node):
            if isinstance(node, dict):
                if "value" in node and isinstance(node["value"], (int, float)):
                    out[prefix] = float(node["value"])
                    return
                for k, v in node.items():
                    walk(f"{prefix}.{k}" if prefix else str(k), v)
            elif isinstance(node, list):
                for i, v in enumerate(node):
                    walk(f"{prefix}[{i}]", v)
            elif isinstance(node, (int, float)) and not isinstance(node, bool):
                out[prefix] = float(node)
            elif isinstance(node, str):
                t = node.strip().lstrip("(").rstrip(")")
                for suf in ("Mhz", "MHz", "W", "%"):
                    if t.endswith(suf):
                        t = t[: -len(suf)]
                try:
                    out[prefix] = float(t)
                except ValueError:
                    pass

        walk(cmd[0], doc)
    keep = {}
    for k, v in out.items():
        kl = k.lower()
        if any(s in kl for s in ("gfx", "sclk", "power", "socket", "usage", "busy", "mclk", "mem_0")) and "limit" not in kl and "max" not in kl.split(".")[-1] and "min" not in kl.split(".")[-1]:
            keep[k] = v
    return keep


class Sampler(threading.Thread):
    def __init__(self, src, use_smi):
        super().__init__(daemon=True)
        self.src, self.use_smi = src, use_smi
        self.stop_flag = False
        self.data = {}

    def run(self):
        last_smi = 0.0
        while not self.stop_flag:
            for name, path in self.src.items():
                v = read_src(name, path)
                if v is not None:
                    self.data.setdefault(name, []).append(v)
            now = time.perf_counter()
            if self.use_smi and now - last_smi > 0.4:
                for k, v in smi_sample().items():
                    self.data.setdefault(k, []).append(v)
                last_smi = time.perf_counter()
            time.sleep(0.02)


def main():
    import torch
    import radiorust_amd as rr

    args = [a for a in sys.argv[1:]]
    seconds = 4.0
    if "--seconds" in args:
        i = args.index("--seconds")
        seconds = float(args[i + 1])
        del args[i : i + 2]
    loads = args or ["idle", "cfg5", "chain", "copy", "cfg5"]
    src = sysfs_sources()
    print("sysfs sources:", json.dumps(src, indent=1))
    first = smi_sample()
    print("smi keys:", sorted(first))
    N = 1 << 26
    st = torch.cuda.current_stream().cuda_stream
    d_in = torch.empty(N, dtype=torch.complex64, device="cuda")
    rr.synth_iq_dev(0, st, 1, 0, N, d_in.data_ptr())
    d_out = torch.empty(N, dtype=torch.complex64, device="cuda")
    torch.cuda.synchronize()
    f = rr.Filter.new(lambda b, fr: 1.0 if abs(fr) <= 200e6 else 0.0)
    f.set_stream(st)
    g = rr.Chain(shift=25e6, filter_len=64, freq_resp=lambda b, fr: 1.0 if abs(fr) <= 20e6 else 0.0, output_rate=50e6,
                 bandwidth=40e6, fft_len=4096, fft_window=rr.Kaiser.with_null_at_bin(2.0))
    g.set_stream(st)

    def one(load):
        if load == "cfg5":
            f.process_dev(2e9, 1024, d_in.data_ptr(), N, d_out.data_ptr(), N)
        elif load == "chain":
            g.process_dev(200e6, d_in.data_ptr(), N, d_out.data_ptr(), N)
        elif load == "copy":
            d_out.copy_(d_in)
        else:
            time.sleep(0.01)

    report = {}
    for load in loads:
        for _ in range(3):
            one(load)
        torch.cuda.synchronize()
        smp = Sampler(src, bool(first))
        smp.start()
        t0 = time.perf_counter()
        calls = 0
        while time.perf_counter() - t0 < seconds:
            for _ in range(8):
                one(load)
            calls += 8
            torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        smp.stop_flag = True
        smp.join(timeout=30)
        row = {"ms_per_call": 1e3 * dt / calls, "calls": calls}
        for k, v in sorted(smp.data.items()):
            v = sorted(v)
            row[k] = {"n": len(v), "min": v[0], "med": v[len(v) // 2], "max": v[-1]}
        report[load + f"@{len(report)}"] = row
        print(load, json.dumps(row))
        time.sleep(0.5)
    print("CLOCK_PROBE_JSON " + json.dumps(report))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Benchmark of the IQ hot path on MI355X — BASELINE.json's metric:

    MSamples/s (complex IQ) through shift -> FIR -> decimate -> FFT; % HBM roofline

Workload (BASELINE configs[1], SURVEY §8(d) "cfg2"): one channel at 200 MS/s,
FreqShifter 25 MHz (1 Hz precision) -> 64-tap low-pass Filter (|f| <= 20 MHz,
Kaiser null-at-bin 2) -> Downsampler(4096, 50 MS/s, 40 MHz) (L = 120, 4x) ->
Fourier 4096-pt, Kaiser null-at-bin 2.  A step = one pass of the chain over one
batch of 2^26 synthetic complex samples already resident in HBM (the stream
continues from step to step, so every step is steady state).

  python bench.py --gpus N --steps K --warmup W      (defaults: N = 1, K = 400, W = 20; ~1 s of GPU time)

Before the W warm-up steps the chain runs for --settle-ms (100 ms) so that the GPU's power management
has reached its steady state: measured on MI355X, the first ~10 steps after an idle gap run at 0.201
ms, steps 10-40 at 0.22-0.23 ms, and from ~100 steps on 0.193-0.205 ms; a streaming workload lives in
the last regime.

N > 1: one rank per GPU, one independent IQ channel per rank (seed = rank + 1).  Launched either by
torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE in the environment) or bare -- `python bench.py
--gpus N` without WORLD_SIZE starts the N ranks itself as child processes, BEFORE anything in this
process touches the GPU, relays rank 0's JSON line and exits non-zero if any rank fails.  The path shards by channel: there is NO
data-path collective; torch.distributed (RCCL) is used only for the barriers and
the max-over-ranks of the elapsed time.  scaling = weak.

The JSON line also carries
  roofline     — dominant kernel: algorithmic bytes (10 B per input sample:
                 8 B read + 8 B / 4 written) / its average
                 duration from hipEvents recorded by the library around each
                 launch on the stream it runs on (--time-every N samples one
                 launch in N instead), against the 8 TB/s HBM peak.
  cpu_baseline — the CPU oracle (C restatement of the reference blocks, kind = "port")
                 timed on this host on a bounded sample: one thread per block (4 cores)
                 and single-threaded.
  host_fed     — NOT `value`: after the timed region every rank feeds its chain from page-locked host
                 buffers (rr_chain_enqueue + rr_wait), all ranks at once: the PCIe-inclusive rate of
                 the boundary's host-pointer entry points (what N GPUs behind one host's
                 links deliver together).  host_placement: each rank restricts itself to the CPUs of
                 its GPU's NUMA node before it allocates anything (radiorust_amd/dist.py::pin_to_gpu_numa).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ALG_BYTES_PER_SAMPLE = 10.0  # 8 B read + 8 B / D written at D = 4
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def lowpass20(_bin, f):
    return 1.0 if abs(f) <= 20e6 else 0.0


def cpu_baseline(budget_s: float = 6.0):
    """The CPU oracle (C restatement of the reference blocks, chunk by chunk) timed on this host's cores on
    bounded samples of BASELINE's CPU-runnable workloads: cfg2 (the benchmark's chain; `value`:
    one thread per block with capacity-1 hand-off — the parallelism tokio gives the reference, SURVEY 8(d) — and
    single-threaded), cfg1 (FreqShifter + 4096-tap Filter at 48 kS/s) and cfg5 (1024-tap Filter at 2 GS/s).
    Timing build: the oracle's sources compiled on this host with the fastest of a few gcc flag sets
    (`-O3 -march=native ..`, contraction allowed); the parity build (-ffp-contract=off) is only used for the
    spot check of the GPU's first spectrum."""
    import numpy as np

    from oracle import rr_oracle as o

    kw = dict(shift=25e6, filter_len=64, freq_resp=lowpass20, output_rate=50e6, bandwidth=40e6, fft_len=4096,
              fft_window=o.Kaiser.with_null_at_bin(2.0), flt=np.float32, max_frames=1)
    _path, flags, flag_rates = o.pick_native()
    probe = 1 << 20
    x = o.synth_iq(1, 0, probe)
    out, configs = {}, {}
    with o.native():
        res = {}
        for threads in (4, 1):
            t = time.perf_counter()
            o.run_chain_c(x, 200e6, threads=threads, **kw)
            rate = probe / (time.perf_counter() - t)
            n = int(min(max(rate * budget_s, probe), 1 << 28))
            res[threads] = max(probe, n // probe * probe)
        x = o.synth_iq(1, 0, max(res.values()))
        for threads in (4, 1):
            xs = x[: res[threads]]
            t = time.perf_counter()
            _, frames = o.run_chain_c(xs, 200e6, threads=threads, **kw)
            dt = time.perf_counter() - t
            out[threads] = (len(xs) / dt / 1e6, len(xs), frames, dt)

        def time_blocks(blocks, fs, chunk, budget):
            """chunks of `chunk` samples through the blocks in turn on one thread, for about `budget` seconds"""
            n_done, t0 = 0, time.perf_counter()
            pos = 0
            while True:
                c = x[pos : pos + chunk]
                pos = (pos + chunk) % (len(x) - chunk)
                for b in blocks:
                    c = b(fs, c)
                    if c is None:
                        break
                n_done += chunk
                dt = time.perf_counter() - t0
                if dt >= budget:
                    return n_done / dt / 1e6, n_done, dt

        # cfg1: FreqShifter 700 Hz -> Filter |f| <= 16 kHz in chunks of 4096 at 48 kS/s
        fs1 = o.FreqShifter(1.0, 700.0, flt=np.float32)
        fl1 = o.Filter(lambda _b, f: 1.0 if abs(f) <= 16e3 else 0.0, flt=np.float32)
        v, nn, dt = time_blocks([fs1.process, fl1.process], 48000.0, 4096, budget_s / 2)
        configs["cfg1"] = {"value": round(v, 3), "unit": "MSamples/s", "cores": 1,
                           "sample": f"{nn} samples in chunks of 4096 (FreqShifter + 4096-tap Filter, 48 kS/s), {dt:.1f} s"}
        # cfg5: Filter n = 1024, |f| <= 200 MHz at 2 GS/s
        fl5 = o.Filter(lambda _b, f: 1.0 if abs(f) <= 200e6 else 0.0, flt=np.float32)
        v, nn, dt = time_blocks([fl5.process], 2e9, 1024, budget_s / 2)
        configs["cfg5"] = {"value": round(v, 3), "unit": "MSamples/s", "cores": 1,
                           "sample": f"{nn} samples in chunks of 1024 (1024-tap Filter, 2 GS/s), {dt:.1f} s"}
    v4, n4, f4, d4 = out[4]
    v1, n1, f1, d1 = out[1]
    configs["cfg2"] = {"value": round(v4, 3), "unit": "MSamples/s", "cores": 4, "single_thread": round(v1, 3)}
    head = 4096 * 4 + 64 + 4  # inputs the first spectrum depends on
    kw64 = dict(kw, flt=np.float64)
    kw64.pop("max_frames")
    ref0 = o.run_chain_c(o.synth_iq(1, 0, head + 64), 200e6, **kw64)[0][0]  # parity build
    return {
        "value": round(v4, 3),
        "unit": "MSamples/s",
        "cores": 4,
        "kind": "port",
        "sample": f"{n4} complex samples of the same cfg2 stream ({f4} spectra), {d4:.1f} s, C restatement of the "
                  f"reference blocks chunk by chunk (gcc {flags}, built on this host), one thread per block, capacity-1 hand-off",
        "single_thread": {"value": round(v1, 3), "cores": 1, "sample": f"{n1} samples ({f1} spectra), {d1:.1f} s"},
        "configs": configs,
        "cpu_model": o.cpu_model(),
        "nproc": os.cpu_count(),
        "host_cores_available": os.cpu_count(),
        "build": {"chosen": flags, "probe_MSamples_s_single_thread": flag_rates},
    }, ref0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--samples", type=int, default=1 << 26, help="complex samples per step per GPU")
    ap.add_argument("--settle-ms", type=float, default=100.0, help="GPU load before the warm-up steps, for steady clocks")
    ap.add_argument("--no-fused", action="store_true", help="force the block-by-block kernels")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget-s", type=float, default=6.0, help="seconds of CPU work per timed leg of the cpu_baseline")
    ap.add_argument("--time-all", action="store_true", help="time every kernel inside the timed region, not only the dominant one")
    ap.add_argument("--time-every", type=int, default=1,
                    help="inside the timed region one launch in this many of the dominant kernel records its start / end.  "
                         "Measured: timing every launch costs 1 %% of `value` (0.1676 against 0.1659 ms per step), but a launch timed "
                         "between untimed neighbours also waits for its predecessor's tail (0.1666 against 0.1619 ms) and so no "
                         "longer agrees with the profiler's isolated duration: the default stays 1")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="rank logic rehearsal only: all ranks share cuda:0 and line up over gloo (RCCL refuses two ranks on one GPU)")
    ap.add_argument("--traffic-json", default=None, help="file holding measured HBM bytes per launch (PMC pass)")
    ap.add_argument("--no-host-fed", action="store_true", help="skip the un-timed host-buffer (PCIe-inclusive) leg")
    ap.add_argument("--no-general-nco", action="store_true", help="skip the un-timed leg with a 40 000-entry phase table")
    ap.add_argument("--no-numa-pin", action="store_true", help="leave the rank's CPU affinity alone")
    ap.add_argument("--profile", action="store_true",
                    help="profiler runs: only the chain's own launches (no block-by-block replay, no copy benchmark, no extra "
                         "timing steps, no CPU baseline), so that a rocprofv3 --kernel-trace --stats of this command averages "
                         "exactly the launches the line reports")
    args = ap.parse_args()
    if args.profile:
        args.no_cpu_baseline = True

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # the driver's form `python bench.py --gpus N`: this process only starts the ranks (fresh child
        # processes, nothing here has touched torch.cuda or HIP) and passes rank 0's line on
        from radiorust_amd.dist import spawn_ranks

        raise SystemExit(spawn_ranks([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], args.gpus))

    import numpy as np
    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the backend has no CPU path")
    from radiorust_amd.dist import Ranks, whole_job_rate

    if args.rehearse_on_one_gpu:
        torch.cuda.set_device(0)
        ranks = Ranks("gloo")
        ranks.local_rank = 0
    else:
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        ranks = Ranks("nccl")
    world, rank, local_rank = ranks.world, ranks.rank, ranks.local_rank
    if args.gpus != world and rank == 0:
        print(f"note: --gpus {args.gpus} but WORLD_SIZE={world}; using {world} rank(s)", file=sys.stderr)

    import radiorust_amd as rr

    rr._lib.lib()
    # the feeder of a GPU (this rank) and its pinned buffers live on the GPU's NUMA node
    host_info = {"pinned": False}
    if not args.no_numa_pin:
        import ctypes as C0

        from radiorust_amd.dist import pin_to_gpu_numa

        buf = C0.create_string_buffer(64)
        if rr._lib.lib().rr_device_pci_bus_id(local_rank, buf, 64) == 0:
            host_info = pin_to_gpu_numa(buf.value.decode())
    fs, n = 200e6, int(args.samples)
    stream = torch.cuda.current_stream().cuda_stream
    chain = rr.Chain(shift=25e6, filter_len=64, freq_resp=lowpass20, output_rate=50e6, bandwidth=40e6, fft_len=4096,
                     fft_window=rr.Kaiser.with_null_at_bin(2.0), device=local_rank, allow_fused=not args.no_fused)
    chain.set_stream(stream)
    d_in = torch.empty(n, dtype=torch.complex64, device="cuda")
    rr.synth_iq_dev(local_rank, stream, ranks.channel_seed(), 0, n, d_in.data_ptr())  # one channel per rank
    cap = (n // 4 // 4096 + 2) * 4096
    d_out = torch.empty(cap, dtype=torch.complex64, device="cuda")

    def step():
        return chain.process_dev(fs, d_in.data_ptr(), n, d_out.data_ptr(), cap)

    # the very first output frame is kept for the parity spot check of the cpu_baseline leg (not timed)
    first_frames = step() // 4096
    torch.cuda.synchronize()
    first_spectrum = d_out[:4096].cpu().numpy().astype(np.complex128) if rank == 0 else None
    # Clock settle (not a step count of the contract, not timed): after an idle gap
    # the power management needs ~50 ms of load to reach its steady state (measured: 0.201 ms/step over the
    # first 10 steps, 0.223-0.231 over steps 10-40, 0.201-0.205 from ~100 steps on).  A streaming workload
    # lives in the steady state, so the W warm-up and K timed steps are taken there.
    ref_steps = 1
    settle = max(0, int(args.settle_ms / 0.2))
    for _ in range(settle):
        step()
    ref_steps += settle
    torch.cuda.synchronize()
    for _ in range(max(args.warmup - 1, 0)):
        step()

    lib = rr._lib.lib()
    import ctypes as C

    # inside the timed region only the dominant kernel is timed (a timed launch costs ~5 us of stream
    # time); the other kernels' averages come from a few extra steps after the timed region
    lib.rr_chain_timing_enable(chain._h, 2 if not (args.no_fused or args.time_all) else 1)
    lib.rr_chain_timing_every(chain._h, max(1, args.time_every) if not (args.no_fused or args.time_all) else 1)
    lib.rr_chain_timing_reset(chain._h)

    barrier = ranks.barrier
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    frames = 0
    for _ in range(args.steps):
        frames += step() // 4096
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    fused = chain.last_path_fused()
    fused_kernel = chain.last_path_kernel()
    per_rank_elapsed = ranks.gather_over_ranks(elapsed)
    elapsed = ranks.max_over_ranks(elapsed)

    # General-NCO leg (never `value`): the headline's shift is fs / 8, an NCO period of 8, which lets
    # the library fold the mixer into the response tables (the device then does nothing for it).  The same chain with
    # SURVEY 8(a1)'s stress shift - 12.345 MHz at 1 kHz precision: ratio 2469 / 40000, a 40 000-entry table - runs the
    # kernel instance with the mixer inside; the same K steps, bracketed the same way, reported beside `value`.
    general = None
    if not args.no_fused and not args.profile and not args.no_general_nco:
        gch = rr.Chain(shift=12.345e6, precision=1e3, filter_len=64, freq_resp=lowpass20, output_rate=50e6, bandwidth=40e6,
                       fft_len=4096, fft_window=rr.Kaiser.with_null_at_bin(2.0), device=local_rank)
        gch.set_stream(stream)
        d_out_g = torch.empty(cap, dtype=torch.complex64, device="cuda")  # (d_out keeps the headline chain's last spectra for the replay check)
        for _ in range(max(args.warmup, 3)):
            gch.process_dev(fs, d_in.data_ptr(), n, d_out_g.data_ptr(), cap)
        lib.rr_chain_timing_enable(gch._h, 2)
        lib.rr_chain_timing_every(gch._h, 1)
        lib.rr_chain_timing_reset(gch._h)
        barrier()
        torch.cuda.synchronize()
        tg = time.perf_counter()
        for _ in range(args.steps):
            gch.process_dev(fs, d_in.data_ptr(), n, d_out_g.data_ptr(), cap)
        torch.cuda.synchronize()
        barrier()
        g_el = ranks.max_over_ranks(time.perf_counter() - tg)
        g_ms, g_cnt, i = 0.0, 0, 0
        while lib.rr_chain_timing_stage_name(i):
            ms, cnt = C.c_double(), C.c_uint64()
            rr._lib.check(lib.rr_chain_timing_read(gch._h, i, C.byref(ms), C.byref(cnt)))
            if cnt.value and ms.value / cnt.value > g_ms:
                g_ms, g_cnt = ms.value / cnt.value, cnt.value
            i += 1
        general = {
            "value": round(whole_job_rate(n, args.steps, world, g_el), 1),
            "unit": "MSamples/s",
            "ms_per_step": round(g_el / args.steps * 1e3, 4),
            "frac": round(ALG_BYTES_PER_SAMPLE * n / (g_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if g_ms else None,
            "avg_launch_ms": round(g_ms, 5),
            "launches": g_cnt,
            "pct_hbm_roofline_whole_chain": round(100.0 * ALG_BYTES_PER_SAMPLE * whole_job_rate(n, args.steps, world, g_el) * 1e6
                                                  / world / (HBM_PEAK_GBS * 1e9), 3),
            "kernel": gch.last_path_kernel(),
            "mixer": ("behind the filter: the blocks transform the samples as they are with the tables of c[i] w^-i, each result times "
                      "the phase table's entry at its position (k_ols_frame<.., GP>: any NCO period)"
                      if gch.last_path_mixer_folded() else "in front of the block transform, in the kernel"),
            "shift": "12.345 MHz at 1 kHz precision: 2469 / 40000 of fs, a 40 000-entry phase table",
        }
        if rank == 0:
            # the same calls through the block-by-block kernels (FreqShifter, Filter, Downsampler, Fourier one at a time)
            gref = rr.Chain(shift=12.345e6, precision=1e3, filter_len=64, freq_resp=lowpass20, output_rate=50e6, bandwidth=40e6,
                            fft_len=4096, fft_window=rr.Kaiser.with_null_at_bin(2.0), device=local_rank, allow_fused=False)
            gref.set_stream(stream)
            d_ref_g = torch.empty(cap, dtype=torch.complex64, device="cuda")
            wrote = 0
            for _ in range(max(args.warmup, 3) + args.steps):
                wrote = gref.process_dev(fs, d_in.data_ptr(), n, d_ref_g.data_ptr(), cap)
            torch.cuda.synchronize()
            a, b = d_out_g[:wrote], d_ref_g[:wrote]
            general["parity_vs_block_by_block"] = float((torch.linalg.vector_norm(a - b) / torch.linalg.vector_norm(b)).item())
            del gref, d_ref_g
        del gch, d_out_g

    # Full-size consistency check (not timed): replay the same calls through the block-by-block
    # kernels and compare the last step's spectra; together with the first-spectrum check against
    # the f64 oracle above this ties the fused kernels to the oracle at the benchmark size.
    fused_vs_blocks = None
    if rank == 0 and fused and not args.no_fused and not args.profile:
        ref_chain = rr.Chain(shift=25e6, filter_len=64, freq_resp=lowpass20, output_rate=50e6, bandwidth=40e6,
                             fft_len=4096, fft_window=rr.Kaiser.with_null_at_bin(2.0), device=local_rank,
                             allow_fused=False)
        ref_chain.set_stream(stream)
        d_ref = torch.empty(cap, dtype=torch.complex64, device="cuda")
        wrote = 0
        for _ in range(ref_steps + max(args.warmup - 1, 0) + args.steps):
            wrote = ref_chain.process_dev(fs, d_in.data_ptr(), n, d_ref.data_ptr(), cap)
        torch.cuda.synchronize()
        a, b = d_out[:wrote], d_ref[:wrote]
        fused_vs_blocks = float((torch.linalg.vector_norm(a - b) / torch.linalg.vector_norm(b)).item())
        del ref_chain, d_ref

    # the "measured-copy" denominator next to the 8 TB/s spec -- a device-to-device
    # copy of 1 GiB (read + write counted), median of 10, timed with events on the same stream
    copy_gbs = None
    if rank == 0 and not args.profile:
        a = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
        b = torch.empty_like(a)
        b.copy_(a)
        ts = []
        for _ in range(10):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            b.copy_(a)
            e1.record()
            e1.synchronize()
            ts.append(e0.elapsed_time(e1))
        copy_gbs = 2 * a.numel() * 4 / (sorted(ts)[len(ts) // 2] * 1e-3) / 1e9
        del a, b

    # per-kernel device time from the library's hipEvents
    def read_stages():
        out, i = {}, 0
        while True:
            name = lib.rr_chain_timing_stage_name(i)
            if not name:
                return out
            ms, cnt = C.c_double(), C.c_uint64()
            rr._lib.check(lib.rr_chain_timing_read(chain._h, i, C.byref(ms), C.byref(cnt)))
            if cnt.value:
                out[name.decode()] = {"launches": cnt.value, "avg_ms": ms.value / cnt.value}
            i += 1

    timed_stages = read_stages()
    other = {}
    if not args.no_fused and not args.profile:  # the remaining kernels, outside the timed region
        lib.rr_chain_timing_enable(chain._h, 1)
        lib.rr_chain_timing_every(chain._h, 1)
        lib.rr_chain_timing_reset(chain._h)
        for _ in range(20):
            step()
        torch.cuda.synchronize()
        other = {k: v for k, v in read_stages().items() if k not in timed_stages}
    stages = {}
    stages.update(timed_stages)

    # Host-fed leg (never `value`; not part of the timed region): every rank feeds its chain from page-locked host
    # buffers through rr_chain_enqueue + rr_wait, all ranks at the same time - the PCIe-inclusive rate of the
    # boundary's host-pointer entry points, which is what N GPUs behind one host's links deliver together.
    host_fed = None
    if not args.profile and not args.no_host_fed:
        hn = min(n, 1 << 24)
        hcap = hn // 4 + 8192
        p_in, p_out = C.c_void_p(), C.c_void_p()
        rr._lib.check(lib.rr_host_alloc(hn * 8, C.byref(p_in)))
        rr._lib.check(lib.rr_host_alloc(hcap * 8, C.byref(p_out)))
        h_src = d_in[:hn].cpu().numpy()  # (kept alive across the copy)
        C.memmove(p_in, h_src.ctypes.data, hn * 8)
        del h_src
        hchain = rr.Chain(shift=25e6, filter_len=64, freq_resp=lowpass20, output_rate=50e6, bandwidth=40e6, fft_len=4096,
                          fft_window=rr.Kaiser.with_null_at_bin(2.0), device=local_rank)
        hchain._ensure_design(fs)
        cnt = C.c_size_t()
        for _ in range(3):
            rr._lib.check(lib.rr_chain_enqueue(hchain._h, fs, p_in, hn, p_out, hcap, C.byref(cnt)))
        hchain.wait()
        hk = 8
        barrier()
        th = time.perf_counter()
        for _ in range(hk):
            rr._lib.check(lib.rr_chain_enqueue(hchain._h, fs, p_in, hn, p_out, hcap, C.byref(cnt)))
        hchain.wait()
        h_el = time.perf_counter() - th
        h_per_rank = ranks.gather_over_ranks(h_el)
        h_max = ranks.max_over_ranks(h_el)
        host_fed = {
            "value": round(hn * hk * world / h_max / 1e6, 1),
            "unit": "MSamples/s",
            "per_rank_MSamples_s": [round(hn * hk / e / 1e6, 1) for e in h_per_rank],
            "link_GB_s_per_rank": round((hn * 8 + cnt.value * 8) * hk / h_max / 1e9, 1),
            "sample": f"{hk} calls of {hn} samples per rank from page-locked host buffers (rr_chain_enqueue + rr_wait), "
                      "all ranks at once; PCIe-inclusive, never `value`",
        }
        del hchain
        lib.rr_host_free(p_in)
        lib.rr_host_free(p_out)
    host_infos = [host_info] if world == 1 else None
    if world > 1:
        # (node per rank, in rank order)
        nodes = ranks.gather_over_ranks(float(host_info.get("numa_node", -1)))
        pins = ranks.gather_over_ranks(1.0 if host_info.get("pinned") else 0.0)
        host_infos = [{"numa_node": int(a), "pinned": bool(b)} for a, b in zip(nodes, pins)]

    if rank == 0:
        value = whole_job_rate(n, args.steps, world, elapsed)
        dom = max(stages, key=lambda k: stages[k]["avg_ms"])
        avg_s = stages[dom]["avg_ms"] * 1e-3
        achieved = ALG_BYTES_PER_SAMPLE * n / avg_s / 1e9
        # HBM bytes per launch of the dominant kernel come from separate rocprofv3 --pmc
        # passes (FETCH_SIZE x2 on gfx950 + WRITE_SIZE; scripts/gpu_pmc.sh), committed
        # under profiles/; they cannot be collected in the same run as the timing.
        traffic, traffic_source = None, None
        tj = args.traffic_json or os.path.join(ROOT, "profiles", "traffic_fused_fir.json")
        if fused and os.path.exists(tj):
            t = json.load(open(tj))
            if int(t.get("samples_per_launch", 0)) == n and str(t.get("kernel", "")).startswith(fused_kernel):
                traffic = t.get("hbm_bytes_per_launch")
                # (not measured in THIS run: PMC passes cannot share a run with the timing; the file says where it was)
                traffic_source = {"file": os.path.relpath(tj, ROOT), "passes": t.get("source"), "session": t.get("session"),
                                  "box": t.get("box"), "ratio_to_algorithmic": t.get("ratio")}
        line = {
            "metric": "MSamples/s (complex IQ) through shift->FIR->decimate->FFT chain; % HBM roofline",
            "value": round(value, 1),
            "unit": "MSamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "cfg2: 1 channel/GPU @200 MS/s, FreqShifter 25 MHz -> Filter 64-tap LP 20 MHz -> "
                            "Downsampler 4x (L=120) -> Fourier 4096 Kaiser(null@2)",
                "samples_per_step_per_gpu": n,
                "spectra_per_step_per_gpu": frames // max(args.steps, 1),
                "path": "fused" if fused else "block-by-block",
                "sharding": "one independent channel per GPU, no collective",
            },
            "per_rank_MSamples_s": [round(n * args.steps / e / 1e6, 1) for e in per_rank_elapsed],
            "pct_hbm_roofline_whole_chain": round(100.0 * ALG_BYTES_PER_SAMPLE * value * 1e6 / world / (HBM_PEAK_GBS * 1e9), 3),
            "roofline": {
                "bound": "hbm",
                "kernel": (fused_kernel if dom == "fused_mix_fir_decim" and fused_kernel else dom),
                "stage": dom,
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5),
                "traffic": traffic,
                "traffic_source": traffic_source,
                "alg_bytes_per_launch": ALG_BYTES_PER_SAMPLE * n,
                "avg_launch_ms": round(stages[dom]["avg_ms"], 5),
                "measured_copy_GBs": round(copy_gbs, 1) if copy_gbs else None,
                "frac_of_measured_copy": round(achieved / copy_gbs, 5) if copy_gbs else None,
            },
            # (launches = the launches that recorded their start / end: one in --time-every inside the timed region)
            "kernels": {k: {"launches": v["launches"], "avg_ms": round(v["avg_ms"], 5)} for k, v in stages.items()},
            "timed_launch_every": max(1, args.time_every) if not (args.no_fused or args.time_all) else 1,
            "kernels_outside_timed_region": {k: {"launches": v["launches"], "avg_ms": round(v["avg_ms"], 5)}
                                             for k, v in other.items()},
            "general_nco": general,
            "host_fed": host_fed,
            "host_placement": host_infos,
            "parity_first_spectrum_rms": None,  # filled in by the cpu_baseline leg
            "parity_fused_vs_block_by_block_last_step_rms": fused_vs_blocks,
        }
        assert first_frames >= 0
        if args.profile:
            line["profile_run"] = "chain launches only: no replay, no copy benchmark, no extra timing steps"
        if args.rehearse_on_one_gpu:
            line["rehearsal"] = "all ranks shared cuda:0 over gloo: not a measurement"
    barrier()  # rank 0's un-timed GPU checks are done before anybody tears the group down
    ranks.close()
    if rank == 0:
        if not args.no_cpu_baseline:
            # the only place the oracle is used: the CPU baseline, and with it the spot check of the GPU's first spectrum
            # against the f64 oracle run on the same leading samples.  At any world size: it runs on rank 0 AFTER the
            # process group is gone, so with N > 1 the other ranks have left and the host's cores are rank 0's alone.
            line["cpu_baseline"], ref0 = cpu_baseline(args.cpu_budget_s)
            line["parity_first_spectrum_rms"] = float(np.sqrt(np.sum(np.abs(first_spectrum - ref0) ** 2) / np.sum(np.abs(ref0) ** 2)))
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""cfg5 (1024-tap Filter, 2^26 samples per call; CFG5_N / CFG5_F16 change the case) for library variants, one
subprocess per run, interleaved rounds in one GPU session.
usage: bench_cfg5_variants.py [rounds] [variant ..]   variant = "-" (the product) or "name:ENV=value:.."
       e.g. "-" "merge:RR_LIB=scripts/ubench/lib_f4kmerge.so" """
import os, subprocess, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time
sys.path.insert(0, %r)
import torch
import radiorust_amd as rr
n, N, fs = int(os.environ.get("CFG5_N", "1024")), 1 << 26, 2e9
st = torch.cuda.current_stream().cuda_stream
d_in = torch.empty(N, dtype=torch.complex64, device="cuda")
rr.synth_iq_dev(0, st, 1, 0, N, d_in.data_ptr())
f16 = os.environ.get("CFG5_F16", "0") != "0"
d_out = torch.empty(N, dtype=torch.complex64, device="cuda")
f = rr.Filter.new(lambda b, fr: 1.0 if abs(fr) <= 200e6 else 0.0)
f.set_stream(st)
def run():
    if f16:
        f.process_dev_f16(fs, n, d_in.data_ptr(), N, d_out.data_ptr(), N, response_f16=False)
    else:
        f.process_dev(fs, n, d_in.data_ptr(), N, d_out.data_ptr(), N)
for _ in range(30): run()
torch.cuda.synchronize()
K = 50
t = time.perf_counter()
for _ in range(K): run()
torch.cuda.synchronize()
print("MS", (time.perf_counter() - t) / K * 1e3)
''' % ROOT
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
variants = sys.argv[2:] or ["-"]
res = {v: [] for v in variants}
for r in range(rounds):
    for v in variants:
        env = dict(os.environ)
        for kv in v.split(":")[1:]:
            k, _, val = kv.partition("=")
            env[k] = val
        out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=300)
        ms = [float(l.split()[1]) for l in out.stdout.splitlines() if l.startswith("MS")]
        if not ms:
            print(v, "FAILED", out.stderr[-800:]); sys.exit(1)
        res[v].append(ms[0])
B = 12 if os.environ.get("CFG5_F16", "0") != "0" else 16
for v in variants:
    m = statistics.median(res[v])
    print(f"variant {v:24s} med {m:.4f} ms min {min(res[v]):.4f}  = {(1<<26)/m/1e3:.0f} MSamples/s = {100*B*(1<<26)/(m*1e-3)/8e12:.1f} % of {B} B/sample at 8 TB/s   all: {' '.join('%.4f'%x for x in res[v])}")

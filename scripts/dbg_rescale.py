import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import rr_oracle as o
from radiorust_amd import metering as m
x = o.synth_iq(31, 0, 4096)
sp = o.Fourier(o.Kaiser.with_null_at_bin(2.0), flt=np.float32).process(x)
a = m.rescale_energy(300, sp); b = o.rescale_energy(300, sp, np.float32)
bad = np.nonzero(a != b)[0]
print(len(bad), bad[:10], a[bad[:5]], b[bad[:5]], (a[bad[:5]]-b[bad[:5]])/b[bad[:5]])
# pure numpy restatement, f32
def ref(res, inp):
    n = len(inp); out = np.zeros(res, np.float32)
    ns = (inp.real.astype(np.float32)*inp.real.astype(np.float32) + inp.imag.astype(np.float32)*inp.imag.astype(np.float32)).astype(np.float32)
    for oi in range(res):
        left = np.float32(np.float32(oi)/np.float32(res))*np.float32(n)
        right = np.float32(np.float32(np.float32(oi)+np.float32(1))/np.float32(res))*np.float32(n)
        lf = min(int(np.floor(left)), n-1); rc = min(int(np.ceil(right)), n)
        acc = np.float32(0)
        for ii in range(lf, rc):
            lb = max(np.float32(ii), left); rb = min(np.float32(ii)+np.float32(1), right)
            acc = np.float32(acc + np.float32(ns[ii]*np.float32(rb-lb)))
        out[oi] = acc
    return out
r = ref(300, sp)
print("gpu==numpy", np.array_equal(a, r), "oracle==numpy", np.array_equal(b, r))

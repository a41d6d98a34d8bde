"""Static budget of the headline kernels (no GPU needed: hipcc cross-compiles gfx950 to assembly): no register spills - scratch is
HBM traffic, DESIGN.md 4.1 - and the instruction mix the profiles were taken with.  A shared device function that changes under
one kernel changes under all of them; this is where it shows (the frame kernel once lost its LDS reads of G_p that way)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def kernels(src, tmp_path):
    out = tmp_path / (os.path.basename(src) + ".s")
    subprocess.run([HIPCC, "-O3", "-std=c++20", "--offload-arch=gfx950", "--offload-device-only", "-S", os.path.join(ROOT, "radiorust_amd", "csrc", src),
                    "-o", str(out)], check=True, capture_output=True)
    txt = open(out).read()
    res = {}
    # instruction counts per kernel body
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)\.end_amdhsa_kernel", txt, re.S | re.M):
        ops = [l.split()[0] for l in m.group(2).splitlines() if l.strip() and not l.strip().startswith((".", ";", "//")) and not l.strip().endswith(":")]
        res[m.group(1)] = {"lds": sum(o.startswith("ds_") for o in ops),
                           "vmem": sum(o.startswith(("global_", "buffer_", "flat_", "scratch_")) for o in ops),
                           "scratch": sum(o.startswith("scratch_") for o in ops)}
    for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)\.wavefront_size", txt, re.S):
        meta = dict(re.findall(r"\.(vgpr_count|vgpr_spill_count|private_segment_fixed_size):\s+(\d+)", m.group(2)))
        if m.group(1) in res:
            res[m.group(1)].update({k: int(v) for k, v in meta.items()})
    return res


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_the_headline_kernels_do_not_spill(tmp_path):
    ols = kernels("rr_ols.hip", tmp_path)
    frame = {k: v for k, v in kernels("rr_ols_frame.hip", tmp_path).items() if "k_ols_frame" in k}
    assert len(frame) >= 16
    for name, k in frame.items():
        mixer_folded_or_behind = "k_ols_frameILb1E" in name  # MF = true: every instance but the mixer-in-front one
        lf4096 = name.endswith("Li4096EEEvNS_9FrameArgsE")
        if mixer_folded_or_behind and lf4096:
            assert k["vgpr_spill_count"] == 0 and k["private_segment_fixed_size"] == 0 and k["scratch"] == 0, (name, k)
            assert k["vgpr_count"] <= 128, (name, k)
    # the benchmark's instance (fs / 8 folded into the tables, SW, full frames, no metering): the first half of G_p comes from LDS
    bench = [k for n, k in frame.items() if "ILb1ELb1ELb1ELb0ELb0ELi4096E" in n]
    assert len(bench) == 1
    assert bench[0]["lds"] >= 380 and bench[0]["vmem"] <= 216, bench[0]
    for name, k in ols.items():
        if "k_ols_wave" in name or "k_filter_wave" in name:
            assert k["vgpr_spill_count"] == 0, (name, k)
    # 8 : 1 with a wave per 2048-sample block: three waves per SIMD (at four it spilled 58 .. 66 registers)
    w2k = kernels("rr_ols_wave2k.hip", tmp_path)
    assert len(w2k) == 6
    for name, k in w2k.items():
        assert k["vgpr_spill_count"] == 0 and k["private_segment_fixed_size"] == 0 and k["vgpr_count"] <= 168, (name, k)
    # 16 / 32 / 64 : 1 with a workgroup of 4 / 8 / 16 waves per block: no scratch, 16 waves per CU (128 registers)
    wg = kernels("rr_ols_wg.hip", tmp_path)
    assert len(wg) == 24  # 8 workgroup sizes x 3 mixer forms
    for name, k in wg.items():
        assert k["vgpr_spill_count"] == 0 and k["private_segment_fixed_size"] == 0 and k["vgpr_count"] <= 128, (name, k)
    fo = kernels("rr_filter_ols.hip", tmp_path)
    for name, k in fo.items():
        if "k_filter_blk" in name:
            assert k["vgpr_spill_count"] == 0 and k["private_segment_fixed_size"] == 0, (name, k)
    # Complex<f64>: the register transform and the overlap-save kernel on it (two waves per SIMD: at most 256 registers)
    f64 = kernels("rr_f64.hip", tmp_path)
    seen = 0
    for name, k in f64.items():
        if "k_fft4096_f64" in name or "k_ols4096_f64" in name:
            seen += 1
            assert k["vgpr_spill_count"] == 0 and k["vgpr_count"] <= 256, (name, k)
    assert seen == 2

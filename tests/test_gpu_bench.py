"""GPU: the benchmark entry point itself, at a reduced size - the JSON contract of its one line, and the bare
`python bench.py --gpus N` form that starts its own ranks (rehearsed on one GPU over gloo: RCCL refuses two
ranks on one device, and a 1-GPU box is what the tests get)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout  # ONE JSON line on standard output
    return json.loads(lines[0])


def test_bench_line_contract_small():
    d = run_bench("--steps", "3", "--warmup", "1", "--samples", str(1 << 22), "--settle-ms", "0", "--no-cpu-baseline")
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["unit"] == "MSamples/s" and d["dtype"] == "f32"
    assert d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1
    assert d["config"]["path"] == "fused" and len(d["per_rank_MSamples_s"]) == 1
    assert d["parity_fused_vs_block_by_block_last_step_rms"] < 1e-5
    # the un-timed host-fed leg (PCIe-inclusive) and the rank's NUMA placement ride along
    assert d["host_fed"]["value"] > 0 and d["host_fed"]["value"] < d["value"] and len(d["host_placement"]) == 1
    # the general-NCO leg beside `value` (a 40 000-entry phase table: no fold of the mixer into period-8 tables), checked against
    # the block-by-block kernels like the headline
    g = d["general_nco"]
    assert g["value"] > 0 and 0 < g["frac"] < 1 and g["kernel"].startswith("k_ols") and "mixer" in g
    assert g["parity_vs_block_by_block"] < 1e-5


def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher and no WORLD_SIZE: two ranks are started by bench.py itself."""
    d = run_bench("--gpus", "2", "--rehearse-on-one-gpu", "--steps", "3", "--warmup", "1", "--samples", str(1 << 22),
                  "--settle-ms", "0", "--cpu-budget-s", "0.2")
    assert d["n_gpus"] == 2 and len(d["per_rank_MSamples_s"]) == 2
    # with N > 1 rank 0 still reports the CPU baseline (timed after the process group is gone) and the first-spectrum check
    assert d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["kind"] == "port" and d["parity_first_spectrum_rms"] < 1e-5
    assert "rehearsal" in d
    assert len(d["host_fed"]["per_rank_MSamples_s"]) == 2 and len(d["host_placement"]) == 2
    # whole-job value = both ranks' samples over the slower rank's time
    assert d["value"] <= sum(d["per_rank_MSamples_s"]) * 1.001

"""CPU, world_size 2, gloo: the control plane of the N-GPU benchmark.  The data
path has no collective (one independent channel per rank), so what needs
covering is rank discovery, per-rank channel seeds, the barrier and the
max-over-ranks of the elapsed time that `bench.py` reports."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent(
    """
    import json, os, sys, time
    sys.path.insert(0, %r)
    from radiorust_amd.dist import Ranks, whole_job_rate
    r = Ranks("gloo")
    r.barrier()
    elapsed = 0.5 + 0.25 * r.rank            # rank 1 is the slow one
    worst = r.max_over_ranks(elapsed)
    total = r.sum_over_ranks(float(r.channel_seed()))
    r.barrier()
    out = dict(rank=r.rank, world=r.world, seed=r.channel_seed(), worst=worst, total=total,
               rate=whole_job_rate(1000, 10, r.world, worst))
    print("RESULT " + json.dumps(out), flush=True)
    r.close()
    """
) % ROOT


def test_two_ranks_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    procs = []
    for rank in range(2):
        env = dict(os.environ, WORLD_SIZE="2", RANK=str(rank), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT="29533")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    results = []
    for p in procs:
        out, err = p.communicate(timeout=120)
        assert p.returncode == 0, err[-2000:]
        line = [l for l in out.splitlines() if l.startswith("RESULT ")][0]
        results.append(json.loads(line[7:]))
    results.sort(key=lambda d: d["rank"])
    assert [d["seed"] for d in results] == [1, 2]  # one channel per rank
    assert all(d["world"] == 2 for d in results)
    assert all(abs(d["worst"] - 0.75) < 1e-12 for d in results)  # MAX over ranks, seen by both
    assert all(d["total"] == 3.0 for d in results)
    # whole-job rate: both ranks' samples over the slowest rank's time
    assert all(abs(d["rate"] - 2 * 1000 * 10 / 0.75 / 1e6) < 1e-12 for d in results)


def test_single_rank_needs_no_process_group():
    sys.path.insert(0, ROOT)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        os.environ.pop(k, None)
    from radiorust_amd.dist import Ranks

    r = Ranks("gloo")
    assert r.world == 1 and r.rank == 0 and r.dist is None
    r.barrier()
    assert r.max_over_ranks(1.25) == 1.25 and r.channel_seed() == 1

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


SPAWNED = textwrap.dedent(
    """
    import json, os, sys
    sys.path.insert(0, %r)
    from radiorust_amd.dist import Ranks
    r = Ranks("gloo")
    r.barrier()
    rates = r.gather_over_ranks(100.0 * (r.rank + 1))
    if os.environ.get("FAIL_RANK") == str(r.rank):
        sys.exit(7)
    r.barrier()
    if r.rank == 0:
        print(json.dumps({"n_gpus": r.world, "per_rank": rates}), flush=True)
    else:
        print("not the line", flush=True)
    r.close()
    """
) % ROOT

LAUNCHER = textwrap.dedent(
    """
    import sys
    sys.path.insert(0, %r)
    from radiorust_amd.dist import spawn_ranks
    sys.exit(spawn_ranks([sys.executable, sys.argv[1]], int(sys.argv[2]), timeout_s=100))
    """
) % ROOT


def _launch(tmp_path, world, env=None):
    worker, launcher = tmp_path / "w.py", tmp_path / "l.py"
    worker.write_text(SPAWNED)
    launcher.write_text(LAUNCHER)
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run([sys.executable, str(launcher), str(worker), str(world)], env=e, capture_output=True, text=True,
                          timeout=150)


def test_spawn_ranks_relays_rank0_line(tmp_path):
    """What `python bench.py --gpus N` does without a launcher: N child ranks, rank 0's standard output is
    the job's standard output (one JSON line), the other ranks' output goes to standard error."""
    p = _launch(tmp_path, 3)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d == {"n_gpus": 3, "per_rank": [100.0, 200.0, 300.0]}
    assert "not the line" in p.stderr


def test_spawn_ranks_fails_if_any_rank_fails(tmp_path):
    p = _launch(tmp_path, 2, {"FAIL_RANK": "1"})
    assert p.returncode == 7, (p.returncode, p.stderr[-2000:])
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]


def test_bench_spawns_before_touching_the_gpu():
    """bench.py decides to spawn right after parsing its arguments: nothing above that point imports torch
    or the backend (a process that has initialised the GPU must not start or become another one)."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    head = src[: src.index("spawn_ranks([sys.executable")]
    assert "import torch" not in head and "import radiorust_amd" not in head


def test_numa_placement_helpers(tmp_path):
    """a rank restricts itself to the CPUs of its GPU's NUMA node before it allocates pinned
    buffers.  The sysfs walk is exercised on a fabricated tree; the affinity of this process is restored."""
    import os

    from radiorust_amd.dist import cpus_of_numa_node, numa_node_of_pci, parse_cpulist, pin_to_gpu_numa

    assert parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert parse_cpulist("") == []
    dev = tmp_path / "bus/pci/devices/0000:c1:00.0"
    dev.mkdir(parents=True)
    (dev / "numa_node").write_text("1\n")
    allowed = sorted(os.sched_getaffinity(0))
    node = tmp_path / "devices/system/node/node1"
    node.mkdir(parents=True)
    (node / "cpulist").write_text(f"{allowed[0]},4000-4001\n")  # one CPU this process may use, two it does not have
    sysfs = str(tmp_path)
    assert numa_node_of_pci("0000:C1:00.0", sysfs) == 1 and numa_node_of_pci("0000:00:00.0", sysfs) == -1
    assert cpus_of_numa_node(1, sysfs) == [allowed[0], 4000, 4001] and cpus_of_numa_node(7, sysfs) == []
    try:
        info = pin_to_gpu_numa("0000:c1:00.0", sysfs)
        assert info == {"pci_bus_id": "0000:c1:00.0", "numa_node": 1, "pinned": True, "cpus": 1}
        assert os.sched_getaffinity(0) == {allowed[0]}
    finally:
        os.sched_setaffinity(0, allowed)
    # unknown device, or a node none of whose CPUs are allowed: nothing changes
    assert pin_to_gpu_numa("0000:00:00.0", sysfs)["pinned"] is False
    (node / "cpulist").write_text("4000-4001\n")
    assert pin_to_gpu_numa("0000:c1:00.0", sysfs)["pinned"] is False and os.sched_getaffinity(0) == set(allowed)

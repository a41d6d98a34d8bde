"""Control-plane helpers for N independent channels on N GPUs.

The hot path shards by IQ channel (one chain per GPU): there is no
data-path collective.  torch.distributed is used only to line the ranks up
(barrier) and to take the maximum of the per-rank elapsed time — `nccl` (= RCCL)
on the GPU box, `gloo` in the CPU tests.
"""
from __future__ import annotations

import os


class Ranks:
    def __init__(self, backend: str | None = None, device_index: int | None = None):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.dist = None
        self.backend = backend
        if self.world > 1:
            import torch
            import torch.distributed as dist

            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            backend = backend or "nccl"
            kw = {}
            if backend == "nccl":
                kw["device_id"] = torch.device("cuda", self.local_rank if device_index is None else device_index)
            dist.init_process_group(backend, rank=self.rank, world_size=self.world, **kw)
            self.dist = dist
            self.backend = backend

    # one independent IQ channel per rank: seeds 1..N
    def channel_seed(self) -> int:
        return self.rank + 1

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def max_over_ranks(self, value: float) -> float:
        if self.dist is None:
            return float(value)
        import torch

        dev = "cuda" if self.backend == "nccl" else "cpu"
        t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, value: float) -> float:
        if self.dist is None:
            return float(value)
        import torch

        dev = "cuda" if self.backend == "nccl" else "cpu"
        t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def gather_over_ranks(self, value: float) -> list:
        """Every rank's value, in rank order (the per-rank rates next to the aggregate)."""
        if self.dist is None:
            return [float(value)]
        import torch

        dev = "cuda" if self.backend == "nccl" else "cpu"
        t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
        out = [torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return [float(o.item()) for o in out]

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()
            self.dist = None


def parse_cpulist(text: str) -> list:
    """'0-31,64-95' -> [0..31, 64..95] (the format of /sys/devices/system/node/node*/cpulist)."""
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        cpus.extend(range(int(a), int(b or a) + 1))
    return cpus


def numa_node_of_pci(bus_id: str, sysfs: str = "/sys") -> int:
    """NUMA node of a PCI device ("0000:c1:00.0"), -1 when the platform does not say."""
    try:
        return int(open(os.path.join(sysfs, "bus/pci/devices", bus_id.lower(), "numa_node")).read().strip())
    except (OSError, ValueError):
        return -1


def cpus_of_numa_node(node: int, sysfs: str = "/sys") -> list:
    try:
        return parse_cpulist(open(os.path.join(sysfs, "devices/system/node", f"node{node}", "cpulist")).read())
    except (OSError, ValueError):
        return []


def pin_to_gpu_numa(bus_id: str, sysfs: str = "/sys") -> dict:
    """the feeder of a GPU (here: the rank's own process, one per GPU) and the pinned pool it allocates
    afterwards live on the GPU's NUMA node.  Restricts this process to the node's CPUs that it is allowed to use
    (page-locked allocations follow the allocating thread's node) and says what it did; a platform without NUMA
    information, or a node none of whose CPUs are allowed, leaves the affinity alone."""
    info = {"pci_bus_id": bus_id, "numa_node": numa_node_of_pci(bus_id, sysfs), "pinned": False}
    if info["numa_node"] < 0 or not hasattr(os, "sched_setaffinity"):
        return info
    allowed = os.sched_getaffinity(0)
    want = set(cpus_of_numa_node(info["numa_node"], sysfs)) & allowed
    if want:
        os.sched_setaffinity(0, want)
        info["pinned"] = True
        info["cpus"] = len(want)
    return info


def whole_job_rate(samples_per_rank_per_step: int, steps: int, world: int, elapsed_max_s: float) -> float:
    """MSamples/s of the whole job: every rank processed its own channel."""
    return float(world) * samples_per_rank_per_step * steps / elapsed_max_s / 1e6


def free_port() -> int:
    import socket

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def spawn_ranks(argv: list, world: int, timeout_s: float | None = None, extra_env: dict | None = None) -> int:
    """Start `world` copies of the command `argv` (one process per GPU: RANK = LOCAL_RANK = 0 .. world - 1,
    WORLD_SIZE, MASTER_ADDR = 127.0.0.1 and a free MASTER_PORT in their environment), relay rank 0's
    standard output to ours and everybody's standard error to ours, wait for all of them, and return 0
    only if every rank exited with 0 (otherwise the first non-zero exit code; the remaining ranks are
    then ended by PID so that a rank waiting at a barrier does not hang the job).

    The caller must not have touched the GPU: the children are fresh processes started with
    subprocess (no exec of an initialised process, no fork of a CUDA/HIP context)."""
    import subprocess
    import sys
    import threading
    import time

    env0 = dict(os.environ)
    env0.setdefault("MASTER_ADDR", "127.0.0.1")
    env0["MASTER_PORT"] = str(free_port())
    env0["WORLD_SIZE"] = str(world)
    env0.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if extra_env:
        env0.update(extra_env)
    procs = []
    for r in range(world):
        env = dict(env0, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen(list(argv), env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))

    def pump(src, dst, prefix=""):
        for line in src:
            if dst is sys.stdout and not line.lstrip().startswith("{"):
                # library chatter on rank 0's standard output (gloo prints its peer count there): the job's
                # standard output stays the one JSON line
                sys.stderr.write("[rank 0] " + line)
                sys.stderr.flush()
                continue
            dst.write(prefix + line)
            dst.flush()

    threads = []
    for r, p in enumerate(procs):
        # rank 0's stdout carries the JSON line; the other ranks' stdout goes to stderr, labelled
        threads.append(threading.Thread(target=pump, args=(p.stdout, sys.stdout if r == 0 else sys.stderr, "" if r == 0 else f"[rank {r}] "), daemon=True))
        threads.append(threading.Thread(target=pump, args=(p.stderr, sys.stderr, f"[rank {r}] " if world > 1 else ""), daemon=True))
    for t in threads:
        t.start()
    deadline = None if timeout_s is None else time.monotonic() + timeout_s
    rc = 0
    pending = set(range(world))
    while pending:
        for r in sorted(pending):
            code = procs[r].poll()
            if code is not None:
                pending.discard(r)
                if code != 0 and rc == 0:
                    rc = code
        if rc != 0 or (deadline is not None and time.monotonic() > deadline):
            if rc == 0:
                rc = 124
            for r in pending:  # exact PIDs we started
                procs[r].terminate()
            for r in pending:
                try:
                    procs[r].wait(timeout=10)
                except subprocess.TimeoutExpired:
                    procs[r].kill()
            break
        time.sleep(0.05)
    for t in threads:
        t.join(timeout=5)
    return rc

"""Builds and runs the C++ host-layer tests (tests/cpp/test_host_blocks.cpp):
the block API mirror with threads, pinned ChunkBufPool and the capacity-1
broadcast channel.  `--cpu` part runs everywhere, `--gpu` part on the MI355X."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "_build", "test_host_blocks")


def build():
    from oracle import rr_oracle
    from radiorust_amd import build as rbuild

    lib = rbuild.build_library()
    ora = rr_oracle.build()
    src = os.path.join(ROOT, "tests", "cpp", "test_host_blocks.cpp")
    hdr = os.path.join(ROOT, "radiorust_amd", "host", "radiorust_amd.hpp")
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    if os.path.exists(EXE) and all(os.path.getmtime(EXE) >= os.path.getmtime(p) for p in (src, hdr, lib, ora)):
        return EXE
    cmd = ["g++", "-O2", "-std=c++17", "-pthread", src, "-o", EXE, lib, ora,
           f"-Wl,-rpath,{os.path.dirname(lib)}", f"-Wl,-rpath,{os.path.dirname(ora)}", "-Wl,-rpath,/opt/rocm/lib",
           "-L/opt/rocm/lib", "-lamdhip64"]
    subprocess.run(cmd, check=True)
    return EXE


def run(flag):
    exe = build() if not os.path.exists(EXE) or os.environ.get("RR_REBUILD_CPP", "1") == "1" else EXE
    p = subprocess.run([exe, flag], capture_output=True, text=True, timeout=120)
    print(p.stdout)
    print(p.stderr)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    return p.stdout


def test_cpp_host_cpu():
    out = run("--cpu")
    assert "0 failures" in out and "broadcast" in out


@pytest.mark.gpu
def test_cpp_host_gpu():
    out = run("--gpu")
    assert "0 failures" in out and "pipeline_vs_oracle" in out

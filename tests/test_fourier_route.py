"""CPU: which kernels the Fourier block picks for a chunk length (rr_fourier_route - host logic only, the decision
rr_fourier_process takes; analysis.rs:82-115 accepts any length).  The invariants every route must keep, over a sweep of
lengths; no GPU involved."""
import math
import re

import numpy as np
import pytest


@pytest.fixture(scope="module")
def route():
    from radiorust_amd import build

    build.build_library()
    import radiorust_amd.blocks as b

    return b.fourier_route


def is_pow2(n):
    return n & (n - 1) == 0


def smooth(n, primes=(2, 3, 5, 7, 11, 13)):
    for p in primes:
        while n % p == 0:
            n //= p
    return n == 1


LENGTHS = sorted(set(list(range(1, 700)) + [1000, 1001, 1024, 1536, 1999, 2000, 2048, 2049, 3000, 3125, 4004, 4096, 4097, 4800,
                                            5000, 6006, 8000, 8191, 8192, 8193, 10000, 12000, 16384, 20000, 20011, 30375, 32768,
                                            48000, 65536, 77000, 91091, 100000, 131072, 250000, 262144, 262145, 524288, 1 << 20,
                                            1 << 22, 1 << 24, (1 << 23) - 1]))


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_every_route_is_consistent(route, dtype, monkeypatch):
    monkeypatch.delenv("RR_FOURIER_MIXED", raising=False)
    monkeypatch.delenv("RR_FOURIER_GENERIC", raising=False)
    monkeypatch.delenv("RR_FOURIER_BIG", raising=False)
    C = 16 if dtype == np.float32 else 8
    one_image = 8192 if dtype == np.float32 else 4096
    pow2_image = 16384 if dtype == np.float32 else 4096  # (f32: k_fft16384's single image of 16 384 points)
    seen = set()
    for n in LENGTHS:
        r = route(n, dtype)
        seen.add(r.split(" ")[0] + (" " + r.split(" ")[1] if r.startswith(("pow2 ", "mixed two", "bluestein")) else ""))
        if is_pow2(n):
            assert r.startswith("pow2"), (n, r)
            m = re.match(r"pow2 (two passes|five launches|strided) (\d+) x (\d+)$", r)
            if n <= pow2_image:
                assert r == "pow2", (n, r)
            else:
                assert m and int(m.group(2)) * int(m.group(3)) == n, (n, r)
                if m.group(1) == "two passes":
                    assert max(int(m.group(2)), int(m.group(3))) <= 512
            continue
        if r.startswith("mixed two passes"):
            m = re.match(r"mixed two passes (\d+) x (\d+)$", r)
            n1, n2 = int(m.group(1)), int(m.group(2))
            assert n1 * n2 == n and C <= n1 <= 512 and C <= n2 <= 512 and smooth(n) and n > one_image, (n, r)
        elif r.startswith("mixed"):
            rad = [int(x) for x in r.split()[1:]]
            assert rad and math.prod(rad) == n and all(x in (2, 3, 4, 5, 7, 11, 13) for x in rad), (n, r)
            assert rad.count(2) <= 1 and n <= one_image and n >= 32, (n, r)
        elif r.startswith("bluestein"):
            M = int(r.rsplit("=", 1)[1])
            assert is_pow2(M) and M >= 2 * n - 1 and n >= 32, (n, r)
            if "wave" in r:
                assert dtype == np.float32 and M == 1024 and n <= 512
            elif "one kernel" in r:  # k_bluestein4096 (f32, 513 .. 2048 points) or k_bluestein_lds (two LDS images of M elements)
                assert M <= pow2_image and (M < 4 * n or (dtype == np.float32 and M == 4096 and 512 < n <= 2048)), (n, r)
                if M > one_image:  # k_bluestein_big<16384>: f32, 4097 .. 8192 points
                    assert dtype == np.float32 and 4096 < n <= 8192, (n, r)
            else:
                assert M < 4 * n and M > pow2_image, (n, r)  # the smallest power of two that holds the circular convolution
        else:
            assert r == "direct" and n < 32, (n, r)
        # a length the mixed-radix kernels can serve beyond 2048 points never falls back to the five launches
        if n > 2048 and smooth(n) and (n <= one_image or r.startswith("mixed two")):
            assert r.startswith("mixed") or "one kernel" in r, (n, r)  # (f32, 2800 .. 4096 and 6600 .. 8192 points: k_bluestein_big is ahead)
    assert {"pow2", "mixed", "direct"} <= {s.split(" ")[0] for s in seen}


def test_switches(route, monkeypatch):
    monkeypatch.setenv("RR_FOURIER_MIXED", "0")
    assert route(3000) == "bluestein one kernel M=8192" and route(5000) == "bluestein one kernel M=16384"
    monkeypatch.setenv("RR_FOURIER_BS_BIG", "0")
    assert route(5000) == "bluestein four launches M=16384"
    monkeypatch.delenv("RR_FOURIER_BS_BIG")
    assert route(20000) == "bluestein four launches M=65536" and route(200003) == "bluestein many launches M=524288"
    monkeypatch.setenv("RR_FOURIER_MIXED", "2")
    assert route(2000) == "mixed 5 5 5 4 4" and route(1001) == "mixed 13 11 7"
    monkeypatch.delenv("RR_FOURIER_MIXED")
    assert route(2000) == "bluestein one kernel M=4096" and route(1000) == "mixed 5 5 5 4 2"
    assert route(20000) == "mixed two passes 125 x 160" and route(250000) == "mixed two passes 500 x 500"
    # 2049 .. 8192 points in f32: the mixed passes against k_bluestein_big by length
    assert route(2500) == "mixed 5 5 5 5 4" and route(3000) == "bluestein one kernel M=8192" and route(4004) == "bluestein one kernel M=8192"
    assert route(5000).startswith("mixed") and route(8000) == "bluestein one kernel M=16384" and route(3000, np.float64).startswith("mixed")
    assert route(91091) == "bluestein four launches M=262144"  # 7^2 11 13^2: no split into two factors <= 512
    assert route(1 << 16) == "pow2 two passes 256 x 256" and route(1 << 20) == "pow2 five launches 1024 x 1024"
    monkeypatch.setenv("RR_FOURIER_BIG", "transpose")
    assert route(1 << 16) == "pow2 five launches 256 x 256"
    monkeypatch.delenv("RR_FOURIER_BIG")
    monkeypatch.setenv("RR_FOURIER_GENERIC", "1")
    assert route(1 << 16) == "pow2 strided 256 x 256" and route(1000) == "direct"


def test_unsupported_lengths_are_refused(route):
    from radiorust_amd._lib import BackendError as RRError

    for n in (0, (1 << 23) + 1, (1 << 24) + 2):
        with pytest.raises(RRError):
            route(n)
    assert route(1 << 24).startswith("pow2 five launches")

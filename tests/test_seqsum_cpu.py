"""scavislam_amd/csrc/seqsum.h: the parallel evaluation of a SEQUENTIAL float sum (the reference's `float chi2 += res * res`, dense_tracking.cpp:229-262)
must give the bits of the sequential sum.  The header's arithmetic is compiled for the host together with a loop mirror of the workgroup orchestration
(tests/cpp/seqsum_host.cpp) and driven with the tracker's kind of terms and with hostile ones (ties, carries, huge dynamic range, zeros)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    out = tmp_path_factory.mktemp("seqsum") / "libseqsum_host.so"
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-shared", "-fPIC", os.path.join(ROOT, "tests", "cpp", "seqsum_host.cpp"), "-o", str(out)])
    L = C.CDLL(str(out))
    L.svs_host_seq_sum_plain.restype = C.c_float
    L.svs_host_seq_sum_plain.argtypes = [C.c_void_p, C.c_int]
    L.svs_host_seq_sum_emulated.restype = C.c_float
    L.svs_host_seq_sum_emulated.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    return L


def both(L, t, nt=512):
    t = np.ascontiguousarray(t, np.float32)
    st = np.zeros(5, np.int32)
    a = np.float32(L.svs_host_seq_sum_plain(t.ctypes.data, len(t)))
    b = np.float32(L.svs_host_seq_sum_emulated(t.ctypes.data, len(t), nt, st.ctypes.data))
    assert st[4] == 0, "the float-unit form of the term update disagrees with the integer form"
    return a, b, st


def tracker_terms(rng, n, sigma=0.03, zero_frac=0.3):
    res = np.clip(rng.normal(0, sigma, n), -0.1, 0.1).astype(np.float32)
    res[rng.random(n) < zero_frac] = 0
    return res * res


def test_plain_sum_is_numpy_cumsum(lib):
    rng = np.random.default_rng(0)
    t = tracker_terms(rng, 19200)
    a, _, _ = both(lib, t)
    assert a.view(np.uint32) == np.cumsum(t, dtype=np.float32)[-1].view(np.uint32)


@pytest.mark.parametrize("n", [19200, 4800, 1200, 12288, 1, 2, 63, 64, 65, 511, 512, 513, 12345, 230400])
def test_tracker_like_terms(lib, n):
    rng = np.random.default_rng(n)
    slow = []
    for rep in range(60 if n <= 19200 else 6):
        t = tracker_terms(rng, n, sigma=rng.choice([0.003, 0.01, 0.03, 0.08]), zero_frac=rng.choice([0.0, 0.2, 0.7]))
        for nt in (512, 256):
            a, b, st = both(lib, t, nt)
            assert a.view(np.uint32) == b.view(np.uint32), (n, rep, nt, a, b, st)
            assert st[2] == 0, "an assumption of the emulation failed (result came from the fallback)"
            slow.append(st[0])
    if n == 19200:
        # the point of the exercise: the walker adds a few hundred terms the slow way, not 19 200
        assert np.mean(slow) < 1500, np.mean(slow)


def test_ties_and_carries(lib):
    rng = np.random.default_rng(5)
    ties_seen = 0
    for rep in range(300):
        n = int(rng.integers(1, 6000))
        kind = rep % 5
        if kind == 0:      # multiples of a power of two: every other addition is a tie once the accumulator has grown
            t = (rng.integers(0, 64, n) * 2.0 ** rng.integers(-30, -8)).astype(np.float32)
        elif kind == 1:    # constant terms: carries at exact powers of two
            t = np.full(n, rng.choice([0.01, 0.25, 1.0, 3.0]), np.float32)
        elif kind == 2:    # half-ulp terms behind a large head
            t = np.concatenate([[np.float32(2.0 ** rng.integers(-3, 6))], np.full(n, 2.0 ** -rng.integers(20, 30), np.float32)]).astype(np.float32)
        elif kind == 3:    # huge dynamic range (far outside the tracker's)
            t = (10.0 ** rng.uniform(-35, 3, n)).astype(np.float32)
            t[rng.random(n) < 0.3] = 0
        else:              # subnormal and tiny terms, long runs of zeros
            t = np.zeros(n, np.float32)
            idx = rng.integers(0, n, max(1, n // 7))
            t[idx] = (rng.integers(1, 1 << 20, len(idx)).astype(np.uint32)).view(np.float32)
        a, b, st = both(lib, t, int(rng.choice([64, 256, 512])))
        assert a.view(np.uint32) == b.view(np.uint32), (rep, kind, n, a, b, st)
        ties_seen += int(st[3])
    assert ties_seen > 100      # the tie path was exercised


def test_zeros_and_single_terms(lib):
    for t in (np.zeros(0, np.float32), np.zeros(1000, np.float32), np.array([0.0, 0.0, 1e-3], np.float32), np.array([5e-3], np.float32)):
        a, b, st = both(lib, t)
        assert a.view(np.uint32) == b.view(np.uint32)

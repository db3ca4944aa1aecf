"""The camera's photon-noise sampler (rlao_amd/csrc/poisson_alias.hpp): Poisson(lam) = Poisson(32 c) + Poisson(j / 4) + Poisson(d)
with alias tables for the two grid terms and a 7-step inversion for the remainder.

The reference draws `RandomState(wall clock).poisson(frame)` (OOPAO/Detector.py:127-130, 204-206): only the LAW can be matched.
CPU tests: the tables the kernels read reproduce the Poisson probabilities of every grid value to the quantisation of their
thresholds, and a NumPy restatement of the device arithmetic on those very tables samples the right law.  GPU test: the device
sampler itself, over lambdas on both sides of every hand-over (1/4, 32, the table's end / PTRS)."""
import ctypes as C

import numpy as np
import pytest
from scipy import stats

FINE, COARSE_STEP, HEADER = 0.25, 32.0, 4


@pytest.fixture(scope="module")
def table():
    import __graft_entry__ as g
    g.build()
    from rlao_amd import _lib as L
    lib = L.load()
    n = C.c_size_t()
    L.check(lib.aoenv_test_poisson_table(None, 0, C.byref(n)))
    t = np.zeros(n.value, dtype=np.uint32)
    L.check(lib.aoenv_test_poisson_table(t.ctypes.data_as(C.c_void_p), t.size, C.byref(n)))
    return t


def _rows(t):
    nf, nc = int(t[0]), int(t[1])
    for r in range(nf + nc):
        base, d = int(t[HEADER + 2 * r]), int(t[HEADER + 2 * r + 1])
        lam = FINE * r if r < nf else COARSE_STEP * (r - nf)
        yield r, lam, base, d >> 16, d & 0xFFFF


def test_tables_reproduce_the_poisson_probabilities(table):
    t = table
    assert int(t[0]) == 128 and int(t[1]) == 32 and int(t[2]) == t.size and t.size % 4 == 0 and t.size <= 15360
    end = HEADER + 2 * (int(t[0]) + int(t[1]))
    worst = 0.0
    for r, lam, base, kmin, n in _rows(t):
        assert base == end, "rows are stored back to back, in row order (a kernel with less LDS keeps a prefix)"
        assert 1 <= n < 512 and kmin + n < 65536
        ent = t[base:base + n].astype(np.int64)
        thr, alias = (ent >> 9).astype(np.float64) / 2 ** 23, ent & 511
        assert (alias < n).all()
        p = np.zeros(n)
        np.add.at(p, np.arange(n), thr / n)
        np.add.at(p, alias, (1 - thr) / n)
        full = (ent >> 9) == 2 ** 23 - 1                 # a full cell gives itself on both branches
        assert (alias[full] == np.arange(n)[full]).all()
        want = stats.poisson.pmf(kmin + np.arange(n), lam) if lam > 0 else np.array([1.0])
        worst = max(worst, float(np.abs(p - want).max()))
        assert abs(p.sum() - 1) < 1e-12
        # the outcomes left out carry less than 2^-33 of the mass
        if lam > 0:
            assert stats.poisson.cdf(kmin - 1, lam) + stats.poisson.sf(kmin + n - 1, lam) < 2.0 ** -32
        end = base + n
    assert worst < 2.0 ** -22, worst                       # threshold quantisation (2^-24 / n per cell, a few cells per outcome)


def _sample_numpy(t, lam, rs):
    """The device arithmetic of poisson_alias() / alias_draw() in NumPy (float32 where the kernel is), on the same tables."""
    lam = np.asarray(lam, dtype=np.float32)
    wf, wr, wc = (rs.randint(0, 2 ** 32, size=lam.shape, dtype=np.uint64) for _ in range(3))
    c = np.floor(lam * np.float32(1 / 32))
    r = (lam - np.float32(32) * c).astype(np.float32)
    j = np.minimum(np.floor(r * np.float32(4)), 127)
    dl = np.maximum(r - np.float32(0.25) * j, 0).astype(np.float32)

    def draw(row, w):
        base = t[HEADER + 2 * row].astype(np.int64)
        d = t[HEADER + 2 * row + 1].astype(np.int64)
        n, kmin = d & 0xFFFF, d >> 16
        prod = w * n.astype(np.uint64)
        cell, frac = (prod >> np.uint64(32)).astype(np.int64), (prod & np.uint64(0xFFFFFFFF)).astype(np.int64)
        en = t[base + cell].astype(np.int64)
        return kmin + np.where((frac >> 9) < (en >> 9), cell, en & 511)

    k = draw(j.astype(np.int64), wf) + draw(128 + c.astype(np.int64), wc)
    u = ((wr >> np.uint64(9)).astype(np.float32) + np.float32(0.5)) * np.float32(1 / 8388608)
    p = np.exp(-dl).astype(np.float32)
    cdf = p.copy()
    for s in range(7):
        k = k + (u > cdf)
        p = (p * (dl * np.float32(1.0 / (s + 1)))).astype(np.float32)
        cdf = (cdf + p).astype(np.float32)
    return k


def _pit(x, lam, rs):
    """Randomised probability-integral transform: uniform on (0, 1) iff x ~ Poisson(lam), whatever lam each sample has."""
    return stats.poisson.cdf(x - 1, lam) + rs.uniform(size=np.shape(x)) * stats.poisson.pmf(x, lam)


def _assert_uniform(u, what):
    n = u.size
    h = np.histogram(u, bins=64, range=(0, 1))[0]
    chi2 = float(((h - n / 64) ** 2 / (n / 64)).sum())
    ks = float(stats.kstest(u.ravel(), "uniform").statistic) * np.sqrt(n)
    assert chi2 < stats.chi2.ppf(1 - 1e-6, 63), (what, chi2)      # 63 dof: 99.9999 % point = 137
    assert ks < 2.2, (what, ks)                                   # P(sqrt(n) D > 2.2) = 1.2e-4


LAMBDAS = (0.01, 0.2, 0.26, 3.7, 9.99, 31.9, 32.0, 33.3, 100.5, 511.9, 1023.9)


def test_numpy_restatement_of_the_device_sampler_follows_the_poisson_law(table):
    rs = np.random.RandomState(11)
    n = 400_000
    for lam in LAMBDAS:
        x = _sample_numpy(table, np.full(n, lam), rs)
        _assert_uniform(_pit(x, np.float64(np.float32(lam)), rs), lam)
        assert abs(x.mean() - lam) < 5 * np.sqrt(lam / n) + 1e-6
    # a continuum of lambdas, as a camera frame has them
    lam = np.exp(rs.uniform(np.log(0.02), np.log(1000), size=n)).astype(np.float32)
    x = _sample_numpy(table, lam, rs)
    _assert_uniform(_pit(x, lam.astype(np.float64), rs), "continuum")


@pytest.mark.gpu
def test_device_sampler_follows_the_poisson_law_on_both_sides_of_every_hand_over():
    from rlao_amd import _lib as L
    lib = L.load()
    rs = np.random.RandomState(5)
    n = 1 << 19

    def draw(lam, frame, lmax=0.0):
        lam = np.ascontiguousarray(lam, dtype=np.float32)
        out = np.empty(lam.size, dtype=np.float32)
        L.check(lib.aoenv_test_poisson(0, lam.ctypes.data_as(C.c_void_p), lam.size, 1234, frame, lmax, out.ctypes.data_as(C.c_void_p)))
        assert (out == np.floor(out)).all() and out.min() >= 0
        return out.astype(np.float64)

    for i, lam in enumerate(LAMBDAS + (1024.0, 1500.0, 20000.0)):           # from 1024 on: PTRS
        x = draw(np.full(n, lam), frame=i)
        l64 = np.float64(np.float32(lam))
        _assert_uniform(_pit(x, l64, rs), lam)
        assert abs(x.mean() - l64) < 5 * np.sqrt(l64 / n) + 1e-6
        assert abs(x.var() / l64 - 1) < 0.02
    # the PTRS hand-over lowered to 64 photons: lanes of one wave on both sides of it
    lam = np.exp(rs.uniform(np.log(0.02), np.log(4000), size=n)).astype(np.float32)
    for lmax in (0.0, 64.0):
        x = draw(lam, frame=100, lmax=lmax)
        _assert_uniform(_pit(x, lam.astype(np.float64), rs), f"continuum lmax={lmax}")
    # streams: a new frame number is a new draw, the same one replays
    a, b, c = draw(lam, 7), draw(lam, 7), draw(lam, 8)
    assert (a == b).all() and (a != c).mean() > 0.5

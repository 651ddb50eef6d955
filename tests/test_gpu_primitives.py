"""GPU parity of the primitive C-ABI entry points against the C oracle (bit-exact)."""
import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cref():
    from oracle import cref
    return cref


@pytest.mark.parametrize("which", [0, 1])
def test_field_mul(zk_ctx, cref, which):
    r = H.rng(10 + which)
    mod = H.R if which == 0 else H.P
    n = 5000
    a = H.ints_to_array([r.randrange(mod) for _ in range(n)])
    b = H.ints_to_array([r.randrange(mod) for _ in range(n)])
    a[0] = 0
    b[1] = 0
    a[2] = H.ints_to_array([mod - 1])[0]
    b[2] = H.ints_to_array([mod - 1])[0]
    got = zk_ctx.field_mul(which, a, b)
    want = cref.fr_mul(a, b) if which == 0 else cref.fq_mul(a, b)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("log_n", [1, 2, 3, 4, 5, 8, 9, 13])
@pytest.mark.parametrize("inverse,coset", [(0, 0), (1, 0), (0, 1), (1, 1)])
def test_ntt(zk_ctx, cref, log_n, inverse, coset):
    r = H.rng(100 + log_n)
    n = 1 << log_n
    batch = 3 if log_n > 8 else 70   # 70: exercises padding of the batch to 128 lanes
    data = np.stack([H.rand_fr(r, n)[1] for _ in range(batch)])
    want = np.stack([cref.ntt(data[i], log_n, inverse, coset) for i in range(batch)])
    got = data.copy()
    zk_ctx.ntt_batch(got, log_n, batch, inverse, coset)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("log_n", [3, 4, 8, 9, 10, 11, 12, 13, 14, 15])
def test_h(zk_ctx, cref, log_n):
    r = H.rng(200 + log_n)
    n = 1 << log_n
    batch = 5
    a, b, c = (np.stack([H.rand_fr(r, n)[1] for _ in range(batch)]) for _ in range(3))
    want = np.stack([cref.compute_h(a[i], b[i], c[i], log_n) for i in range(batch)])
    out = np.zeros_like(a)
    zk_ctx.h_batch(a, b, c, out, log_n, batch)
    assert np.array_equal(out, want)


@pytest.mark.parametrize("group", [1, 2])
@pytest.mark.parametrize("n,c", [(1, 4), (37, 4), (200, 7), (64, 10),
                                 # 100 + c: one shared table per base, per-window accumulators
                                 (1, 104), (37, 105), (200, 107), (64, 113), (90, 116), (300, 0),
                                 # 200 + k: comb tables over groups of k bases, one-bit windows
                                 (1, 203), (37, 205), (200, 208), (90, 212), (33, 216),
                                 # 300 + k: sign-pattern comb tables (2^(k-1) entries per group,
                                 # the auto plan's layout), incl. a lone base and ragged last groups
                                 (1, 303), (2, 303), (37, 305), (200, 308), (90, 313), (33, 317),
                                 (64, 321)])
def test_msm(zk_ctx, cref, group, n, c):
    r = H.rng(300 + n + group)
    gen = H.g1_gen_mont() if group == 1 else H.g2_gen_mont()
    ks, _ = H.rand_fr(r, n, special=False)
    ks = [k or 1 for k in ks]
    if n > 2:
        ks[1] = ks[0]                         # duplicate base: forces the doubling branch
        ks[2] = (H.R - ks[0]) % H.R           # negated base: forces the infinity branch
    bases = cref.batch_mul(group, gen, H.to_mont_array(ks))
    batch = 5
    sc = np.stack([H.rand_fr(r, n)[1] for _ in range(batch)])
    sc[1] = H.to_mont_array([1] * n)          # all ones: sum of bases
    sc[2] = 0                                 # all zero: infinity
    want = np.stack([cref.msm(group, bases, sc[i], naive=(n <= 40)) for i in range(batch)])
    h = zk_ctx.msm_bases_load(group, bases, n, c)
    out = np.zeros((batch, 8 if group == 1 else 16), dtype=np.uint64)
    zk_ctx.msm_batch(h, sc, batch, out)
    zk_ctx.msm_bases_free(h)
    assert np.array_equal(out, want)
    assert not out[2].any()


@pytest.mark.parametrize("group", [1, 2])
def test_fixed_base_mul(zk_ctx, cref, group):
    r = H.rng(400 + group)
    gen = H.g1_gen_mont() if group == 1 else H.g2_gen_mont()
    n = 300
    _, sc = H.rand_fr(r, n)
    want = cref.batch_mul(group, gen, sc)
    out = np.zeros_like(want)
    zk_ctx.fixed_base_mul(group, gen, sc, n, out)
    assert np.array_equal(out, want)


def test_field_mul_bench_reports(zk_ctx):
    rate = zk_ctx.field_mul_bench(1, 1 << 18, 64)
    assert rate > 1e9


@pytest.mark.parametrize("c", [6, 106, 207, 307])
def test_msm_edge_cases(zk_ctx, cref, c):
    """Infinity among the bases, extreme scalars (0, 1, r-1, r-2, 2^253), batch sizes around the
    64-lane padding (1, 63, 64, 65), and an empty batch -- both table layouts."""
    r = H.rng(555)
    n = 24
    ks = [r.randrange(1, H.R) for _ in range(n)]
    bases = cref.batch_mul(1, H.g1_gen_mont(), H.to_mont_array(ks))
    bases[5] = 0                                  # the point at infinity: contributes nothing
    bases[9] = 0
    h = zk_ctx.msm_bases_load(1, bases, n, c)
    extreme = [0, 1, H.R - 1, H.R - 2, 1 << 253, (1 << 253) - 1, 32767, 32768, 32769, 65535, 65536]
    for batch in (1, 63, 64, 65):
        rows = []
        for p in range(batch):
            row = [r.randrange(H.R) for _ in range(n)]
            for k in range(n):
                if (p + k) % 3 == 0:
                    row[k] = extreme[(p * 7 + k) % len(extreme)]
            rows.append(H.to_mont_array(row))
        sc = np.stack(rows)
        want = np.stack([cref.msm(1, bases, sc[i], naive=True) for i in range(batch)])
        out = np.zeros((batch, 8), dtype=np.uint64)
        zk_ctx.msm_batch(h, sc, batch, out)
        assert np.array_equal(out, want)
    zk_ctx.msm_batch(h, np.zeros((0, n, 4), dtype=np.uint64), 0, np.zeros((0, 8), dtype=np.uint64))
    zk_ctx.msm_bases_free(h)
    # no bases at all: every sum is the point at infinity
    h0 = zk_ctx.msm_bases_load(1, np.zeros((0, 8), dtype=np.uint64), 0, c)
    out = np.ones((3, 8), dtype=np.uint64)
    zk_ctx.msm_batch(h0, np.zeros((3, 0, 4), dtype=np.uint64), 3, out)
    assert not out.any()
    zk_ctx.msm_bases_free(h0)


def test_empty_batches(zk_ctx):
    e = np.zeros((0, 16, 4), dtype=np.uint64)
    zk_ctx.ntt_batch(e, 4, 0)
    zk_ctx.h_batch(e, e, e, e.copy(), 4, 0)
    zk_ctx.fixed_base_mul(1, H.g1_gen_mont(), np.zeros((0, 4), dtype=np.uint64), 0,
                          np.zeros((0, 8), dtype=np.uint64))

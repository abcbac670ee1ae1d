"""GPU: float-precision dosages on the int8 matrix cores (rvtests_amd/csrc/suffstat_fdx.hip.h — K = g 2^37 in five balanced
base-256 digits; quantitative trait, M <= 64).  The kernel's own arithmetic is checked exactly by tools/fdx_bench check; here
the engine path: which genes take it, that the records equal the fp64 path's and the oracle's, that off-grid values and wide
genes fall back, and that BGEN genes take it without being told."""
import numpy as np
import pytest

import orc
import rvtests_amd

pytestmark = pytest.mark.gpu


def _bgen_dosages(rng, N, M, rate=0.04):
    """Dosages as an 8-bit BGEN block gives them: p = float32(v) * float32(1/255), dosage = p1 + 2 p2 in double."""
    s = np.float32(1.0 / 255.0)
    carrier = rng.random((N, M)) < rate
    doubt = rng.random((N, M)) < 2 * rate
    v1 = np.where(carrier, 255 - rng.integers(0, 32, (N, M)), np.where(doubt, rng.integers(0, 8, (N, M)), 0))
    v2 = np.where(carrier, rng.integers(0, 16, (N, M)), np.where(doubt, rng.integers(0, 2, (N, M)), 0))
    v1 = np.minimum(v1, 255 - v2)
    p1 = (v1.astype(np.float32) * s).astype(np.float64)
    p2 = (v2.astype(np.float32) * s).astype(np.float64)
    return np.asfortranarray(p1 + 2.0 * p2)


def _setup(N, seed):
    rng = np.random.default_rng(seed)
    X = np.asfortranarray(np.column_stack([np.ones(N), rng.normal(size=(N, 2))]))
    return rng, X


def _records(eng, genes, afs, tests):
    ptrs = [eng.upload_block(G) for G in genes]
    eng.set_profiling(True)
    eng.timing(reset=True)
    try:
        out = eng.run_blocks(ptrs, [G.shape[1] for G in genes], afs, tests=tests)
        tm = eng.timing(reset=True)
    finally:
        eng.set_profiling(False)
    for p in ptrs:
        eng.free_block(p)
    return out, tm


FIELDS = ("skat_Q", "skat_p", "skato_Q", "skato_p", "skato_rho", "cmc_stat", "cmc_p", "zeg_stat", "zeg_p", "cmc_nonref")


def _same(a, b, rel=1e-9, floor=1e-11):
    for f in FIELDS:
        x, y = float(getattr(a, f)), float(getattr(b, f))
        assert abs(x - y) <= rel * max(abs(x), abs(y)) + floor, (f, x, y)


@pytest.mark.parametrize("N", [5000, 8192 + 37])
def test_float_dosages_take_the_int8_kernel_and_match_the_fp64_path(N):
    rng, X = _setup(N, 7)
    genes = [_bgen_dosages(rng, N, M) for M in (5, 20, 33, 50, 64, 70)]
    y = X @ np.array([0.3, 0.5, -0.2]) + rng.normal(size=N) + 0.4 * genes[3][:, :4].sum(1)
    afs = [np.clip(G.mean(0) / 2, 1e-6, 1.0) for G in genes]
    tests = rvtests_amd.TEST_ALL
    out = {}
    for mode in ("fdx", "fp64"):
        eng = rvtests_amd.Engine(0)
        eng.fit_null(0, X, y.copy())
        eng.set_content_hint(0)
        if mode == "fdx":
            eng.set_dosage_float(True)
        recs, tm = _records(eng, genes, afs, tests)
        out[mode] = recs
        if mode == "fdx":
            assert tm.genes_hard_call == 5          # M = 70 is wider than the kernel's classes: fp64 kernel from the start
            assert tm.genes_handed_back == 0
        else:
            assert tm.genes_hard_call == 0
        eng.close()
    for a, b in zip(out["fdx"], out["fp64"]):
        _same(a, b)


def test_off_grid_values_are_handed_back():
    N = 6000
    rng, X = _setup(N, 11)
    G = _bgen_dosages(rng, N, 40)
    G[1234, 7] = 0.998                      # a decimal dosage: not a multiple of 2^-37
    H = _bgen_dosages(rng, N, 30)
    y = X @ np.array([0.1, 0.2, 0.3]) + rng.normal(size=N)
    afs = [np.clip(g.mean(0) / 2, 1e-6, 1.0) for g in (G, H)]
    eng = rvtests_amd.Engine(0)
    eng.fit_null(0, X, y.copy())
    eng.set_content_hint(0)
    eng.set_dosage_float(True)
    recs, tm = _records(eng, [G, H], afs, rvtests_amd.TEST_ALL)
    assert tm.genes_hard_call == 2 and tm.genes_handed_back == 1
    eng2 = rvtests_amd.Engine(0)
    eng2.fit_null(0, X, y.copy())
    eng2.set_content_hint(0)
    ref, _ = _records(eng2, [G, H], afs, rvtests_amd.TEST_ALL)
    for a, b in zip(recs, ref):
        _same(a, b)


def test_against_the_oracle():
    N = 4000
    rng, X = _setup(N, 13)
    G = _bgen_dosages(rng, N, 24, rate=0.06)
    y = X @ np.array([0.2, -0.4, 0.1]) + rng.normal(size=N) + 0.5 * G[:, :3].sum(1)
    af = np.clip(G.mean(0) / 2, 1e-6, 1.0)
    eng = rvtests_amd.Engine(0)
    nm = eng.fit_null(0, X, y.copy())
    eng.set_content_hint(0)
    eng.set_dosage_float(True)
    recs, tm = _records(eng, [G], [af], rvtests_amd.TEST_ALL)
    assert tm.genes_hard_call == 1 and tm.genes_handed_back == 0
    r = recs[0]
    beta = np.linalg.lstsq(X, y, rcond=None)[0]
    res = y - X @ beta
    v = np.ones(N)
    rc, o = orc.skat(G, af, X, res, v, 0)
    assert rc == 0 and r.n_poly == o.n_poly
    assert abs(r.skat_Q - o.Q) <= 1e-9 * o.Q and abs(r.skat_p - o.pvalue) <= 1e-6 * o.pvalue + 1e-14
    rc2, so = orc.skato(G, af, X, res, v, 0)
    assert rc2 == 0 and abs(r.skato_p - so.pvalue) <= 1e-6 * so.pvalue + 5e-13
    for which, stat_p in ((0, r.cmc_p), (1, r.zeg_p)):
        rcb, b = orc.burden(G, X, y, 0, which)
        if rcb == 0:
            assert abs(stat_p - b.pvalue) <= 1e-6 * b.pvalue + 1e-14
    if which == 0:
        pass


def test_bgen_genes_take_it_when_asked():
    """rvt_submit_gene_bgen marks its blocks as dosages (kind 0).  Round 5: the float-digit kernel is OPT-IN
    (rvt_set_dosage_float(1)) — it gives G'G as an exact integer but does not pay in speed, so BGEN genes start on the fp64
    kernel unless the caller asks.  Asked: complete 8-bit blocks (every value a multiple of 2^-31) go to gene_suffstat_fdx and give
    the records of the fp64 path; not asked: none of them does."""
    import bgengen
    import synth
    rng = np.random.default_rng(5)
    N, d = 6000, 3
    X, y, res, v, s2 = synth.make_null(N, d, 0, seed=9)
    genes = [[bgengen.layout2_block_fast(rng, N, bits=8, missing=0.0) for _ in range(M)] for M in (7, 30, 48)]
    out = {}
    for mode in ("default", "fp64", "not_asked"):
        eng = rvtests_amd.Engine(0)
        eng.set_null(0, X, res, v, s2)
        if mode == "fp64":
            eng.set_hardcall(False)
        if mode == "default":
            eng.set_dosage_float(True)
        eng.set_profiling(True)
        eng.timing(reset=True)
        for g, blocks in enumerate(genes):
            eng.submit_gene_bgen(g, blocks, 2)
        out[mode] = eng.collect()
        tm = eng.timing(reset=True)
        eng.set_profiling(False)
        if mode == "default":
            assert tm.genes_hard_call == len(genes) and tm.genes_handed_back == 0
        else:
            assert tm.genes_hard_call == 0
        eng.close()
    for a, b in zip(out["default"], out["fp64"]):
        assert a.n_poly == b.n_poly and a.status == b.status
        _same(a, b)


@pytest.mark.parametrize("N,d,M", [(777, 1, 9), (1000, 6, 33), (64, 2, 3), (33, 1, 1), (4097, 4, 64)])
def test_shapes_covariate_counts_flipped_and_constant_columns(N, d, M):
    """Ragged ends (N not a multiple of 16 / 32 / 128), one to six covariates (3 to 15 null columns), a column whose allele
    frequency is above 1/2 (flipped), a constant column, a column of zeros."""
    rng = np.random.default_rng(100 + N + d)
    X = np.asfortranarray(np.column_stack([np.ones(N)] + [rng.normal(size=N) for _ in range(d - 1)]))
    G = _bgen_dosages(rng, N, M, rate=0.08)
    if M > 2:
        G[:, 1] = 2.0 - G[:, 1]                      # major-allele dosages: the column is flipped
        G[:, 2] = np.float32(11.0 / 255.0)           # monomorphic (a constant on the grid)
    if M > 4:
        G[:, 4] = 0.0
    y = X @ rng.normal(size=d) + rng.normal(size=N) + 0.3 * G[:, 0]
    af = np.clip(G.mean(0) / 2, 0.0, 1.0)
    out = {}
    for mode in ("fdx", "fp64"):
        eng = rvtests_amd.Engine(0)
        eng.fit_null(0, X, y.copy())
        eng.set_content_hint(0)
        if mode == "fdx":
            eng.set_dosage_float(True)
        recs, tm = _records(eng, [G], [af], rvtests_amd.TEST_ALL)
        if mode == "fdx":
            assert tm.genes_hard_call == 1 and tm.genes_handed_back == 0
        out[mode] = recs[0]
        eng.close()
    a, b = out["fdx"], out["fp64"]
    assert a.n_poly == b.n_poly and a.status == b.status and a.cmc_nonref == b.cmc_nonref
    _same(a, b)

"""GPU: KinshipHolder::decompose on the device (rvt_kinship_decompose, one-sided block Jacobi).  The reference calls
Eigen's SelfAdjointEigenSolver<MatrixXf> (base/KinshipHolder.cpp:270-290) and holds no test or fixture for it; the
checks here are the defining properties (K U = U diag(S), U'U = I, ascending S equal to LAPACK's eigenvalues of the same
float matrix) and the invariance the family tests rely on: statistics computed through the device's U, S equal those
computed through LAPACK's U, S although the bases of repeated eigenvalues differ."""
import numpy as np
import pytest

import orc
import synth
from test_fam_cpu import make_family_case

pytestmark = pytest.mark.gpu


@pytest.fixture
def eng():
    import rvtests_amd
    e = rvtests_amd.Engine(0)
    yield e
    e.close()


def _check(K32, U, S, info):
    K = K32.astype(np.float64)
    N = K.shape[0]
    ref = np.linalg.eigvalsh(K)
    scale = np.abs(ref).max()
    assert (np.diff(S) >= 0).all()
    assert np.abs(S - ref).max() <= 3e-7 * scale                 # float storage of S
    Ud = U.astype(np.float64)
    assert np.abs(Ud.T @ Ud - np.eye(N)).max() <= 5e-6           # float storage of U
    assert np.abs(K @ Ud - Ud * S.astype(np.float64)).max() <= 5e-6 * scale
    assert info.max_cosine < 1e-10 and info.padded_order % 64 == 0 and info.padded_order >= N


@pytest.mark.parametrize("n_fam", [16, 75, 333])
def test_family_kinship(eng, n_fam):
    """Nuclear-family kinship: eigenvalues 0.5 / 1 / 2 each repeated n_fam times or more."""
    N, K, U0, S0, X, y = make_family_case(n_fam, 2, 7)
    K32 = K.astype(np.float32)
    U, S, info = eng.kinship_decompose(K32)
    _check(K32, U, S, info)
    assert info.shift == 0.0
    if n_fam == 333:
        # the device's eigenvectors of a block-diagonal kinship are block-sparse themselves (columns of different
        # families never mix in the Jacobi rotations): installing them finds the structure
        assert (np.count_nonzero(U, axis=0) <= 4).all()
        eng.kinship_decompose(K32, install=True, want_vectors=False)
        assert eng.kinship_structure() < 0.35


@pytest.mark.parametrize("solver", ["default", "jacobi"])
@pytest.mark.parametrize("N,kind", [(300, "psd"), (1000, "grm"), (257, "indefinite"), (130, "opposite")])
def test_dense_matrices(eng, N, kind, solver, monkeypatch):
    """Dense matrices take the tridiagonalisation + bisection + inverse iteration (sweeps = 0) unless eigenvalues repeat
    (the rank-deficient matrix: its null space goes to the Jacobi iteration); RVT_KINSHIP_JACOBI=1 keeps them all on the
    Jacobi iteration."""
    if solver == "jacobi":
        monkeypatch.setenv("RVT_KINSHIP_JACOBI", "1")
    rng = np.random.default_rng(N)
    if kind == "grm":                                            # genetic relationship matrix Z Z' / m: PSD, dense, full rank
        Z = rng.standard_normal((N, 3 * N))
        K = Z @ Z.T / (3 * N)
    elif kind == "psd":                                          # rank-deficient
        Z = rng.standard_normal((N, N // 2))
        K = Z @ Z.T / N
    elif kind == "indefinite":
        A = rng.standard_normal((N, N))
        K = (A + A.T) / 2
    else:                                                        # exactly opposite eigenvalue pairs: needs the shift
        Q, _ = np.linalg.qr(rng.standard_normal((N, N)))
        lam = np.concatenate([np.linspace(0.5, 3.0, N // 2), -np.linspace(0.5, 3.0, N // 2)])
        K = (Q * lam) @ Q.T
    K32 = ((K + K.T) / 2).astype(np.float32)
    U, S, info = eng.kinship_decompose(K32)
    _check(K32.astype(np.float32), U, S, info)
    if kind in ("psd", "grm"):
        assert info.shift == 0.0
    if solver == "jacobi" or kind == "psd":
        assert info.sweeps > 0
        if kind == "opposite":
            assert info.shift > 0.0
    else:
        assert info.sweeps == 0 and info.shift == 0.0 and info.max_residual <= 1e-12 * np.abs(S).max() * N
    if kind == "grm":                                            # simple spectrum: eigenvectors themselves, up to sign
        w, V = np.linalg.eigh(K32.astype(np.float64))
        gaps = np.minimum(np.diff(w, prepend=-np.inf), np.diff(w, append=np.inf))
        ok = gaps > 1e-3 * np.abs(w).max()
        dots = np.abs(np.sum(V * U.astype(np.float64), axis=0))
        assert (dots[ok] > 1 - 1e-4).all() and ok.sum() > 50


def test_inverse_iteration_in_batches_gives_the_same_vectors(eng, monkeypatch):
    """The eigenvectors of the tridiagonal form are computed a batch of columns at a time (as many as the free device memory
    holds factors for: several batches only at N ~ 100 000).  A start vector depends on the eigenvector's number in the spectrum,
    not on its place in a batch, so U must not depend on the batch size: bit-identical with batches of 192 and of 64 columns."""
    rng = np.random.default_rng(5)
    N = 700
    Z = rng.standard_normal((N, 2 * N))
    K32 = ((Z @ Z.T) / (2 * N)).astype(np.float32)
    K32 = (K32 + K32.T) / 2
    U0, S0, info0 = eng.kinship_decompose(K32)
    assert info0.sweeps == 0
    _check(K32, U0, S0, info0)
    for batch in (192, 64):
        monkeypatch.setenv("RVT_TRIDIAG_BATCH", str(batch))
        U1, S1, info1 = eng.kinship_decompose(K32)
        assert info1.sweeps == 0 and np.array_equal(S1, S0) and np.array_equal(U1, U0)


def test_famskat_through_the_device_decomposition(eng):
    """install = 1: FamSKAT through the device's own U, S equals FamSKAT through LAPACK's (different bases of the repeated
    eigenvalues, same U f(S) U')."""
    N, K, U0, S0, X, y = make_family_case(60, 2, 21)
    genes = [synth.make_gene(N, M, seed=500 + M, missing=0.02, common=True)[1] for M in (8, 25)]

    def run(install_from_device):
        if install_from_device:
            eng.kinship_decompose(K.astype(np.float32), install=True, want_vectors=False)
        else:
            eng.set_kinship(U0, S0)
        nul = eng.fit_fam_null(X, y)
        ptrs = [eng.upload_block(G) for G in genes]
        out = eng.run_fam_blocks(ptrs, [G.shape[1] for G in genes])
        return nul, [(r.famskat_Q, r.famskat_p) for r in out]

    nul_a, a = run(False)
    nul_b, b = run(True)
    assert abs(nul_a.delta - nul_b.delta) <= 2e-3 + 1e-3 * nul_a.delta      # Brent's own stopping accuracy
    for (qa, pa), (qb, pb) in zip(a, b):
        assert abs(qa - qb) <= 2e-2 * qa and abs(pa - pb) <= 5e-2 * pa + 1e-6
    # with the SAME variance components the two bases must agree to float accuracy: pin delta through the oracle path
    onul = orc.FamNull()
    onul.ok = 1
    onul.delta, onul.sigma2 = nul_a.delta, nul_a.sigma2_g
    for k in range(2):
        onul.beta[k] = nul_a.beta[k]
    U1, S1, info = eng.kinship_decompose(K.astype(np.float32))
    for G in genes:
        rc0, o0 = orc.famskat(G, X, y, U0, S0, onul)
        rc1, o1 = orc.famskat(G, X, y, U1.astype(np.float64), S1.astype(np.float64), onul)
        assert rc0 == rc1 == 0
        assert abs(o0.Q - o1.Q) <= 2e-5 * o0.Q and abs(o0.pvalue - o1.pvalue) <= 1e-4 * o0.pvalue


def _block_kinship(rng, sizes):
    N = int(np.sum(sizes))
    K = np.zeros((N, N))
    o = 0
    for sz in sizes:
        A = rng.uniform(0.05, 0.5, (sz, sz))
        K[o:o + sz, o:o + sz] = (A + A.T) / 2 + np.eye(sz)
        o += sz
    return K


def test_block_diagonal_kinship_is_decomposed_family_by_family(eng):
    """Separate families (sizes 1..9, an all-zero row / column among them): no block sweeps, the eigenpairs of the
    blocks merged in ascending order — with the families listed one after the other and with the samples shuffled
    (connected components of the sparsity pattern).  One family larger than a tile sends the matrix to the dense
    iteration; the spectrum is the same every time."""
    rng = np.random.default_rng(12)
    sizes = rng.choice([1, 2, 3, 4, 6, 9], size=260)
    K = _block_kinship(rng, sizes)
    K[7, :] = 0.0
    K[:, 7] = 0.0                                        # an isolated sample without even a diagonal entry
    K32 = K.astype(np.float32)
    U, S, info = eng.kinship_decompose(K32)
    _check(K32, U, S, info)
    assert info.sweeps == 0 and info.shift == 0.0
    assert (np.count_nonzero(U, axis=0) <= 9).all()
    perm = rng.permutation(K.shape[0])
    Kp = np.ascontiguousarray(K32[np.ix_(perm, perm)])
    U2, S2, info2 = eng.kinship_decompose(Kp)
    _check(Kp, U2, S2, info2)
    assert info2.sweeps == 0 and (np.count_nonzero(U2, axis=0) <= 9).all()
    assert np.abs(S2 - S).max() <= 2e-6 * np.abs(S).max()
    big = _block_kinship(rng, [70, 3, 4, 2, 5] * 20).astype(np.float32)
    U3, S3, info3 = eng.kinship_decompose(big)
    _check(big, U3, S3, info3)
    # (a dense route took it — the Jacobi iteration, or the tridiagonal form when no two eigenvalues are too close — : the
    #  family-by-family route would have refused the family of 70)
    assert info3.padded_order >= big.shape[0]


def test_shuffled_families_through_famskat(eng):
    """Families interleaved in the sample order: decomposition by connected components, installation, FamSKAT against the
    oracle on the same (shuffled) data."""
    N, K, U0, S0, X, y = make_family_case(90, 2, 33)
    rng = np.random.default_rng(4)
    perm = rng.permutation(N)
    Kp = np.ascontiguousarray(K[np.ix_(perm, perm)]).astype(np.float32)
    Xp, yp = X[perm], y[perm]
    genes = [synth.make_gene(N, M, seed=300 + M, missing=0.01, common=True)[1] for M in (6, 21)]
    U1, S1, info = eng.kinship_decompose(Kp)
    assert info.sweeps == 0
    eng.kinship_decompose(Kp, install=True, want_vectors=False)
    assert eng.kinship_structure() < 0.02                      # 4 non-zeros per eigenvector of 360: the gather rotation
    nul = eng.fit_fam_null(Xp, yp)
    ptrs = [eng.upload_block(G) for G in genes]
    out = eng.run_fam_blocks(ptrs, [G.shape[1] for G in genes])
    onul = orc.FamNull()
    onul.ok = 1
    onul.delta, onul.sigma2 = nul.delta, nul.sigma2_g
    for k in range(2):
        onul.beta[k] = nul.beta[k]
    for r, G in zip(out, genes):
        rc, o = orc.famskat(G, Xp, yp, U1.astype(np.float64), S1.astype(np.float64), onul)
        assert rc == 0 and r.famskat_ok == 1
        assert abs(r.famskat_Q - o.Q) <= 1e-6 * o.Q and abs(r.famskat_p - o.pvalue) <= 1e-5 * o.pvalue + 1e-12

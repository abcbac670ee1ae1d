"""The C++ host side (rvtests_amd/csrc/host): ModelParser / ModelManager::create / the GPU ModelFitter adapters,
driven like the reference's gene loop (src/Main.cpp:1207-1256) by host_driver.
CPU part: registry, parser and output formats (without a GPU every fit fails loudly -> NA rows, no CPU fallback).
GPU part: printed numbers equal the oracle's, formatted as the reference prints them (%g / 6 significant digits)."""
import os
import struct
import subprocess

import numpy as np
import pytest

import orc
import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "rvtests_amd", "csrc", "host", "host_driver")


def _ensure_driver():
    if not os.path.exists(DRIVER):
        import __graft_entry__ as g
        g.build()


def write_input(path, y, cov, binary, genes):
    N = len(y)
    with open(path, "wb") as f:
        f.write(struct.pack("<qiii", N, cov.shape[1] if cov.size else 0, int(binary), len(genes)))
        f.write(np.ascontiguousarray(y, dtype="<f8").tobytes())
        if cov.size:
            f.write(np.asfortranarray(cov, dtype="<f8").tobytes(order="F"))
        for G, af in genes:
            f.write(struct.pack("<i", G.shape[1]))
            f.write(np.ascontiguousarray(af, dtype="<f8").tobytes())
            f.write(np.asfortranarray(G, dtype="<f8").tobytes(order="F"))


def run_driver(path, kernel, burden, batch=None, perm_exact=None):
    env = dict(os.environ)
    env.pop("RVT_PERM_EXACT", None)                # default of the drop-in: replay the reference's rand() stream
    if perm_exact is not None:
        env["RVT_PERM_EXACT"] = "1" if perm_exact else "0"
    if batch:
        env["RVT_ADAPTER_BATCH"] = str(batch)      # genes the adapters keep in flight before collecting
    p = subprocess.run([DRIVER, path, kernel, burden], capture_output=True, text=True, timeout=300, env=env)
    sections = {}
    cur = None
    for line in p.stdout.splitlines():
        if line.startswith("== "):
            cur = line[3:]
            sections[cur] = []
        elif cur:
            sections[cur].append(line.split("\t"))
    return p.returncode, sections, p.stderr


def _case(tmp_path, binary=0, d=3, N=1500):
    genes = [synth.make_gene(N, M, seed=40 + M, missing=0.01, common=True, mono=(M > 8))[1:] for M in (6, 21, 1, 40)]
    X, y, res, v, s2 = synth.make_null(N, d, binary, seed=4, G_effect=0.5 * genes[0][0][:, :2].sum(1))
    path = str(tmp_path / "in.bin")
    write_input(path, y, X[:, 1:], binary, genes)
    return path, genes, X, y, res, v


def test_registry_parser_and_na_rows_without_gpu(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    _ensure_driver()
    path, genes, X, y, res, v = _case(tmp_path)
    rc, sec, err = run_driver(path, "skat[nPerm=0:beta1=1,beta2=25],SkatO", "cmc,zeggini")
    assert rc == 0
    assert list(sec) == ["out.Skat.assoc", "out.SkatO.assoc", "out.CMC.assoc", "out.Zeggini.assoc"]
    assert sec["out.Skat.assoc"][0][-2:] == ["Q", "Pvalue"]
    assert sec["out.SkatO.assoc"][0][-3:] == ["Q", "rho", "Pvalue"]
    assert sec["out.CMC.assoc"][0][-2:] == ["NonRefSite", "Pvalue"]
    assert sec["out.Zeggini.assoc"][0][-1:] == ["Pvalue"]
    for name, ncol in (("out.Skat.assoc", 2), ("out.SkatO.assoc", 3), ("out.CMC.assoc", 2), ("out.Zeggini.assoc", 1)):
        assert len(sec[name]) == 1 + len(genes)
        for row in sec[name][1:]:
            assert row[-ncol:] == ["NA"] * ncol      # no device => fit() fails => NA, never a CPU result
    rc, sec, err = run_driver(path, "nosuchmodel", "-")
    assert rc == 1 and "Unknown model name: nosuchmodel" in err
    rc, sec, err = run_driver(path, "skat[nperm=5", "-")
    assert rc == 1 and "format" in err


@pytest.mark.gpu
@pytest.mark.parametrize("binary,d,batch", [(0, 3, None), (1, 1, None), (0, 3, 3), (1, 2, 64)])
def test_driver_output_matches_oracle(tmp_path, binary, d, batch):
    _ensure_driver()
    path, genes, X, y, res, v = _case(tmp_path, binary=binary, d=d)
    rc, sec, err = run_driver(path, "skat[nPerm=0],skato", "cmc,zeggini", batch)
    assert rc == 0, err
    if batch:       # deferred, batched collection must not change a single character of the output
        rc1, sec1, err1 = run_driver(path, "skat[nPerm=0],skato", "cmc,zeggini")
        assert sec1 == sec

    def g(x):
        return "%g" % x

    def f6(x):
        return np.format_float_positional(float("%.6g" % x), trim="-") if False else ("%.6g" % x)

    for i, (G, af) in enumerate(genes):
        rc1, a = orc.skat(G, af, X, res, v, binary)
        rc2, o = orc.skato(G, af, X, res, v, binary)
        row = sec["out.Skat.assoc"][1 + i]
        assert row[0] == "gene%d" % i and row[2] == str(G.shape[1])
        if a.n_poly == 0:
            assert row[-2:] == ["NA", "NA"]
            continue
        assert abs(float(row[-2]) - a.Q) <= 6e-6 * a.Q and abs(float(row[-1]) - a.pvalue) <= 6e-6 * a.pvalue + 1e-14
        row = sec["out.SkatO.assoc"][1 + i]
        if rc2 == 0:
            assert abs(float(row[-3]) - o.Q) <= 6e-6 * o.Q and float(row[-2]) == o.rho      # %g: 6 digits
            assert abs(float(row[-1]) - o.pvalue) <= 6e-6 * o.pvalue + 1e-12
        else:
            assert row[-3:] == ["NA"] * 3
        rc3, c = orc.burden(G, X, y, binary, 0)
        row = sec["out.CMC.assoc"][1 + i]
        if rc3 == 0:
            assert int(row[-2]) == c.nonref_site and abs(float(row[-1]) - c.pvalue) <= 6e-6 * c.pvalue
        rc4, z = orc.burden(G, X, y, binary, 1)
        row = sec["out.Zeggini.assoc"][1 + i]
        if rc4 == 0:
            assert abs(float(row[-1]) - z.pvalue) <= 6e-6 * z.pvalue


def run_driver_meta(path, meta, sites_path, block=None, rect_above=None):
    env = dict(os.environ)
    if block:
        env["RVT_METACOV_BLOCK"] = str(block)
    if rect_above:
        env["RVT_METACOV_RECT_ABOVE"] = str(rect_above)
    p = subprocess.run([DRIVER, path, "-", "-", meta, sites_path], capture_output=True, text=True, timeout=300,
                       env=env)
    lines = [ln for ln in p.stdout.splitlines()]
    return p.returncode, lines, p.stderr


def test_meta_registry_without_gpu(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    _ensure_driver()
    path, genes, X, y, res, v = _case(tmp_path)
    sites = str(tmp_path / "sites.txt")
    with open(sites, "w") as f:
        k = 0
        for G, af in genes:
            for j in range(G.shape[1]):
                f.write("1 %d\n" % (100 + 10 * k))
                k += 1
    rc, lines, err = run_driver_meta(path, "cov[windowSize=200]", sites)
    assert rc == 0, err
    assert lines[0] == "== out.MetaCov.assoc"
    assert lines[1].split("\t") == ["CHROM", "START_POS", "END_POS", "NUM_MARKER", "MARKER_POS", "COV"]
    assert len(lines) == 2                      # no device => no rows, never a CPU result
    rc, lines, err = run_driver_meta(path, "nosuch", sites)
    assert rc == 1 and "Unknown model name: nosuch" in err


@pytest.mark.gpu
@pytest.mark.parametrize("binary,gwama,block,rect", [(0, False, None, None), (0, True, None, None),
                                                     (1, False, None, None), (0, False, 16, None),
                                                     (1, True, 19, None), (0, False, 8, 8), (1, True, 8, 12)])
def test_driver_metacov_rows_match_oracle(tmp_path, binary, gwama, block, rect):
    """--meta cov through the C++ adapter: row structure (window rule, monomorphic sites skipped, chromosome change)
    identical to the oracle's, numbers equal to its fp64 values after the reference's float / 1/N / %g formatting."""
    _ensure_driver()
    N, d = 1500, 3
    genes = [synth.make_gene(N, M, seed=70 + M, missing=0.01, common=True, mono=True)[1:] for M in (30, 25)]
    X, y, res, v, s2 = synth.make_null(N, d, binary, seed=9)
    path = str(tmp_path / "in.bin")
    write_input(path, y, X[:, 1:], binary, genes)
    G = np.concatenate([g for g, af in genes], axis=1)
    V = G.shape[1]
    rng = np.random.default_rng(3)
    pos = np.cumsum(rng.integers(1, 300, V)).astype(np.int32)
    chrom = np.where(np.arange(V) < 40, 1, 2).astype(np.int32)
    sites = str(tmp_path / "sites.txt")
    with open(sites, "w") as f:
        for c, p_ in zip(chrom, pos):
            f.write("%d %d\n" % (c, p_))
    window = 1200
    # block: a device ring far smaller than the stream, so rows are emitted by several mid-stream flushes with
    # compaction in between (the window then has to fit the ring: at most ~10 sites per window here)
    if block and not rect:
        window = 450
    # rect: the ring starts smaller than one window (it has to grow) and everything wider than `rect` columns goes
    # through the heads-by-window rectangle instead of the symmetric block kernel
    rc, lines, err = run_driver_meta(path, "cov[windowSize=%d%s]" % (window, ":gwama" if gwama else ""), sites, block,
                                     rect)
    assert rc == 0, err
    assert lines[0] == "== out.MetaCov.assoc"
    rows = [ln.split("\t") for ln in lines[2:]]
    rc, kept, cov, row_end, xz, zz = orc.metacov(G, chrom, pos, X, y, binary, window)
    assert rc == 0
    heads = [h for h in range(V) if kept[h]]
    assert len(rows) == len(heads)
    scale = np.float32(1.0 / N)
    for row, h in zip(rows, heads):
        js = [j for j in range(h, row_end[h] + 1) if kept[j] and not np.isnan(cov[h, j])]
        assert row[0] == str(chrom[h]) and row[1] == str(pos[h]) and row[2] == str(pos[row_end[h]])
        assert int(row[3]) == len(js)
        assert row[4] == ",".join(str(pos[j]) for j in js)
        parts = row[5].split(":")
        assert len(parts) == (3 if (binary or gwama) else 1)
        got = np.array([float(t) for t in parts[0].split(",")])
        want = np.array([float(np.float32(cov[h, j]) * scale) for j in js])
        assert (np.abs(got - want) <= 6e-6 * np.abs(want) + 1e-30).all()      # %g prints 6 significant digits
        if len(parts) == 3:
            gx = np.array([float(t) for t in parts[1].split(",")])
            wx = np.array([float(np.float32(x) * scale) for x in xz[h]])
            assert np.allclose(gx, wx, rtol=2e-5, atol=1e-5 * max(np.abs(wx).max(), 1e-30))
            gz = np.array([float(t) for t in parts[2].split(",")])
            wz = np.array([zz[a, b] * float(scale) for a in range(d) for b in range(a + 1)])
            assert np.allclose(gz, wz, rtol=2e-5, atol=1e-12)


@pytest.mark.gpu
def test_driver_famskat_matches_oracle(tmp_path):
    """--kernel famSkat through the C++ adapter: kinship + FastLMM null installed once, one row per gene."""
    _ensure_driver()
    from test_fam_cpu import make_family_case
    N, K, U, S, X, y = make_family_case(50, 2, 31)
    genes = [synth.make_gene(N, M, seed=90 + M, missing=0.01, common=True, mono=(M > 5))[1:] for M in (9, 24, 3)]
    genes.append((np.ones((N, 2)), np.zeros(2)))          # monomorphic only -> NA row
    path = str(tmp_path / "in.bin")
    write_input(path, y, X[:, 1:], 0, genes)
    kin = str(tmp_path / "kin.bin")
    with open(kin, "wb") as f:
        f.write(struct.pack("<q", N))
        f.write(np.asfortranarray(U, dtype="<f4").tobytes(order="F"))
        f.write(np.ascontiguousarray(S, dtype="<f4").tobytes())
    p = subprocess.run([DRIVER, path, "famSkat[beta1=1:beta2=25]", "famcmc,famzeggini", "-", "-", kin],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    sections, cur = {}, None
    for ln in p.stdout.splitlines():
        if ln.startswith("== "):
            cur = ln[3:]
            sections[cur] = []
        else:
            sections[cur].append(ln.split("\t"))
    assert list(sections) == ["out.FamSkat.assoc", "out.FamCMC.assoc", "out.FamZeggini.assoc"]
    assert sections["out.FamCMC.assoc"][0][-6:] == ["NumSite", "AF", "U", "V", "Effect", "Pvalue"]
    assert sections["out.FamZeggini.assoc"][0][-6:] == ["NumSite", "MeanBurden", "U", "V", "Effect", "Pvalue"]
    lines = ["== out.FamSkat.assoc"] + ["\t".join(r) for r in sections["out.FamSkat.assoc"]]
    assert lines[1].split("\t")[-2:] == ["Q", "Pvalue"]
    rows = [ln.split("\t") for ln in lines[2:]]
    assert len(rows) == len(genes)
    rc0, onul0 = orc.fastlmm_null(X, y, U, S)
    for name, which in (("out.FamCMC.assoc", 0), ("out.FamZeggini.assoc", 1)):
        for row, (G, af) in zip(sections[name][1:], genes):
            rcb, ob = orc.fam_burden(G, X, y, U, S, onul0, which)
            if rcb != 0:
                assert row[-6:] == ["NA"] * 6
                continue
            assert int(row[-6]) == ob.num_site
            assert abs(float(row[-5]) - ob.af) <= 1e-5 * abs(ob.af) + 1e-12
            # (the null fit is pinned to the reference's Brent stopping accuracy only)
            assert abs(float(row[-4]) - ob.U) <= 2e-2 * abs(ob.U) + 1e-6
            assert abs(float(row[-3]) - ob.V) <= 2e-2 * ob.V
            assert abs(float(row[-2]) - ob.U / ob.V) <= 3e-2 * abs(ob.U / ob.V) + 1e-6
            assert abs(np.log(float(row[-1])) - np.log(ob.pvalue)) <= 5e-2 * max(1.0, abs(np.log(ob.pvalue)))
    rc, onul = orc.fastlmm_null(X, y, U, S)
    assert rc == 0
    for row, (G, af) in zip(rows, genes):
        rc, o = orc.famskat(G, X, y, U, S, onul)
        if rc != 0:
            assert row[-2:] == ["NA", "NA"]
            continue
        # the null fit is pinned only to the accuracy of the reference's Brent stopping rule (test_gpu_fam.py)
        assert abs(float(row[-2]) - o.Q) <= 2e-2 * o.Q
        assert abs(np.log(float(row[-1])) - np.log(o.pvalue)) <= 5e-2 * max(1.0, abs(np.log(o.pvalue)))


@pytest.mark.gpu
def test_driver_skat_with_permutations(tmp_path):
    """default-style `skat` (nPerm > 0): the eight columns of SkatTest::writeOutput with Permutation's fields, counts
    equal to the oracle's replay of the same rand() stream across consecutive genes."""
    _ensure_driver()
    path, genes, X, y, res, v = _case(tmp_path, binary=0, d=2, N=700)
    rc, sec, err = run_driver(path, "skat[nPerm=150:alpha=0.1]", "-")      # no mode selected: the DEFAULT is exact
    assert rc == 0, err
    rows = sec["out.Skat.assoc"]
    assert rows[0][-8:] == ["Q", "Pvalue", "NumPerm", "ActualPerm", "Stat", "NumGreater", "NumEqual", "PermPvalue"]
    orc.rand_seed(1)
    for row, (G, af) in zip(rows[1:], genes):
        rc1, a = orc.skat(G, af, X, res, v, 0)
        if a.n_poly == 0:
            assert row[-8:] == ["NA"] * 8
            continue
        rc2, p = orc.skat_permute(G, af, res, a.Q, 150, 0.1)
        assert row[-6] == "150" and int(row[-5]) == p.actual_perm
        assert int(row[-3]) == p.num_x and int(row[-2]) == p.num_equal
        assert abs(float(row[-1]) - p.pvalue) <= 1e-5 * max(p.pvalue, 1e-30)
        assert row[-4] == "%g" % a.Q or abs(float(row[-4]) - a.Q) <= 2e-6 * a.Q


@pytest.mark.gpu
@pytest.mark.parametrize("rect", [None, 8])
def test_driver_metacov_with_kinship(tmp_path, rect):
    """--meta cov with a kinship decomposition: MetaCovFamQtl through the adapter (rows and numbers vs the oracle);
    rect = 8 forces the heads x window rectangles (rvt_cov_rect_fam) that windows wider than one block take."""
    _ensure_driver()
    from test_fam_cpu import make_family_case
    N, K, U, S, X, y = make_family_case(45, 2, 61)
    _, G, af = synth.make_gene(N, 28, seed=12, missing=0.01, common=True, mono=True)
    path = str(tmp_path / "in.bin")
    write_input(path, y, X[:, 1:], 0, [(G, af)])
    kin = str(tmp_path / "kin.bin")
    with open(kin, "wb") as f:
        f.write(struct.pack("<q", N))
        f.write(np.asfortranarray(U, dtype="<f4").tobytes(order="F"))
        f.write(np.ascontiguousarray(S, dtype="<f4").tobytes())
    V = G.shape[1]
    pos = np.cumsum(np.random.default_rng(4).integers(1, 300, V)).astype(np.int32)
    chrom = np.ones(V, dtype=np.int32)
    sites = str(tmp_path / "sites.txt")
    with open(sites, "w") as f:
        for p_ in pos:
            f.write("1 %d\n" % p_)
    env = dict(os.environ)
    if rect:
        env["RVT_METACOV_RECT_ABOVE"] = str(rect)
        env["RVT_METACOV_BLOCK"] = str(rect)
    p = subprocess.run([DRIVER, path, "-", "-", "cov[windowSize=1000]", sites, kin], capture_output=True, text=True,
                       timeout=300, env=env)
    assert p.returncode == 0, p.stderr
    lines = p.stdout.splitlines()
    assert lines[0] == "== out.MetaCov.assoc"
    rows = [ln.split("\t") for ln in lines[2:]]
    rc, onul = orc.fastlmm_null(X, y, U, S)
    rc, kept, cov, row_end, xz, zz = orc.metacov_fam(G, chrom, pos, X, U, S, onul, 1000)
    heads = [h for h in range(V) if kept[h]]
    assert len(rows) == len(heads)
    scale = np.float32(1.0 / N)
    for row, h in zip(rows, heads):
        js = [j for j in range(h, row_end[h] + 1) if kept[j] and not np.isnan(cov[h, j])]
        assert row[1] == str(pos[h]) and int(row[3]) == len(js)
        got = np.array([float(t) for t in row[5].split(",")])
        want = np.array([float(np.float32(cov[h, j]) * scale) for j in js])
        # the null fit (delta) is pinned to the reference's Brent stopping accuracy only, see test_gpu_fam.py
        assert np.allclose(got, want, rtol=2e-2, atol=2e-3 * np.abs(want).max())


def _score_case(tmp_path, binary, N=1500, d=3):
    genes = [synth.make_gene(N, M, seed=170 + M, missing=0.01, common=True, mono=True)[1:] for M in (30, 25)]
    X, y, res, v, s2 = synth.make_null(N, d, binary, seed=19)
    path = str(tmp_path / "in.bin")
    write_input(path, y, X[:, 1:], binary, genes)
    G = np.concatenate([g for g, af in genes], axis=1)
    sites = str(tmp_path / "sites.txt")
    with open(sites, "w") as f:
        for k in range(G.shape[1]):
            f.write("1 %d\n" % (100 + 10 * k))
    return path, sites, G, X, y


def test_metascore_registry_without_gpu(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    _ensure_driver()
    path, sites, G, X, y = _score_case(tmp_path, 0)
    rc, lines, err = run_driver_meta(path, "score[se]", sites)
    assert rc == 0, err
    assert lines[0] == "== out.MetaScore.assoc"
    # no device => no null model: as when the reference's null fit fails, every row carries the caller's site counters
    # and NA statistics — never a CPU result
    assert lines[1].split("\t")[-5:] == ["U_STAT", "SQRT_V_STAT", "ALT_EFFSIZE", "ALT_EFFSIZE_SE", "PVALUE"]
    assert len(lines) == 2 + G.shape[1]
    assert all(ln.split("\t")[-5:] == ["NA"] * 5 for ln in lines[2:])


@pytest.mark.gpu
@pytest.mark.parametrize("binary,se,block", [(0, False, None), (0, True, 7), (1, False, None), (1, True, 16)])
def test_driver_metascore_rows_match_oracle(tmp_path, binary, se, block):
    """--meta score through the C++ adapter: the deferred summary header with the null-model estimates, one row per
    site in file order (NA statistics for monomorphic sites), numbers equal to the oracle's after %g formatting;
    `block` forces several mid-stream flushes."""
    _ensure_driver()
    path, sites, G, X, y = _score_case(tmp_path, binary)
    N, V = G.shape
    d = X.shape[1]
    env = dict(os.environ)
    if block:
        env["RVT_METASCORE_BLOCK"] = str(block)
    p = subprocess.run([DRIVER, path, "-", "-", "score[se]" if se else "score", sites], capture_output=True,
                       text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stderr
    lines = p.stdout.splitlines()
    assert lines[0] == "== out.MetaScore.assoc"
    rc, o = orc.metascore(G, X, y, binary)
    assert rc == 0
    # summary header (MetaScoreTest::PrintNullModel)
    assert lines[1] == "##NullModelEstimates" and lines[2] == "## - Name\tBeta\tSD"
    est = [ln.split("\t") for ln in lines[3:3 + d + 1]]
    assert [e[0] for e in est] == ["## - Intercept"] + ["## - cov%d" % k for k in range(1, d)] + ["## - Sigma2"]
    for k in range(d):
        assert float(est[k][1]) == pytest.approx(o["beta"][k], rel=6e-6, abs=1e-9)
        assert float(est[k][2]) == pytest.approx(o["covb"][k], rel=6e-6)
    if binary:
        assert est[d][1:] == ["NA", "NA"]
    else:
        assert float(est[d][1]) == pytest.approx(o["sigma2"], rel=6e-6) and est[d][2] == "NA"
    hdr = lines[4 + d].split("\t")
    cols = ["CHROM", "POS", "AF", "INFORMATIVE_ALT_AC", "CALL_RATE", "HWE_PVALUE", "N_REF", "N_HET", "N_ALT", "U_STAT",
            "SQRT_V_STAT", "ALT_EFFSIZE"] + (["ALT_EFFSIZE_SE"] if se else []) + ["PVALUE"]
    assert hdr == cols
    rows = [ln.split("\t") for ln in lines[5 + d:]]
    assert len(rows) == V
    n_na = 0
    for h, row in enumerate(rows):
        assert len(row) == len(cols)
        assert row[0] == "1" and row[1] == str(100 + 10 * h)
        g = G[:, h]
        if binary:
            af = [float(t) for t in row[2].split(":")]
            assert len(af) == 3 and af[0] == pytest.approx(g.sum() / (2 * N), rel=6e-6, abs=1e-12)
            assert af[1] == pytest.approx(g[y == 1].sum() / (2 * (y == 1).sum()), rel=6e-6, abs=1e-12)
            nref = [int(t) for t in row[6].split(":")]
            assert nref[0] == nref[1] + nref[2] == int((np.rint(g) <= 0).sum())
        else:
            assert float(row[2]) == pytest.approx(g.sum() / (2 * N), rel=6e-6, abs=1e-12)
            assert [int(row[6]), int(row[7]), int(row[8])] == [int((np.rint(g) <= 0).sum()), int((np.rint(g) == 1).sum()),
                                                              int((np.rint(g) >= 2).sum())]
        stats = row[9:]
        if not o["ok"][h]:
            assert stats == ["NA"] * len(stats)
            n_na += 1
            continue
        want = [o["U"][h], np.sqrt(o["V"][h]), o["effect"][h]] + ([o["se"][h]] if se else []) + [o["p"][h]]
        for got, w in zip(stats, want):
            assert float(got) == pytest.approx(w, rel=6e-6, abs=1e-12)
    assert 0 < n_na < V // 2


@pytest.mark.gpu
@pytest.mark.parametrize("binary", [0, 1])
def test_driver_metascore_with_kinship(tmp_path, binary):
    """--meta score with a kinship decomposition: MetaFamQtl / MetaFamBinary through the adapter — SigmaG2 / SigmaE2
    header, the GLS allele frequency in the AF column of tested sites, U / sqrt(V) / effect / SE / p of the FastLMM score
    test (binary: uncentred genotypes and the b scaling)."""
    _ensure_driver()
    from test_fam_cpu import make_family_case
    N, K, U, S, X, y = make_family_case(45, 2, 61)
    if binary:
        y = (y > np.median(y)).astype(float)
    _, G, af = synth.make_gene(N, 28, seed=12, missing=0.01, common=True, mono=True)
    path = str(tmp_path / "in.bin")
    write_input(path, y, X[:, 1:], binary, [(G, af)])
    kin = str(tmp_path / "kin.bin")
    with open(kin, "wb") as f:
        f.write(struct.pack("<q", N))
        f.write(np.asfortranarray(U, dtype="<f4").tobytes(order="F"))
        f.write(np.ascontiguousarray(S, dtype="<f4").tobytes())
    V = G.shape[1]
    sites = str(tmp_path / "sites.txt")
    with open(sites, "w") as f:
        for k in range(V):
            f.write("1 %d\n" % (100 + k))
    p = subprocess.run([DRIVER, path, "-", "-", "score[se]", sites, kin], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    lines = p.stdout.splitlines()
    assert lines[0] == "== out.MetaScore.assoc" and lines[1] == "##NullModelEstimates"
    Uf, Sf = U.astype(np.float32).astype(np.float64), S.astype(np.float32).astype(np.float64)
    rc, onul = orc.fastlmm_null(X, y, Uf, Sf)
    assert rc == 0
    b = 1.0
    if binary:
        b = orc.obtain_b(float(np.float32(np.log((y == 1).sum() / (y == 0).sum()))))
    est = [ln.split("\t") for ln in lines[3:7]]
    assert [e[0] for e in est] == ["## - Intercept", "## - cov1", "## - SigmaG2", "## - SigmaE2"]
    # the null fit (delta) is pinned to the reference's Brent stopping accuracy only, see test_gpu_fam.py
    assert float(est[2][1]) == pytest.approx(onul.sigma2, rel=2e-2)
    assert float(est[3][1]) == pytest.approx(onul.sigma2 * onul.delta, rel=5e-2)
    assert lines[7].split("\t")[-5:] == ["U_STAT", "SQRT_V_STAT", "ALT_EFFSIZE", "ALT_EFFSIZE_SE", "PVALUE"]
    rows = [ln.split("\t") for ln in lines[8:]]
    assert len(rows) == V
    tested = 0
    for h, row in enumerate(rows):
        rc, o = orc.fam_burden(G[:, [h]], X, y, Uf, Sf, onul, 3 if binary else 2)
        if rc:
            assert row[-5:] == ["NA"] * 5
            continue
        tested += 1
        want = [o.af, o.U * b, np.sqrt(o.V) * b, o.U / o.V / b, 1 / np.sqrt(o.V * b * b) / b, o.pvalue]
        got = [float(row[2].split(":")[0])] + [float(t) for t in row[-5:]]
        assert np.allclose(got, want, rtol=2e-2, atol=2e-3 * np.abs(want).max())
    assert tested > 10


@pytest.mark.gpu
def test_driver_analytic_vt_rows(tmp_path):
    """--vt analytic through the C++ adapter: header, %g-style columns, NA rule, numbers against the oracle."""
    _ensure_driver()
    N, d = 800, 2
    genes = [synth.make_gene(N, M, seed=310 + M, missing=0.01, common=(M > 10), mono=True)[1:] for M in (6, 21, 35)]
    X, y, res, v, s2 = synth.make_null(N, d, 0, seed=29)
    path = str(tmp_path / "in.bin")
    write_input(path, y, X[:, 1:], 0, genes)
    env = dict(os.environ)
    env["RVT_DRIVER_VT"] = "analytic"
    p = subprocess.run([DRIVER, path, "-", "-"], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stderr
    lines = p.stdout.splitlines()
    assert lines[0] == "== out.AnalyticVT.assoc"
    assert lines[1].split("\t")[-8:] == ["MinMAF", "MaxMAF", "OptimMAF", "OptimNumVar", "U", "V", "Stat", "Pvalue"]
    rows = [ln.split("\t") for ln in lines[2:]]
    assert len(rows) == len(genes)
    for row, (G, af) in zip(rows, genes):
        rc, o, cor = orc.analytic_vt(G, af, X, y, mvn_points=1024)
        if rc != 0:
            assert row[-8:] == ["NA"] * 8
            continue
        got = row[-8:]
        assert int(got[3]) == o.opt_num
        for k, want in zip((0, 1, 2, 4, 5, 6), (o.min_maf, o.max_maf, o.opt_maf, o.U, o.V, o.stat)):
            assert abs(float(got[k]) - want) <= 6e-6 * abs(want) + 1e-12          # floatToString: 6 digits
        assert abs(float(got[7]) - o.pvalue) <= 2e-3


@pytest.mark.gpu
def test_driver_kbac_rows(tmp_path):
    """--kernel kbac[nPerm:alpha] through the C++ adapter: binary trait, no covariates; %f p-values equal to the oracle's
    (the oracle is pinned on the reference's own kbac.cpp), genes consuming one random stream in order."""
    _ensure_driver()
    N = 900
    rng = np.random.default_rng(8)
    genes = [synth.make_gene(N, M, seed=410 + M, missing=0.01, common=False, mono=True)[1:] for M in (7, 16, 3)]
    y = (rng.random(N) < 0.45).astype(np.float64)
    path = str(tmp_path / "in.bin")
    write_input(path, y, np.zeros((N, 0)), 1, genes)
    p = subprocess.run([DRIVER, path, "kbac[nPerm=600:alpha=0.05]", "-"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    lines = p.stdout.splitlines()
    assert lines[0] == "== out.Kbac.assoc" and lines[1].split("\t")[-1] == "Pvalue"
    rows = [ln.split("\t") for ln in lines[2:]]
    assert len(rows) == len(genes)
    orc.rand_seed(1)                                   # the stream of a fresh process
    for row, (G, af) in zip(rows, genes):
        Gf, fl, kp = orc.flip_poly(G)
        if Gf.shape[1] == 0:
            assert row[-1] == "NA"
            continue
        pv = orc.kbac(Gf, y, af[:Gf.shape[1]], 600, 0.05)[0]
        assert row[-1] == "%f" % pv
    # a quantitative trait is refused with NA rows, as the reference does
    write_input(path, rng.standard_normal(N), np.zeros((N, 0)), 0, genes)
    p = subprocess.run([DRIVER, path, "kbac", "-"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and all(ln.split("\t")[-1] == "NA" for ln in p.stdout.splitlines()[2:])


@pytest.mark.gpu
def test_driver_fam_analytic_vt_rows(tmp_path):
    """--vt famanalytic through the C++ adapter (kinship file as for famSkat): rows against the oracle's literal
    restatement fed with the oracle's own FastLMM null (delta is pinned to Brent's stopping accuracy only)."""
    _ensure_driver()
    from test_fam_cpu import make_family_case
    N, K, U, S, X, y = make_family_case(45, 2, 71)
    genes = [synth.make_gene(N, M, seed=510 + M, missing=0.01, common=True, mono=(M > 5))[1:] for M in (8, 19)]
    path = str(tmp_path / "in.bin")
    write_input(path, y, X[:, 1:], 0, genes)
    kin = str(tmp_path / "kin.bin")
    with open(kin, "wb") as f:
        f.write(struct.pack("<q", N))
        f.write(np.asfortranarray(U, dtype="<f4").tobytes(order="F"))
        f.write(np.ascontiguousarray(S, dtype="<f4").tobytes())
    env = dict(os.environ)
    env["RVT_DRIVER_VT"] = "famanalytic"
    p = subprocess.run([DRIVER, path, "-", "-", "-", "-", kin], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stderr
    lines = p.stdout.splitlines()
    assert lines[0] == "== out.FamAnalyticVT.assoc"
    rows = [ln.split("\t") for ln in lines[2:]]
    assert len(rows) == len(genes)
    rc, onul = orc.fastlmm_null(X, y, U, S)
    for row, (G, af) in zip(rows, genes):
        rc, o, cor = orc.fam_analytic_vt(G, X, y, U, S, onul, mvn_points=1024)
        got = row[-8:]
        if rc != 0:
            assert got == ["NA"] * 8
            continue
        assert int(got[3]) == o.opt_num
        for k, want in zip((2, 4, 5, 6), (o.opt_maf, o.U, o.V, o.stat)):
            assert abs(float(got[k]) - want) <= 2e-2 * abs(want) + 1e-9
        assert abs(float(got[7]) - o.pvalue) <= 2e-2


@pytest.mark.gpu
def test_dosage_input_with_the_lattice_stated_prints_the_same_rows(tmp_path):
    """`--dosage DS` with three decimals: the adapter states the lattice (RVT_DOSAGE=d3 -> GpuBroker::setDosage(3) ->
    rvt_group_set_content) and the genes run on the int8 lattice kernel; the printed rows are those of the run without
    the statement (fp64 kernel after a hand-back) and carry the oracle's numbers."""
    _ensure_driver()
    N, d = 2000, 3
    rng = np.random.default_rng(6)
    genes = []
    for M in (6, 21, 1, 40):
        H = rng.binomial(2, 10 ** rng.uniform(-2.5, -1.0, M), size=(N, M))
        blur = np.rint(np.abs(rng.normal(0, 40, size=(N, M)))).astype(np.int64) * (rng.random((N, M)) < 0.3)
        K = np.clip(np.where(H == 0, blur, 1000 * H - blur), 0, 2000)
        G = np.asfortranarray(K / 1000.0)
        genes.append((G, G.sum(0) / (2.0 * N)))
    X, y, res, v, s2 = synth.make_null(N, d, 0, seed=4, G_effect=0.5 * genes[0][0][:, :2].sum(1))
    path = str(tmp_path / "in.bin")
    write_input(path, y, X[:, 1:], 0, genes)
    out = {}
    for tag in ("", "d3", "1"):
        os.environ.pop("RVT_DOSAGE", None)
        if tag:
            os.environ["RVT_DOSAGE"] = tag
        try:
            rc, sec, err = run_driver(path, "skat[nPerm=0],skato", "cmc,zeggini")
        finally:
            os.environ.pop("RVT_DOSAGE", None)
        assert rc == 0, err
        out[tag] = sec
    assert out["1"] == out[""]                              # the same kernel either way: hint only

    def same(a, b):                                         # the lattice kernel's G'G is exact, the fp64 one's rounded: a last
        try:                                                # printed digit may differ
            return abs(float(a) - float(b)) <= 2e-6 * abs(float(b))
        except ValueError:
            return a == b
    for name in out[""]:
        assert len(out["d3"][name]) == len(out[""][name])
        for ra, rb in zip(out["d3"][name], out[""][name]):
            assert len(ra) == len(rb) and all(same(x, y_) for x, y_ in zip(ra, rb)), (name, ra, rb)
    rows = out["d3"]["out.Skat.assoc"][1:]
    for (G, af), row in zip(genes, rows):
        rc, a = orc.skat(G, af, X, res, v, 0)
        assert abs(float(row[-2]) - a.Q) <= 6e-6 * a.Q and abs(float(row[-1]) - a.pvalue) <= 6e-6 * a.pvalue + 1e-14


# ---- the synthetic drop-in run (host_driver --synthetic: what bench.py times at configs[2] size as `drop_in`) ---------------------
_M64 = (1 << 64) - 1


def _mix64(x):
    """splitmix64's finaliser on uint64 arrays (wrap-around arithmetic), as host_driver.cpp's mix64."""
    x = (np.asarray(x, dtype=np.uint64) + np.uint64(0x9E3779B97F4A7C15))
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def synthetic_inputs(N, Mlo, Mhi, pool, seed=1):
    """The generator of host_driver --synthetic restated: covariates, phenotype and the pool's genotype blocks."""
    with np.errstate(over="ignore"):
        i = np.arange(N, dtype=np.uint64)
        cov = np.empty((N, 2))
        for k in range(2):
            cov[:, k] = (_mix64(np.uint64((seed * 1000003 + k + 1) & _M64) + i * np.uint64(7)) >> np.uint64(11)) / 9007199254740992.0 - 0.5
        y = 0.3 * cov[:, 0] + (_mix64(np.uint64((seed * 999983 + 5) & _M64) + i * np.uint64(13)) >> np.uint64(11)) / 9007199254740992.0
        blocks = []
        for k in range(pool):
            M = Mlo + int(_mix64(np.uint64((seed * 7919 + k) & _M64))) % (Mhi - Mlo + 1)
            G = np.empty((N, M), order="F")
            for j in range(M):
                thr = np.uint64(200 + int(_mix64(np.uint64((seed * 131 + k * 4099 + j) & _M64))) % 3000)
                base = np.uint64((((k * 1000003 + j) * 0x100000001B3) + seed) & _M64)
                h = _mix64(base + i)
                G[:, j] = ((h & np.uint64(0xffff)) < thr).astype(np.float64) + (((h >> np.uint64(16)) & np.uint64(0xffff)) < thr)
            blocks.append(G)
    return cov, y, blocks


def run_synthetic(mode, N, genes, Mlo, Mhi, pool, rows_path, kernel="skat[nPerm=0],skato", burden="cmc,zeggini"):
    import json
    p = subprocess.run([DRIVER, "--synthetic", str(N), str(genes), str(Mlo), str(Mhi), mode, kernel, burden, "--pool", str(pool),
                        "--rows", rows_path], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    sections, cur = {}, None
    for ln in open(rows_path).read().splitlines(keepends=True):
        if ln.startswith("== "):
            cur = ln[3:].strip()
            sections[cur] = ""
        else:
            sections[cur] += ln
    return line, sections


@pytest.mark.gpu
def test_synthetic_drop_in_rows_match_oracle(tmp_path):
    """host_driver --synthetic (the C++ adapters end to end: ModelManager, fit / writeOutput per gene, GpuBroker's deferred ordered
    rows) from the fp64 boundary and from PLINK 2-bit rows: the same .assoc text either way (same SHA-256, which is the SHA-256
    of the text the driver wrote), one row per gene and model in gene order, and the numbers of sampled genes equal to the
    oracle's on the regenerated inputs, formatted as the reference prints them."""
    import hashlib
    _ensure_driver()
    N, genes, Mlo, Mhi, pool = 3000, 40, 5, 30, 8
    line64, sec64 = run_synthetic("fp64", N, genes, Mlo, Mhi, pool, str(tmp_path / "rows64.txt"))
    lineb, secb = run_synthetic("bed", N, genes, Mlo, Mhi, pool, str(tmp_path / "rowsb.txt"))
    assert line64["rows_sha256"] == lineb["rows_sha256"] and sec64 == secb
    names = ["out.Skat.assoc", "out.SkatO.assoc", "out.CMC.assoc", "out.Zeggini.assoc"]
    assert list(sec64) == names
    sha = hashlib.sha256("".join(hashlib.sha256(sec64[n].encode()).hexdigest() for n in names).encode()).hexdigest()
    assert sha == line64["rows_sha256"]
    assert line64["assoc_lines"] == 4 * (genes + 1) and line64["gene_sets_per_s"] > 0 and line64["caller_us_per_gene"] > 0
    cov, y, blocks = synthetic_inputs(N, Mlo, Mhi, pool)
    X = np.column_stack([np.ones(N), cov])
    rc, beta, pred, res, s2 = orc.fit_linear(X, y)
    assert rc == 0
    v = np.full(N, s2)
    rows = {n: [r.split("\t") for r in sec64[n].splitlines()] for n in names}
    for g in (0, 3, 7, 8, 21, 39):
        G = blocks[g % pool]
        af = G.sum(0) / (2.0 * N)
        for n in names:
            assert rows[n][1 + g][0] == "gene%d" % g and rows[n][1 + g][2] == str(G.shape[1])
        rc1, a = orc.skat(G, af, X, res, v, 0)
        row = rows["out.Skat.assoc"][1 + g]
        if a.n_poly == 0:
            assert row[-2:] == ["NA", "NA"]
            continue
        assert abs(float(row[-2]) - a.Q) <= 6e-6 * a.Q and abs(float(row[-1]) - a.pvalue) <= 6e-6 * a.pvalue + 1e-14
        rc2, o = orc.skato(G, af, X, res, v, 0)
        row = rows["out.SkatO.assoc"][1 + g]
        if rc2 == 0:
            assert abs(float(row[-3]) - o.Q) <= 6e-6 * o.Q and float(row[-2]) == o.rho
            assert abs(float(row[-1]) - o.pvalue) <= 6e-6 * o.pvalue + 1e-12
        rc3, c = orc.burden(G, X, y, 0, 0)
        row = rows["out.CMC.assoc"][1 + g]
        if rc3 == 0:
            assert int(row[-2]) == c.nonref_site and abs(float(row[-1]) - c.pvalue) <= 6e-6 * c.pvalue
        rc4, z = orc.burden(G, X, y, 0, 1)
        row = rows["out.Zeggini.assoc"][1 + g]
        if rc4 == 0:
            assert abs(float(row[-1]) - z.pvalue) <= 6e-6 * z.pvalue
    # genes that share a pool block print the same numbers
    assert rows["out.Skat.assoc"][1 + 3][3:] == rows["out.Skat.assoc"][1 + 3 + pool][3:]


def test_synthetic_generator_restatement_cpu():
    """The numpy restatement of the driver's generator: allele frequencies in the stated range, hard calls only (CPU)."""
    cov, y, blocks = synthetic_inputs(2000, 5, 12, 3)
    assert cov.shape == (2000, 2) and np.abs(cov).max() <= 0.5 and len(blocks) == 3
    for G in blocks:
        assert 5 <= G.shape[1] <= 12 and set(np.unique(G)) <= {0.0, 1.0, 2.0}
        af = G.mean(0) / 2
        assert af.max() < 0.07

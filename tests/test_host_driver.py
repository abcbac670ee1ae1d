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


def run_driver(path, kernel, burden):
    p = subprocess.run([DRIVER, path, kernel, burden], capture_output=True, text=True, timeout=300)
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
@pytest.mark.parametrize("binary,d", [(0, 3), (1, 1)])
def test_driver_output_matches_oracle(tmp_path, binary, d):
    _ensure_driver()
    path, genes, X, y, res, v = _case(tmp_path, binary=binary, d=d)
    rc, sec, err = run_driver(path, "skat[nPerm=0],skato", "cmc,zeggini")
    assert rc == 0, err

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
        assert abs(float(row[-2]) - a.Q) <= 2e-6 * a.Q and abs(float(row[-1]) - a.pvalue) <= 2e-6 * a.pvalue + 1e-14
        row = sec["out.SkatO.assoc"][1 + i]
        if rc2 == 0:
            assert abs(float(row[-3]) - o.Q) <= 2e-6 * o.Q and float(row[-2]) == o.rho
            assert abs(float(row[-1]) - o.pvalue) <= 2e-6 * o.pvalue + 1e-12
        else:
            assert row[-3:] == ["NA"] * 3
        rc3, c = orc.burden(G, X, y, binary, 0)
        row = sec["out.CMC.assoc"][1 + i]
        if rc3 == 0:
            assert int(row[-2]) == c.nonref_site and abs(float(row[-1]) - c.pvalue) <= 2e-6 * c.pvalue
        rc4, z = orc.burden(G, X, y, binary, 1)
        row = sec["out.Zeggini.assoc"][1 + i]
        if rc4 == 0:
            assert abs(float(row[-1]) - z.pvalue) <= 2e-6 * z.pvalue

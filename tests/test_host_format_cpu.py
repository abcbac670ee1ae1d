"""CPU: the adapter's own "%g" (formatG, csrc/host/ModelFitterGpu.cpp) prints the characters printf prints — MetaCov's rows are a
thousand such conversions each, so the adapter does not go through glibc's.  The reference prints these numbers with
fprintf("%g") / floatToString (src/Model.cpp:975-984, base/TypeConversion.h:100-105)."""
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
DRIVER = os.path.join(HERE, "..", "rvtests_amd", "csrc", "host", "host_driver")


def _format(lines):
    if not os.path.exists(DRIVER):
        import __graft_entry__ as g
        g.build()
    pr = subprocess.run([DRIVER, "--format"], input="\n".join(lines) + "\n", capture_output=True, text=True, timeout=600)
    assert pr.returncode == 0, pr.stderr[-300:]
    return [ln.split("\t")[1] for ln in pr.stdout.splitlines()]


def test_formatg_prints_what_printf_prints():
    rng = np.random.default_rng(7)
    vals = []
    # floats as the band holds them (covariances times 1/N), every magnitude the fast path covers and the ones it hands to printf
    mant = rng.uniform(1.0, 10.0, 600_000)
    expo = rng.integers(-25, 26, 600_000)
    sign = rng.choice([-1.0, 1.0], 600_000)
    vals += list((sign * mant * 10.0 ** expo).astype(np.float32).astype(np.float64))
    vals += list(sign[:300_000] * mant[:300_000] * 10.0 ** expo[:300_000])                    # doubles (SKAT's Q, p-values)
    vals += list(rng.integers(0, 2_000_000, 200_000).astype(np.float64))                      # integers: no point, no exponent
    vals += list(rng.integers(0, 2_000_000, 200_000) / 1000.0)                                # short decimals: zeros stripped
    # six-digit ties and near-ties (x.5 in the seventh digit): exactly representable ones, and their neighbours
    ties = rng.integers(100000, 1000000, 100_000) * 10 + 5
    for sh in (-12, -9, -7, -6, -3, 0, 2, 5):
        vals += list(ties[:12_000] * 10.0 ** sh)
        vals += list(np.nextafter(ties[:6_000] * 10.0 ** sh, np.inf))
        vals += list(np.nextafter(ties[:6_000] * 10.0 ** sh, -np.inf))
    vals += [0.0, -0.0, 1.0, -1.0, 9.999995, 9.9999949, 999999.5, 999999.4, 0.0001, 0.00009999995, 1e5, 1e6, 123456.5, 1e-4, 1e-5,
             1e21, 1e22, 1e23, 1e-17, 1e-18, 5e-324, 1.7976931348623157e308, float("inf"), float("-inf"), float("nan"),
             0.1, 0.2, 0.3, 2.5e-5, 100000.0, 1000000.0, 0.5, 1.5, 2.5, 1234565.0, 12345650.0]
    lines = [repr(float(v)) for v in vals]
    got = _format(lines)
    assert len(got) == len(vals)
    bad = [(v, g, "%g" % v) for v, g in zip(vals, got) if g != "%g" % v and not (v != v and g.lstrip("-") == "nan")]
    assert not bad, bad[:10]


def test_formatg_through_a_float():
    """"f:" lines: the value goes through a float first (what the device's band holds)."""
    rng = np.random.default_rng(8)
    x = (rng.normal(size=200_000) * 10.0 ** rng.integers(-9, 3, 200_000)).astype(np.float32)
    got = _format(["f:" + repr(float(v)) for v in x])
    bad = [(float(v), g) for v, g in zip(x, got) if g != "%g" % float(v)]
    assert not bad, bad[:10]

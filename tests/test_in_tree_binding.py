"""CPU: the in-tree binding (rvtests_amd/csrc/host/in_tree/) compiles against the reference's OWN plugin headers —
src/ModelFitter.h, src/Result.h, base/IO.h, base/MathMatrix.h, src/ModelParser.h, src/DataConsolidator.h, src/GenotypeCounter.h
— i.e. fit(DataConsolidator*), writeHeader / writeOutput(FileWriter*, const Result&), setParameter(const ModelParser&)
and the constructor calls of src/ModelManager.cpp really bind.  Syntax-only (nothing of the reference is built or
copied); base/CommonFunction.h includes GSL headers by their path inside the reference's vendored tarball, which is
unpacked to a scratch directory for the include path.  Skipped where /root/reference does not exist (the GPU box)."""
import glob
import os
import shutil
import subprocess
import tarfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
IN_TREE = os.path.join(ROOT, "rvtests_amd", "csrc", "host", "in_tree")


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src")), reason="reference tree not present")
def test_binding_compiles_against_the_reference_headers(tmp_path):
    inc = tmp_path / "incroot" / "third" / "gsl" / "include" / "gsl"
    inc.mkdir(parents=True)
    with tarfile.open(os.path.join(REF, "third", "gsl-1.16.tar.gz")) as tf:
        names = [m for m in tf.getmembers() if os.path.basename(m.name).startswith("gsl_") and m.name.endswith(".h")]
        tf.extractall(tmp_path / "x", members=names)
    for h in glob.glob(str(tmp_path / "x" / "**" / "gsl_*.h"), recursive=True):
        shutil.copy(h, inc)
    flags = ["-std=c++11", "-fsyntax-only", "-Wall", "-I" + REF, "-I" + os.path.join(REF, "src"),
             "-I" + os.path.join(REF, "base"), "-I" + str(tmp_path / "incroot"), "-I" + IN_TREE]
    for src in ("binding_check.cpp", "GpuDcShim.cpp"):
        p = subprocess.run(["g++"] + flags + [os.path.join(IN_TREE, src)], capture_output=True, text=True)
        assert p.returncode == 0, p.stderr[-3000:]
        assert "error" not in p.stderr

"""CPU: the C-ABI library loads and exports every symbol include/rvtests_amd.h declares; struct layouts used by
the ctypes binding match the header; no compute calls (there is no GPU here)."""
import ctypes as C
import os
import re
import subprocess

import pytest

import rvtests_amd
from rvtests_amd import engine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "rvtests_amd.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rvt_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    rvtests_amd.build_library()
    lib = rvtests_amd.load_library()
    names = declared_functions()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), "missing export: " + n
    assert lib.rvt_version().decode().startswith("rvtests_amd")
    assert lib.rvt_padded_ld(500000) == 500000 and lib.rvt_padded_ld(500001) == 500016


def test_struct_layout_matches_header(tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "rvtests_amd.h"\nint main(){printf("%zu %zu %zu %zu %zu\\n",'
                   'sizeof(rvt_gene_result),sizeof(rvt_params),sizeof(rvt_timing),offsetof(rvt_gene_result,skato_p),'
                   'offsetof(rvt_gene_result,davies_terms));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    a, b, c, d, e = map(int, subprocess.check_output([str(exe)]).split())
    assert a == C.sizeof(engine.GeneResult) and b == C.sizeof(engine.Params) and c == C.sizeof(engine.Timing)
    assert d == engine.GeneResult.skato_p.offset and e == engine.GeneResult.davies_terms.offset


def test_no_cpu_fallback():
    """Without a HIP device the engine refuses to start instead of computing on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(rvtests_amd.RvtError):
        rvtests_amd.Engine(0)


def test_product_does_not_import_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "rvtests_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "liboracle" not in txt and "import orc" not in txt and "oracle/" not in txt.replace(
                    "tests/test_oracle_ref.py", ""), f


def test_result_struct_sizes_agree():
    """The ctypes mirrors of rvt_gene_result (engine binding and host-harness binding) must have the size the C
    header gives the struct — a stale mirror lets the library write past a Python-owned buffer."""
    import ctypes as C
    import subprocess
    import tempfile
    import hc
    import rvtests_amd
    src = '#include <stdio.h>\n#include "rvtests_amd.h"\nint main(){printf("%zu %zu\\n", sizeof(rvt_gene_result), ' \
          'sizeof(rvt_fam_null));return 0;}\n'
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as td:
        cpath = os.path.join(td, "sz.c")
        open(cpath, "w").write(src)
        exe = os.path.join(td, "sz")
        subprocess.check_call(["gcc", "-I" + os.path.join(root, "include"), "-o", exe, cpath])
        a, b = (int(t) for t in subprocess.check_output([exe]).split())
    assert C.sizeof(rvtests_amd.GeneResult) == a
    assert C.sizeof(hc.GeneResult) == a
    from rvtests_amd.engine import FamNull
    assert C.sizeof(FamNull) == b

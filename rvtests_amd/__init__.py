"""rvtests_amd — MI355X-native kernel/burden association engine (SKAT, SKAT-O, CMC, Zeggini).

Python is plumbing only: this module loads the C-ABI shared library built from ``csrc/`` (hand-written
HIP for gfx950) with ctypes and offers a thin ``Engine`` wrapper plus the host-side mirror of the
reference's model registry (``rvtests_amd.models``).  There is no CPU fallback: if the HIP library is
missing or there is no GPU, construction fails loudly.
"""
from .engine import (Engine, GeneResult, Params, Timing, RvtError, build_library, library_path, load_library,
                     TEST_SKAT, TEST_SKATO, TEST_CMC, TEST_ZEGGINI, TEST_ALL, TRAIT_QUANTITATIVE, TRAIT_BINARY,
                     MAX_INFLIGHT)

__all__ = ["Engine", "GeneResult", "Params", "Timing", "RvtError", "build_library", "library_path", "load_library",
           "TEST_SKAT", "TEST_SKATO", "TEST_CMC", "TEST_ZEGGINI", "TEST_ALL", "TRAIT_QUANTITATIVE", "TRAIT_BINARY",
           "MAX_INFLIGHT"]

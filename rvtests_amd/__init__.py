"""rvtests_amd — MI355X-native kernel/burden association engine (SKAT, SKAT-O, CMC, Zeggini, FamSKAT, MetaCov,
MetaScore, AnalyticVT, KBAC, the VCF text front end and the kinship decomposition).

Python is plumbing only: this module loads the C-ABI shared library built from ``csrc/`` (hand-written
HIP for gfx950) with ctypes and offers thin wrappers of it: ``Engine`` (one context = one GPU) and ``Group``
(several GPUs behind one caller, rvt_group_*).  The host-side mirror of the reference's ModelFitter /
ModelManager plugin surface is C++ (``csrc/host/ModelFitterGpu.{h,cpp}``), as the reference's is.  There is no
CPU fallback: if the HIP library is missing or there is no GPU, construction fails loudly.
"""
from .engine import (Engine, Group, GeneResult, Params, Timing, RvtError, build_library, library_path, load_library,
                     TEST_SKAT, TEST_SKATO, TEST_CMC, TEST_ZEGGINI, TEST_ALL, TEST_ANALYTICVT, TRAIT_QUANTITATIVE,
                     TRAIT_BINARY, MAX_INFLIGHT, KbacResult, DecomposeInfo)

__all__ = ["Engine", "Group", "GeneResult", "Params", "Timing", "RvtError", "build_library", "library_path", "load_library",
           "TEST_SKAT", "TEST_SKATO", "TEST_CMC", "TEST_ZEGGINI", "TEST_ALL", "TEST_ANALYTICVT", "TRAIT_QUANTITATIVE",
           "TRAIT_BINARY", "MAX_INFLIGHT", "KbacResult", "DecomposeInfo"]

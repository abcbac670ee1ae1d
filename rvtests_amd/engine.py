"""ctypes binding of include/rvtests_amd.h (the C ABI of librvtests_amd.so).

Every call goes through the C ABI; numpy/torch are used only to hold host arrays and device pointers.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBNAME = "librvtests_amd.so"

TEST_SKAT, TEST_SKATO, TEST_CMC, TEST_ZEGGINI, TEST_ALL = 1, 2, 4, 8, 15
TEST_ANALYTICVT = 128
MAX_INFLIGHT = int(os.environ.get("RVT_PY_MAX_INFLIGHT", 8))  # RVT_MAX_INFLIGHT (include/rvtests_amd.h; the env only for tools/build_variant.sh builds)
TRAIT_QUANTITATIVE, TRAIT_BINARY = 0, 1

c_double_p = C.POINTER(C.c_double)
c_int_p = C.POINTER(C.c_int)


class RvtError(RuntimeError):
    pass


class Params(C.Structure):
    _fields_ = [("skat_beta1", C.c_double), ("skat_beta2", C.c_double), ("skato_beta1", C.c_double),
                ("skato_beta2", C.c_double), ("skat_nperm", C.c_int), ("skat_alpha", C.c_double)]

    @staticmethod
    def default():
        return Params(1.0, 25.0, 1.0, 25.0, 0, 0.05)


class GeneResult(C.Structure):
    _fields_ = [
        ("gene_id", C.c_int64), ("status", C.c_uint32), ("n_variants", C.c_int), ("n_poly", C.c_int),
        ("skat_ok", C.c_int), ("skat_Q", C.c_double), ("skat_p", C.c_double), ("skat_nlambda", C.c_int),
        ("skato_ok", C.c_int), ("skato_Q", C.c_double), ("skato_rho", C.c_double), ("skato_p", C.c_double),
        ("skato_qags_status", C.c_int), ("skato_qags_neval", C.c_int),
        ("cmc_ok", C.c_int), ("cmc_nonref", C.c_int), ("cmc_U", C.c_double), ("cmc_V", C.c_double),
        ("cmc_stat", C.c_double), ("cmc_p", C.c_double),
        ("zeg_ok", C.c_int), ("zeg_U", C.c_double), ("zeg_V", C.c_double), ("zeg_stat", C.c_double),
        ("zeg_p", C.c_double), ("davies_terms", C.c_double),
        ("perm_ok", C.c_int), ("perm_num_perm", C.c_int), ("perm_actual_perm", C.c_int),
        ("perm_num_greater", C.c_int), ("perm_num_equal", C.c_int), ("perm_pvalue", C.c_double),
        ("famskat_ok", C.c_int), ("famskat_Q", C.c_double), ("famskat_p", C.c_double),
        ("famcmc_ok", C.c_int), ("famcmc_af", C.c_double), ("famcmc_U", C.c_double), ("famcmc_V", C.c_double),
        ("famcmc_p", C.c_double),
        ("famzeg_ok", C.c_int), ("famzeg_af", C.c_double), ("famzeg_U", C.c_double), ("famzeg_V", C.c_double),
        ("famzeg_p", C.c_double),
        ("vt_ok", C.c_int), ("vt_optnum", C.c_int), ("vt_ncutoff", C.c_int),
        ("vt_minmaf", C.c_double), ("vt_maxmaf", C.c_double), ("vt_optmaf", C.c_double), ("vt_U", C.c_double),
        ("vt_V", C.c_double), ("vt_stat", C.c_double), ("vt_p", C.c_double), ("vt_p_error", C.c_double),
    ]


class FamNull(C.Structure):
    _fields_ = [("delta", C.c_double), ("sigma2_g", C.c_double), ("beta", C.c_double * 16), ("max_index", C.c_int),
                ("brent_evals", C.c_int)]


class Timing(C.Structure):
    _fields_ = [("ms_suffstat", C.c_double), ("ms_burden", C.c_double), ("ms_stats", C.c_double),
                ("ms_pvalue", C.c_double), ("n_suffstat_launches", C.c_int64), ("n_burden_launches", C.c_int64),
                ("n_stats_launches", C.c_int64), ("n_pvalue_launches", C.c_int64), ("genes", C.c_int64),
                ("alg_bytes", C.c_double), ("alg_flops", C.c_double), ("genes_hard_call", C.c_int64),
                ("ms_suffstat_hc", C.c_double), ("n_suffstat_hc_launches", C.c_int64), ("alg_bytes_hc", C.c_double),
                ("genes_handed_back", C.c_int64)]


def library_path():
    return os.environ.get("RVT_LIBRARY", os.path.join(CSRC, LIBNAME))


def build_library(force=False, verbose=False):
    """Compile the engine's translation units (csrc/*.hip) for gfx950 and link librvtests_amd.so (hipcc cross-compiles without a GPU)."""
    out = library_path()
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h", "rvt_group.cpp"))]
    srcs.append(os.path.join(os.path.dirname(HERE), "include", "rvtests_amd.h"))
    if not force and os.path.exists(out) and all(os.path.getmtime(s) <= os.path.getmtime(out) for s in srcs):
        return out
    # one object per translation unit, compiled in parallel (the fully unrolled K2 bodies dominate the compile time), then one
    # link.  Every unit leaves a dependency file (-MMD): a unit whose sources are older than its object is not compiled again.
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-Wno-unused-function"]
    units = ["rvt_engine.hip", "rvt_stream.hip", "rvt_fam.hip", "rvt_perm.hip", "rvt_meta.hip", "k2_unweighted.hip", "k2_weighted.hip", "k2_hardcall.hip", "k2_hardcall_w.hip",
             "k2_hardcall_x.hip", "k2_lattice.hip", "k2_packed.hip", "k2_floatdigit.hip"]

    def fresh(obj, dep):
        if force or not os.path.exists(obj) or not os.path.exists(dep):
            return False
        t = os.path.getmtime(obj)
        try:
            words = open(dep).read().replace("\\\n", " ").split()
        except OSError:
            return False
        deps = [w for w in words if not w.endswith(":")]
        return bool(deps) and all(os.path.exists(d) and os.path.getmtime(d) <= t for d in deps)

    objs, procs = [], []
    for u in units:
        obj = os.path.join(CSRC, u.replace(".hip", ".o"))
        dep = obj[:-2] + ".d"
        objs.append(obj)
        if fresh(obj, dep):
            continue
        cmd = ["hipcc"] + flags + ["-MMD", "-MF", dep, "-c", os.path.join(CSRC, u), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append(subprocess.Popen(cmd))
    gobj = os.path.join(CSRC, "rvt_group.o")     # device groups: plain host C++ over the C ABI
    procs.append(subprocess.Popen(["g++", "-std=c++17", "-O2", "-fPIC", "-c", os.path.join(CSRC, "rvt_group.cpp"), "-o", gobj]))
    objs.append(gobj)
    rcs = [p.wait() for p in procs]
    if any(rcs):
        raise subprocess.CalledProcessError(max(rcs), "hipcc -c")
    link = ["hipcc", "--offload-arch=gfx950", "-shared", "-o", out] + objs + ["-lpthread"]
    if verbose:
        print(" ".join(link))
    subprocess.check_call(link)
    return out


_lib = None


def pin_to_device_node(device=0):
    """Bind this thread (and the threads created after it) to the CPUs of the NUMA node the device hangs on
    (rvt_pin_to_device_node); returns the node or -1."""
    L = load_library()
    L.rvt_pin_to_device_node.restype = C.c_int
    L.rvt_pin_to_device_node.argtypes = [C.c_int]
    return int(L.rvt_pin_to_device_node(int(device)))


def load_library():
    """Load librvtests_amd.so; raises RvtError when it has not been built (no silent fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise RvtError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(the engine has no CPU fallback)" % path)
    L = C.CDLL(path)
    vp = C.c_void_p
    L.rvt_version.restype = C.c_char_p
    L.rvt_last_error.restype = C.c_char_p
    L.rvt_last_error.argtypes = [vp]
    L.rvt_padded_ld.restype = C.c_int64
    L.rvt_padded_ld.argtypes = [C.c_int64]
    L.rvt_init.restype = C.c_int
    L.rvt_init.argtypes = [C.POINTER(vp), C.c_int]
    L.rvt_destroy.restype = None
    L.rvt_destroy.argtypes = [vp]
    L.rvt_set_null.restype = C.c_int
    L.rvt_set_null.argtypes = [vp, C.c_int, C.c_int64, C.c_int, c_double_p, c_double_p, c_double_p, C.c_double]
    L.rvt_block_alloc.restype = C.c_int
    L.rvt_block_alloc.argtypes = [vp, C.c_int, C.POINTER(vp)]
    L.rvt_block_free.restype = C.c_int
    L.rvt_block_free.argtypes = [vp, vp]
    L.rvt_block_upload.restype = C.c_int
    L.rvt_block_upload.argtypes = [vp, vp, C.c_int, c_double_p]
    L.rvt_block_classify.restype = C.c_int
    L.rvt_block_classify.argtypes = [vp, vp, C.c_int, c_int_p]
    L.rvt_set_hardcall.restype = C.c_int
    L.rvt_set_hardcall.argtypes = [vp, C.c_int]
    L.rvt_set_content_hint.restype = C.c_int
    L.rvt_set_content_hint.argtypes = [vp, C.c_int]
    L.rvt_host_register.restype = C.c_int
    L.rvt_host_register.argtypes = [vp, C.c_void_p, C.c_size_t]
    L.rvt_host_unregister.restype = C.c_int
    L.rvt_host_unregister.argtypes = [vp, C.c_void_p]
    L.rvt_set_dosage_lattice.restype = C.c_int
    L.rvt_set_dosage_lattice.argtypes = [vp, C.c_int]
    L.rvt_set_dosage_float.restype = C.c_int
    L.rvt_set_dosage_float.argtypes = [vp, C.c_int]
    run_args = [vp, C.c_int, C.POINTER(vp), c_int_p, c_double_p, C.POINTER(C.c_int64), C.c_uint32,
                C.POINTER(Params), C.POINTER(GeneResult)]
    L.rvt_run_blocks.restype = C.c_int
    L.rvt_run_blocks.argtypes = run_args
    L.rvt_run_blocks_async.restype = C.c_int
    L.rvt_run_blocks_async.argtypes = run_args
    L.rvt_sync.restype = C.c_int
    L.rvt_sync.argtypes = [vp]
    L.rvt_wait_oldest.restype = C.c_int
    L.rvt_wait_oldest.argtypes = [vp]
    L.rvt_reserve.restype = C.c_int
    L.rvt_reserve.argtypes = [vp, C.c_int, c_int_p]
    L.rvt_submit_gene.restype = C.c_int
    L.rvt_submit_gene.argtypes = [vp, C.c_int64, C.c_int, c_double_p, c_double_p, C.c_uint32, C.POINTER(Params)]
    L.rvt_submit_gene_raw.restype = C.c_int
    L.rvt_submit_gene_raw.argtypes = [vp, C.c_int64, C.c_int, c_double_p, C.c_uint32, C.POINTER(Params), c_double_p]
    L.rvt_submit_gene_i8.restype = C.c_int
    L.rvt_submit_gene_i8.argtypes = [vp, C.c_int64, C.c_int, C.POINTER(C.c_int8), C.c_uint32, C.POINTER(Params),
                                     c_double_p]
    L.rvt_submit_gene_bed.restype = C.c_int
    L.rvt_submit_gene_bed.argtypes = [vp, C.c_int64, C.c_int, C.POINTER(C.c_uint8), C.c_uint32, C.POINTER(Params),
                                      c_double_p]
    L.rvt_collect.restype = C.c_int
    L.rvt_collect.argtypes = [vp, C.POINTER(GeneResult), C.c_int, c_int_p]
    L.rvt_collect_ready.restype = C.c_int
    L.rvt_collect_ready.argtypes = [vp, C.POINTER(GeneResult), C.c_int, c_int_p]
    L.rvt_debug_collapse.restype = C.c_int
    L.rvt_debug_collapse.argtypes = [vp, vp, C.c_int, c_double_p, c_double_p, c_int_p, c_int_p]
    L.rvt_debug_suffstat.restype = C.c_int
    L.rvt_debug_suffstat.argtypes = [vp, vp, C.c_int, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p,
                                     c_double_p]
    L.rvt_fit_null.restype = C.c_int
    L.rvt_fit_null.argtypes = [vp, C.c_int, C.c_int64, C.c_int, c_double_p, c_double_p, c_double_p, c_double_p]
    L.rvt_rand_seed.restype = C.c_int
    L.rvt_rand_seed.argtypes = [vp, C.c_uint]
    L.rvt_set_kinship.restype = C.c_int
    L.rvt_set_kinship.argtypes = [vp, C.c_int64, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.rvt_fit_fam_null.restype = C.c_int
    L.rvt_fit_fam_null.argtypes = [vp, C.c_int64, C.c_int, c_double_p, c_double_p, C.POINTER(FamNull)]
    L.rvt_run_fam_blocks.restype = C.c_int
    L.rvt_run_fam_blocks.argtypes = [vp, C.c_int, C.POINTER(vp), c_int_p, C.POINTER(C.c_int64),
                                     C.POINTER(GeneResult)]
    L.rvt_fam_binary_scale.restype = C.c_int
    L.rvt_fam_binary_scale.argtypes = [vp, C.c_int64, C.c_int64, c_double_p, c_double_p]
    L.rvt_run_fam_tests.restype = C.c_int
    L.rvt_run_fam_tests.argtypes = [vp, C.c_int, C.POINTER(vp), c_int_p, C.POINTER(C.c_int64), C.c_uint32,
                                    C.POINTER(GeneResult)]
    L.rvt_cov_rect.restype = C.c_int
    L.rvt_cov_rect.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, c_double_p, c_double_p, c_double_p, c_int_p]
    L.rvt_cov_rect_fam.restype = C.c_int
    L.rvt_cov_rect_fam.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, c_double_p, c_double_p, c_double_p, c_int_p]
    for f in (L.rvt_cov_band, L.rvt_cov_band_fam):
        f.restype = C.c_int
        f.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.POINTER(C.c_float), c_double_p,
                      c_double_p, c_int_p]
    L.rvt_block_copy_columns.restype = C.c_int
    L.rvt_block_copy_columns.argtypes = [vp, vp, C.c_int, vp, C.c_int, C.c_int]
    L.rvt_cov_block_fam.restype = C.c_int
    L.rvt_cov_block_fam.argtypes = [vp, vp, C.c_int, c_double_p, c_double_p, c_double_p, c_int_p]
    L.rvt_score_block.restype = C.c_int
    L.rvt_score_block.argtypes = [vp, vp, C.c_int, c_int_p] + [c_double_p] * 5
    L.rvt_null_summary.restype = C.c_int
    L.rvt_null_summary.argtypes = [vp, c_double_p, c_double_p, c_double_p]
    L.rvt_score_block_fam.restype = C.c_int
    L.rvt_score_block_fam.argtypes = [vp, vp, C.c_int, C.c_int, c_int_p] + [c_double_p] * 4
    L.rvt_fam_null_summary.restype = C.c_int
    L.rvt_fam_null_summary.argtypes = [vp, c_double_p]
    L.rvt_cov_block.restype = C.c_int
    L.rvt_cov_block.argtypes = [vp, vp, C.c_int, c_double_p, c_double_p, c_double_p, c_int_p]
    L.rvt_block_upload_columns.restype = C.c_int
    L.rvt_block_upload_columns.argtypes = [vp, vp, C.c_int, C.c_int, c_double_p]
    L.rvt_block_move_columns.restype = C.c_int
    L.rvt_block_move_columns.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int]
    gp = C.c_void_p
    L.rvt_group_init.restype = C.c_int
    L.rvt_group_init.argtypes = [C.POINTER(gp), C.c_int, c_int_p]
    L.rvt_group_destroy.restype = None
    L.rvt_group_destroy.argtypes = [gp]
    L.rvt_group_size.restype = C.c_int
    L.rvt_group_size.argtypes = [gp]
    L.rvt_group_member.restype = vp
    L.rvt_group_member.argtypes = [gp, C.c_int]
    L.rvt_group_last_error.restype = C.c_char_p
    L.rvt_group_last_error.argtypes = [gp]
    L.rvt_group_set_null.restype = C.c_int
    L.rvt_group_set_null.argtypes = [gp, C.c_int, C.c_int64, C.c_int, c_double_p, c_double_p, c_double_p, C.c_double]
    L.rvt_group_fit_null.restype = C.c_int
    L.rvt_group_fit_null.argtypes = [gp, C.c_int, C.c_int64, C.c_int, c_double_p, c_double_p, c_double_p, c_double_p]
    L.rvt_group_submit_gene.restype = C.c_int
    L.rvt_group_submit_gene.argtypes = [gp, C.c_int64, C.c_int, c_double_p, c_double_p, C.c_uint32, C.POINTER(Params)]
    L.rvt_group_submit_gene_raw.restype = C.c_int
    L.rvt_group_submit_gene_raw.argtypes = [gp, C.c_int64, C.c_int, c_double_p, C.c_uint32, C.POINTER(Params), c_double_p]
    L.rvt_group_submit_gene_i8.restype = C.c_int
    L.rvt_group_submit_gene_i8.argtypes = [gp, C.c_int64, C.c_int, C.POINTER(C.c_int8), C.c_uint32, C.POINTER(Params),
                                           c_double_p]
    c_i64_p = C.POINTER(C.c_int64)
    L.rvt_vcf_locate.restype = C.c_int
    L.rvt_vcf_locate.argtypes = [C.c_char_p, C.c_int64, c_i64_p, c_int_p, c_int_p, c_int_p]
    L.rvt_vcf_set_samples.restype = C.c_int
    L.rvt_vcf_set_samples.argtypes = [vp, C.c_int, c_int_p]
    L.rvt_vcf_set_filters.restype = C.c_int
    L.rvt_vcf_set_filters.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int]
    L.rvt_submit_gene_vcf.restype = C.c_int
    L.rvt_submit_gene_vcf.argtypes = [vp, C.c_int64, C.c_int, C.POINTER(C.c_char_p), c_i64_p, c_int_p, c_int_p, c_int_p,
                                      C.c_uint32, C.POINTER(Params), c_double_p]
    L.rvt_kinship_decompose.restype = C.c_int
    L.rvt_kinship_decompose.argtypes = [vp, C.c_int64, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float),
                                        C.c_int, C.POINTER(DecomposeInfo)]
    L.rvt_vcf_set_dosage.restype = C.c_int
    L.rvt_vcf_set_dosage.argtypes = [vp, C.c_int]
    L.rvt_vcf_format_index.restype = C.c_int
    L.rvt_vcf_format_index.argtypes = [C.c_char_p, C.c_int64, C.c_char_p, c_int_p]
    L.rvt_vcf_decode.restype = C.c_int
    L.rvt_vcf_decode.argtypes = [vp, C.c_int, C.POINTER(C.c_char_p), c_i64_p, c_int_p, c_int_p, c_int_p,
                                 C.POINTER(C.c_int8)]
    L.rvt_group_vcf_set_samples.restype = C.c_int
    L.rvt_group_vcf_set_samples.argtypes = [gp, C.c_int, c_int_p]
    L.rvt_group_vcf_set_filters.restype = C.c_int
    L.rvt_group_vcf_set_filters.argtypes = [gp, C.c_int, C.c_int, C.c_int, C.c_int]
    L.rvt_group_submit_gene_vcf.restype = C.c_int
    L.rvt_group_submit_gene_vcf.argtypes = [gp, C.c_int64, C.c_int, C.POINTER(C.c_char_p), c_i64_p, c_int_p, c_int_p,
                                            c_int_p, C.c_uint32, C.POINTER(Params), c_double_p]
    L.rvt_group_submit_gene_bed.restype = C.c_int
    L.rvt_group_submit_gene_bed.argtypes = [gp, C.c_int64, C.c_int, C.POINTER(C.c_uint8), C.c_uint32, C.POINTER(Params),
                                            c_double_p]
    L.rvt_group_host_register.restype = C.c_int
    L.rvt_group_host_register.argtypes = [gp, C.c_void_p, C.c_size_t]
    L.rvt_group_host_unregister.restype = C.c_int
    L.rvt_group_host_unregister.argtypes = [gp, C.c_void_p]
    L.rvt_group_set_content.restype = C.c_int
    L.rvt_group_set_content.argtypes = [gp, C.c_int, C.c_int]
    L.rvt_group_collect.restype = C.c_int
    L.rvt_group_collect.argtypes = [gp, C.POINTER(GeneResult), C.c_int, c_int_p]
    L.rvt_group_collect_ready.restype = C.c_int
    L.rvt_group_collect_ready.argtypes = [gp, C.POINTER(GeneResult), C.c_int, c_int_p]
    L.rvt_group_set_kinship.restype = C.c_int
    L.rvt_group_set_kinship.argtypes = [gp, C.c_int64, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.rvt_group_fit_fam_null.restype = C.c_int
    L.rvt_group_fit_fam_null.argtypes = [gp, C.c_int64, C.c_int, c_double_p, c_double_p, C.POINTER(FamNull)]
    L.rvt_group_run_fam_tests_host.restype = C.c_int
    L.rvt_group_run_fam_tests_host.argtypes = [gp, C.c_int, C.POINTER(c_double_p), c_int_p, C.POINTER(C.c_int64),
                                               C.c_uint32, C.POINTER(GeneResult)]
    L.rvt_set_profiling.restype = C.c_int
    L.rvt_set_profiling.argtypes = [vp, C.c_int]
    L.rvt_get_timing.restype = C.c_int
    L.rvt_get_timing.argtypes = [vp, C.POINTER(Timing), C.c_int]
    L.rvt_stream.restype = vp
    L.rvt_stream.argtypes = [vp]
    _lib = L
    return L


def _dp(a):
    return a.ctypes.data_as(c_double_p)


class KbacResult(C.Structure):
    _fields_ = [("fit_ok", C.c_int), ("n_poly", C.c_int), ("n_pattern", C.c_int), ("n_carrier", C.c_int),
                ("actual_perm", C.c_int), ("num_ge", C.c_int), ("num_le", C.c_int), ("stat", C.c_double),
                ("pvalue", C.c_double)]


class DecomposeInfo(C.Structure):
    _fields_ = [("sweeps", C.c_int), ("max_cosine", C.c_double), ("padded_order", C.c_int64), ("shift", C.c_double),
                ("max_residual", C.c_double)]


def vcf_locate(L, line):
    """(offset of the first sample column, GT index, GD index, GQ index) of one VCF record line (rvt_vcf_locate)."""
    off = C.c_int64(0)
    gt, gd, gq = C.c_int(-1), C.c_int(-1), C.c_int(-1)
    rc = L.rvt_vcf_locate(line, len(line), C.byref(off), C.byref(gt), C.byref(gd), C.byref(gq))
    if rc:
        raise ValueError("not a VCF record with sample columns")
    return off.value, gt.value, gd.value, gq.value


class Engine:
    """One engine context = one GPU (one process per GPU)."""

    def __init__(self, device=0):
        self.L = load_library()
        self.ctx = C.c_void_p()
        rc = self.L.rvt_init(C.byref(self.ctx), int(device))
        if rc != 0:
            raise RvtError("rvt_init failed (%d): no usable HIP device — the engine has no CPU fallback" % rc)
        self.N = None
        self.d = None
        self._blocks = []

    def _check(self, rc):
        if rc != 0:
            raise RvtError("rvtests_amd error %d: %s" % (rc, self.L.rvt_last_error(self.ctx).decode()))

    def close(self):
        if self.ctx:
            for b in self._blocks:
                self.L.rvt_block_free(self.ctx, b)
            self._blocks = []
            self.L.rvt_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- null model -----------------------------------------------------------------------------------
    def set_null(self, trait, X, res, v, sigma2=1.0):
        X = np.asfortranarray(X, dtype=np.float64)
        res = np.ascontiguousarray(res, dtype=np.float64)
        v = np.ascontiguousarray(v, dtype=np.float64)
        N, d = X.shape
        self._check(self.L.rvt_set_null(self.ctx, int(trait), N, d, _dp(X), _dp(res), _dp(v), float(sigma2)))
        self.N, self.d = N, d

    def fit_null(self, trait, X, y):
        """Fit the unrelated null model on the device and install it; returns (beta, sigma2)."""
        X = np.asfortranarray(X, dtype=np.float64)
        y = np.ascontiguousarray(y, dtype=np.float64)
        N, d = X.shape
        beta = np.zeros(d)
        s2 = C.c_double(0.0)
        self._check(self.L.rvt_fit_null(self.ctx, int(trait), N, d, _dp(X), _dp(y), _dp(beta),
                                        C.cast(C.byref(s2), c_double_p)))
        self.N, self.d = N, d
        return beta, s2.value

    def padded_ld(self, N=None):
        return int(self.L.rvt_padded_ld(int(self.N if N is None else N)))

    # ---- blocks -----------------------------------------------------------------------------------------
    def upload_block(self, G):
        """Copy a host N x M matrix into a device block; returns the device pointer (int)."""
        G = np.asfortranarray(G, dtype=np.float64)
        N, M = G.shape
        assert N == self.N
        p = C.c_void_p()
        self._check(self.L.rvt_block_alloc(self.ctx, M, C.byref(p)))
        self._check(self.L.rvt_block_upload(self.ctx, p, M, _dp(G)))
        self._blocks.append(p)
        return p.value

    def set_hardcall(self, on):
        """Tests / experiments: False keeps every gene on the fp64 kernel (rvt_set_hardcall)."""
        self._check(self.L.rvt_set_hardcall(self.ctx, 1 if on else 0))

    def hardcall_kernel(self):
        """Name of the kernel the hard-call genes of the installed null model take (rvt_hardcall_kernel), or None."""
        self.L.rvt_hardcall_kernel.restype = C.c_int
        self.L.rvt_hardcall_kernel.argtypes = [C.c_void_p]
        return {1: "gene_suffstat_hc", 2: "gene_suffstat_hcw", 3: "gene_suffstat_hcx"}.get(int(self.L.rvt_hardcall_kernel(self.ctx)))

    def set_content_hint(self, hint):
        """What the caller's own fp64 blocks hold: -1 unknown (start on the hard-call kernel), 0 dosages (start on the fp64
        kernel), 1 hard calls / mean-imputed hard calls (rvt_set_content_hint).  Never affects correctness."""
        self._check(self.L.rvt_set_content_hint(self.ctx, int(hint)))

    def host_register(self, arr):
        """Page-lock a numpy array the caller reuses as its hand-off buffer (rvt_host_register): submissions whose source
        lies inside it are DMA straight from it.  Keep the array alive until host_unregister."""
        self._check(self.L.rvt_host_register(self.ctx, C.c_void_p(arr.ctypes.data), arr.nbytes))

    def host_unregister(self, arr):
        self._check(self.L.rvt_host_unregister(self.ctx, C.c_void_p(arr.ctypes.data)))

    def set_dosage_lattice(self, denominator):
        """The dosage doubles are multiples of 1 / denominator rounded to double (VCF DS fields with a fixed number of
        decimals: 1000 for three); 0 = not stated (rvt_set_dosage_lattice).  Never affects correctness."""
        self._check(self.L.rvt_set_dosage_lattice(self.ctx, int(denominator)))

    def set_dosage_float(self, on=True):
        """Blocks of unknown content hold float-precision dosages (BGEN-style values: multiples of 2^-37): the float-digit
        int8 kernel is tried first for M <= 64 (rvt_set_dosage_float).  Never affects correctness."""
        self._check(self.L.rvt_set_dosage_float(self.ctx, 1 if on else 0))

    def classify_block(self, ptr, M):
        """Query (nothing is remembered): does the device block hold hard calls only (rvt_block_classify)?"""
        flag = C.c_int(0)
        self._check(self.L.rvt_block_classify(self.ctx, C.c_void_p(int(ptr)), int(M), C.byref(flag)))
        return bool(flag.value)

    def alloc_block(self, M):
        """Zeroed device block of M columns (rvt_block_alloc); fill it with upload_columns."""
        p = C.c_void_p()
        self._check(self.L.rvt_block_alloc(self.ctx, int(M), C.byref(p)))
        self._blocks.append(p)
        return p.value

    def free_block(self, ptr):
        self._check(self.L.rvt_block_free(self.ctx, C.c_void_p(ptr)))
        self._blocks = [b for b in self._blocks if b.value != ptr]

    # ---- batched run over device-resident blocks ---------------------------------------------------------
    def _pack(self, ptrs, Ms, afs, ids):
        n = len(ptrs)
        arr_p = (C.c_void_p * n)(*[C.c_void_p(int(p)) for p in ptrs])
        arr_m = np.ascontiguousarray(Ms, dtype=np.int32)
        af = np.ascontiguousarray(np.concatenate([np.asarray(a, dtype=np.float64) for a in afs]))
        assert af.size == int(arr_m.sum())
        arr_id = np.ascontiguousarray(ids if ids is not None else np.arange(n), dtype=np.int64)
        return n, arr_p, arr_m, af, arr_id

    def run_blocks(self, ptrs, Ms, afs, tests=TEST_ALL, params=None, ids=None):
        n, arr_p, arr_m, af, arr_id = self._pack(ptrs, Ms, afs, ids)
        out = (GeneResult * n)()
        prm = params or Params.default()
        self._check(self.L.rvt_run_blocks(self.ctx, n, arr_p, arr_m.ctypes.data_as(c_int_p), _dp(af),
                                          arr_id.ctypes.data_as(C.POINTER(C.c_int64)), int(tests), C.byref(prm), out))
        return list(out)

    def prepare(self, ptrs, Ms, afs, tests=TEST_ALL, params=None, ids=None):
        """Pre-pack a batch so that repeated launches (bench) pay no Python packing cost."""
        n, arr_p, arr_m, af, arr_id = self._pack(ptrs, Ms, afs, ids)
        out = (GeneResult * n)()
        prm = params or Params.default()
        return dict(n=n, p=arr_p, m=arr_m, af=af, id=arr_id, out=out, prm=prm, tests=int(tests))

    def reserve(self, Ms):
        """Allocate the workspace of every pipeline slot for batches shaped like Ms up front (rvt_reserve)."""
        arr_m = np.ascontiguousarray(Ms, dtype=np.int32)
        self._check(self.L.rvt_reserve(self.ctx, len(arr_m), arr_m.ctypes.data_as(c_int_p)))

    def launch(self, batch):
        self._check(self.L.rvt_run_blocks_async(self.ctx, batch["n"], batch["p"], batch["m"].ctypes.data_as(c_int_p),
                                                _dp(batch["af"]), batch["id"].ctypes.data_as(C.POINTER(C.c_int64)),
                                                batch["tests"], C.byref(batch["prm"]), batch["out"]))

    def sync(self):
        self._check(self.L.rvt_sync(self.ctx))

    def wait_oldest(self):
        self._check(self.L.rvt_wait_oldest(self.ctx))

    # ---- streaming (ModelFitter-style) interface ----------------------------------------------------------
    def submit_gene(self, gene_id, G, af, tests=TEST_ALL, params=None):
        G = np.asfortranarray(G, dtype=np.float64)
        af = np.ascontiguousarray(af, dtype=np.float64)
        prm = params or Params.default()
        self._check(self.L.rvt_submit_gene(self.ctx, int(gene_id), G.shape[1], _dp(G), _dp(af), int(tests),
                                           C.byref(prm)))

    @staticmethod
    def pack_bed(Graw):
        """PLINK .bed SNP-major packing of an N x M matrix of hard calls (negative = missing): M rows of ceil(N/4)
        bytes, 00 -> 0, 10 -> 1, 11 -> 2, 01 -> missing (what PlinkInputFile reads)."""
        G = np.asarray(Graw)
        N, M = G.shape
        code = np.where(G < 0, 1, np.where(G == 0, 0, np.where(G == 1, 2, 3))).astype(np.uint8)
        pad = (-N) % 4
        if pad:
            code = np.vstack([code, np.zeros((pad, M), dtype=np.uint8)])
        c4 = code.T.reshape(M, -1, 4)
        return np.ascontiguousarray(c4[:, :, 0] | (c4[:, :, 1] << 2) | (c4[:, :, 2] << 4) | (c4[:, :, 3] << 6))

    def submit_gene_bed(self, gene_id, bed, M, tests=TEST_ALL, params=None, want_af=True):
        """PLINK 2-bit codes (pack_bed layout) -> rvt_submit_gene_bed; returns the allele frequencies used (None with
        want_af=False: the call then does not wait for the device)."""
        prm = params or Params.default()
        bed = np.ascontiguousarray(bed, dtype=np.uint8)
        af = np.zeros(M) if want_af else None
        self._check(self.L.rvt_submit_gene_bed(self.ctx, int(gene_id), int(M), bed.ctypes.data_as(C.POINTER(C.c_uint8)),
                                               int(tests), C.byref(prm), _dp(af) if want_af else None))
        return af

    # ---- a .bed matrix resident in device memory (rvt_bed_alloc / rvt_bed_upload / rvt_submit_gene_bed_dev) ------------
    def bed_alloc(self, n_variants):
        p = C.c_void_p()
        self.L.rvt_bed_alloc.restype = C.c_int
        self.L.rvt_bed_alloc.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_void_p)]
        self._check(self.L.rvt_bed_alloc(self.ctx, int(n_variants), C.byref(p)))
        return p.value

    def bed_upload(self, d_bed, first_variant, rows):
        """rows: (n, ceil(N/4)) uint8, pack_bed layout."""
        rows = np.ascontiguousarray(rows, dtype=np.uint8)
        self.L.rvt_bed_upload.restype = C.c_int
        self.L.rvt_bed_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
        self._check(self.L.rvt_bed_upload(self.ctx, C.c_void_p(int(d_bed)), int(first_variant), rows.shape[0],
                                          rows.ctypes.data_as(C.c_void_p)))

    def bed_free(self, d_bed):
        self.L.rvt_bed_free.restype = C.c_int
        self.L.rvt_bed_free.argtypes = [C.c_void_p, C.c_void_p]
        self._check(self.L.rvt_bed_free(self.ctx, C.c_void_p(int(d_bed))))

    def submit_gene_bed_dev(self, gene_id, d_rows, M, tests=TEST_ALL, params=None, want_af=True):
        """M consecutive rows of a resident .bed matrix, d_rows = device address of the first (rvt_submit_gene_bed_dev)."""
        prm = params or Params.default()
        af = np.zeros(M) if want_af else None
        self.L.rvt_submit_gene_bed_dev.restype = C.c_int
        self.L.rvt_submit_gene_bed_dev.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, c_double_p]
        self._check(self.L.rvt_submit_gene_bed_dev(self.ctx, int(gene_id), int(M), C.c_void_p(int(d_rows)), int(tests),
                                                   C.byref(prm), _dp(af) if want_af else None))
        return af

    def score_bed_dev(self, d_rows, V, want_counts=True):
        """Single-variant score tests of V consecutive rows of a resident .bed matrix (rvt_score_bed_dev):
        (ok, ustat, vstat, effect, se, pvalue[, counts (V, 4): n0 n1 n2 missing])."""
        ok = np.zeros(V, dtype=np.int32)
        outs = [np.zeros(V) for _ in range(5)]
        cnt = np.zeros((V, 4), dtype=np.int64) if want_counts else None
        self.L.rvt_score_bed_dev.restype = C.c_int
        self.L.rvt_score_bed_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p] + [c_double_p] * 5 + [C.c_void_p]
        self._check(self.L.rvt_score_bed_dev(self.ctx, C.c_void_p(int(d_rows)), int(V), ok.ctypes.data_as(C.c_void_p),
                                             *[_dp(o) for o in outs], cnt.ctypes.data_as(C.c_void_p) if want_counts else None))
        return (ok,) + tuple(outs) + ((cnt,) if want_counts else ())

    def submit_genes_bed_dev(self, gene_ids, d_rows, Ms, tests=TEST_ALL, params=None):
        """Several genes of a resident .bed matrix in one call (rvt_submit_genes, kind 7): d_rows[g] = device address of the
        first row of gene g."""
        n = len(d_rows)
        prm = params or Params.default()
        ids = np.ascontiguousarray(gene_ids, dtype=np.int64)
        ms = np.ascontiguousarray(Ms, dtype=np.int32)
        ptrs = (C.c_void_p * n)(*[C.c_void_p(int(a)) for a in d_rows])
        self.L.rvt_submit_genes.restype = C.c_int
        self.L.rvt_submit_genes.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                                            C.c_void_p]
        self._check(self.L.rvt_submit_genes(self.ctx, 7, n, ids.ctypes.data_as(C.c_void_p), ms.ctypes.data_as(C.c_void_p), ptrs,
                                            int(tests), C.byref(prm)))

    def submit_genes(self, kind, gene_ids, arrays, Ms, tests=TEST_ALL, params=None):
        """Several genes in ONE call (rvt_submit_genes): kind 1 = doubles with missing codes (N x M, Fortran order), 2 = int8
        (N x M, Fortran order), 3 = PLINK 2-bit rows (pack_bed layout).  The arrays must be contiguous and stay alive until
        the call returns.  No allele frequencies are returned."""
        n = len(arrays)
        prm = params or Params.default()
        ids = np.ascontiguousarray(gene_ids, dtype=np.int64)
        ms = np.ascontiguousarray(Ms, dtype=np.int32)
        ptrs = (C.c_void_p * n)(*[C.c_void_p(a.ctypes.data) for a in arrays])
        self.L.rvt_submit_genes.restype = C.c_int
        self.L.rvt_submit_genes.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                                            C.c_void_p]
        self._check(self.L.rvt_submit_genes(self.ctx, int(kind), n, ids.ctypes.data_as(C.c_void_p),
                                            ms.ctypes.data_as(C.c_void_p), ptrs, int(tests), C.byref(prm)))

    def fam_analytic_vt(self, ptrs, Ms):
        """FamAnalyticVT of device-resident raw blocks (rvt_fam_analytic_vt); vt_* fields of the records."""
        n = len(ptrs)
        arr_p = (C.c_void_p * n)(*[C.c_void_p(int(p)) for p in ptrs])
        arr_m = np.ascontiguousarray(Ms, dtype=np.int32)
        out = (GeneResult * n)()
        self.L.rvt_fam_analytic_vt.restype = C.c_int
        self._check(self.L.rvt_fam_analytic_vt(self.ctx, n, arr_p, arr_m.ctypes.data_as(c_int_p), out))
        return list(out)

    def kbac_blocks(self, ptrs, Ms, afs, y, nperm, alpha):
        """KBAC of device-resident blocks (rvt_kbac_blocks); afs: list of per-gene frequency arrays."""
        n = len(ptrs)
        arr_p = (C.c_void_p * n)(*[C.c_void_p(int(p)) for p in ptrs])
        arr_m = np.ascontiguousarray(Ms, dtype=np.int32)
        af = np.ascontiguousarray(np.concatenate([np.asarray(a, dtype=np.float64) for a in afs]))
        y = np.ascontiguousarray(y, dtype=np.float64)
        out = (KbacResult * n)()
        self.L.rvt_kbac_blocks.restype = C.c_int
        self._check(self.L.rvt_kbac_blocks(self.ctx, n, arr_p, arr_m.ctypes.data_as(c_int_p), _dp(af), _dp(y), int(nperm),
                                           C.c_double(alpha), out))
        return list(out)

    def kinship_decompose(self, K, install=False, want_vectors=True):
        """Eigendecomposition of the symmetric float kinship K on the device (rvt_kinship_decompose).
        Returns (U float32 N x N column-major or None, S float32 ascending, DecomposeInfo)."""
        K = np.asfortranarray(K, dtype=np.float32)
        N = K.shape[0]
        U = np.zeros((N, N), dtype=np.float32, order="F") if want_vectors else None
        S = np.zeros(N, dtype=np.float32)
        info = DecomposeInfo()
        fp = C.POINTER(C.c_float)
        self._check(self.L.rvt_kinship_decompose(self.ctx, N, K.ctypes.data_as(fp),
                                                 U.ctypes.data_as(fp) if want_vectors else None, S.ctypes.data_as(fp),
                                                 1 if install else 0, C.byref(info)))
        if install:
            self.N = N
        return U, S, info

    def vcf_set_samples(self, row_of_sample):
        rows = np.ascontiguousarray(row_of_sample, dtype=np.int32)
        self._check(self.L.rvt_vcf_set_samples(self.ctx, len(rows), rows.ctypes.data_as(c_int_p)))

    def vcf_set_filters(self, gd_min=0, gd_max=0, gq_min=0, gq_max=0):
        self._check(self.L.rvt_vcf_set_filters(self.ctx, int(gd_min), int(gd_max), int(gq_min), int(gq_max)))

    def _vcf_args(self, lines):
        """(M, keep-alive, text[], len[], gt[], gd[], gq[]) for the records `lines` (bytes, no newline); the text pointers
        point INTO the caller's bytes objects (no copy).  A tuple returned earlier is accepted as is."""
        if isinstance(lines, tuple):
            return lines
        M = len(lines)
        text = (C.c_char_p * M)()
        addr = C.cast(text, C.POINTER(C.c_void_p))
        tlen = (C.c_int64 * M)()
        gt = (C.c_int * M)()
        gd = (C.c_int * M)()
        gq = (C.c_int * M)()
        for j, ln in enumerate(lines):
            off, a, b, c_ = vcf_locate(self.L, ln)
            addr[j] = C.cast(C.c_char_p(ln), C.c_void_p).value + off
            tlen[j] = len(ln) - off
            gt[j], gd[j], gq[j] = a, b, c_
        return M, list(lines), text, tlen, gt, gd, gq

    prepare_vcf = _vcf_args

    def vcf_set_alt_alleles(self, alt):
        """Multi-allelic mode for the NEXT vcf_decode / submit_gene_vcf call: alt[j] > 0 counts that allele in record j."""
        a = np.ascontiguousarray(alt, dtype=np.int32)
        self.L.rvt_vcf_set_alt_alleles.restype = C.c_int
        self._check(self.L.rvt_vcf_set_alt_alleles(self.ctx, len(a), a.ctypes.data_as(c_int_p)))

    def vcf_set_sex(self, sex):
        """PLINK sex code (1 male, 2 female, else unknown) of every FILE sample; used in hemizygous records only."""
        a = np.ascontiguousarray(sex, dtype=np.int8)
        self.L.rvt_vcf_set_sex.restype = C.c_int
        self.L.rvt_vcf_set_sex.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        self._check(self.L.rvt_vcf_set_sex(self.ctx, len(a), a.ctypes.data))

    def vcf_set_hemi(self, hemi):
        """hemi[j] != 0: record j of the NEXT vcf_decode / submit_gene_vcf call lies in a hemizygous region."""
        a = np.ascontiguousarray(hemi, dtype=np.int32)
        self.L.rvt_vcf_set_hemi.restype = C.c_int
        self.L.rvt_vcf_set_hemi.argtypes = [C.c_void_p, C.c_int, c_int_p]
        self._check(self.L.rvt_vcf_set_hemi(self.ctx, len(a), a.ctypes.data_as(c_int_p)))

    def vcf_set_dosage(self, on=True):
        self._check(self.L.rvt_vcf_set_dosage(self.ctx, 1 if on else 0))

    def vcf_args_for_tag(self, lines, tag):
        """As _vcf_args, with the FORMAT index of `tag` (bytes) in place of GT's."""
        M, keep, text, tlen, gt, gd, gq = self._vcf_args(lines)
        for j, ln in enumerate(lines):
            idx = C.c_int(-1)
            self.L.rvt_vcf_format_index(ln, len(ln), tag, C.byref(idx))
            gt[j] = idx.value
        return M, keep, text, tlen, gt, gd, gq

    def vcf_decode_dosage(self, lines, tag, n_rows):
        """Dosage matrix (n_rows x M float64, missing = -9) the device reads out of the records' text."""
        M, keep, text, tlen, gt, gd, gq = self.vcf_args_for_tag(lines, tag)
        out = np.zeros((n_rows, M), dtype=np.float64, order="F")
        self.L.rvt_vcf_decode_dosage.restype = C.c_int
        self._check(self.L.rvt_vcf_decode_dosage(self.ctx, M, text, tlen, gt, gd, gq, _dp(out)))
        return out

    def vcf_decode(self, lines, n_rows):
        """Genotype bytes (n_rows x M int8, missing = -9) the device reads out of the records' text."""
        M, keep, text, tlen, gt, gd, gq = self._vcf_args(lines)
        out = np.zeros((n_rows, M), dtype=np.int8, order="F")
        self._check(self.L.rvt_vcf_decode(self.ctx, M, text, tlen, gt, gd, gq, out.ctypes.data_as(C.POINTER(C.c_int8))))
        return out

    def submit_gene_vcf_dosage(self, gene_id, lines, tag, tests=TEST_ALL, params=None, want_af=True):
        """As submit_gene_vcf in dosage mode (vcf_set_dosage(True) first): values of FORMAT tag `tag` through atof."""
        return self.submit_gene_vcf(gene_id, self.vcf_args_for_tag(lines, tag), tests, params, want_af)

    def submit_gene_vcf(self, gene_id, lines, tests=TEST_ALL, params=None, want_af=True):
        """lines: the full text of the gene's VCF records (bytes, no newline).  rvt_vcf_locate finds the sample columns
        and the FORMAT indices on the host; the device decodes the genotypes (rvt_submit_gene_vcf)."""
        prm = params or Params.default()
        M, keep, text, tlen, gt, gd, gq = self._vcf_args(lines)
        af = np.zeros(M) if want_af else None
        self._check(self.L.rvt_submit_gene_vcf(self.ctx, int(gene_id), M, text, tlen, gt, gd, gq, int(tests),
                                               C.byref(prm), _dp(af) if want_af else None))
        return af

    # ---- BGEN genotype-probability blocks (rvt_submit_gene_bgen / rvt_bgen_decode) ------------------------------------
    @staticmethod
    def _bgen_args(blocks):
        """blocks: the UNCOMPRESSED probability block of each variant (bytes-like).  ctypes pointer arrays over them."""
        M = len(blocks)
        keep = [np.frombuffer(b, dtype=np.uint8) for b in blocks]           # zero-copy views; kept alive by the caller
        ptr = (C.c_void_p * M)(*[k.ctypes.data for k in keep])
        blen = (C.c_int64 * M)(*[k.size for k in keep])
        return M, keep, ptr, blen

    def bgen_decode(self, blocks, layout, n_rows):
        """Raw genotype matrix (n_rows x M float64, missing = -9) the device reads out of the blocks."""
        M, keep, ptr, blen = self._bgen_args(blocks)
        out = np.zeros((n_rows, M), dtype=np.float64, order="F")
        self.L.rvt_bgen_decode.restype = C.c_int
        self.L.rvt_bgen_decode.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int64, c_double_p]
        self._check(self.L.rvt_bgen_decode(self.ctx, M, ptr, blen, int(layout), int(n_rows), _dp(out)))
        return out

    def submit_gene_bgen(self, gene_id, blocks, layout, tests=TEST_ALL, params=None, want_af=True):
        """One gene as BGEN probability blocks: dosages, allele frequencies and mean imputation on the device."""
        prm = params or Params.default()
        M, keep, ptr, blen = self._bgen_args(blocks)
        af = np.zeros(M) if want_af else None
        self.L.rvt_submit_gene_bgen.restype = C.c_int
        self.L.rvt_submit_gene_bgen.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_uint32,
                                                C.c_void_p, c_double_p]
        self._check(self.L.rvt_submit_gene_bgen(self.ctx, int(gene_id), M, ptr, blen, int(layout), int(tests),
                                                C.byref(prm), _dp(af) if want_af else None))
        return af

    def submit_gene_raw(self, gene_id, Graw, tests=TEST_ALL, params=None, want_af=True):
        """Raw extractor output (missing < 0): float64 -> rvt_submit_gene_raw, int8 -> rvt_submit_gene_i8.  The device
        imputes and counts allele frequencies; returns the frequencies the tests will use (None with want_af=False)."""
        prm = params or Params.default()
        if Graw.dtype == np.int8:
            G = np.asfortranarray(Graw)
            af = np.zeros(G.shape[1]) if want_af else None
            self._check(self.L.rvt_submit_gene_i8(self.ctx, int(gene_id), G.shape[1],
                                                  G.ctypes.data_as(C.POINTER(C.c_int8)), int(tests), C.byref(prm),
                                                  _dp(af) if want_af else None))
        else:
            G = np.asfortranarray(Graw, dtype=np.float64)
            af = np.zeros(G.shape[1]) if want_af else None
            self._check(self.L.rvt_submit_gene_raw(self.ctx, int(gene_id), G.shape[1], _dp(G), int(tests),
                                                   C.byref(prm), _dp(af) if want_af else None))
        return af

    def collect(self, cap=4096):
        out = (GeneResult * cap)()
        n = C.c_int(0)
        self._check(self.L.rvt_collect(self.ctx, out, cap, C.byref(n)))
        return list(out[: n.value])

    def collect_ready(self, cap=4096):
        """The finished prefix of the submitted genes, without waiting (rvt_collect_ready); may be empty."""
        out = (GeneResult * cap)()
        n = C.c_int(0)
        self._check(self.L.rvt_collect_ready(self.ctx, out, cap, C.byref(n)))
        return list(out[: n.value])

    # ---- inspection ------------------------------------------------------------------------------------------
    def debug_collapse(self, ptr, M):
        cmc = np.zeros(self.N)
        zeg = np.zeros(self.N)
        fl = np.zeros(M, dtype=np.int32)
        kp = np.zeros(M, dtype=np.int32)
        self._check(self.L.rvt_debug_collapse(self.ctx, C.c_void_p(int(ptr)), M, _dp(cmc), _dp(zeg),
                                              fl.ctypes.data_as(c_int_p), kp.ctypes.data_as(c_int_p)))
        return cmc, zeg, fl, kp

    def debug_suffstat(self, ptr, M):
        d = self.d
        S = np.zeros((M, M))
        T = np.zeros((M, d))
        u = np.zeros(M)
        cs = np.zeros(M)
        mn = np.zeros(M)
        mx = np.zeros(M)
        self._check(self.L.rvt_debug_suffstat(self.ctx, C.c_void_p(int(ptr)), M, _dp(S), _dp(T), _dp(u), _dp(cs),
                                              _dp(mn), _dp(mx)))
        return S, T, u, cs, mn, mx

    def set_perm_exact(self, on):
        """True (the default of a single context): SKAT permutations replay the reference's rand() stream (bit-identical
        counters, sequential); False: counter-based permutations keyed by (seed, gene id, shuffle) — rvt_set_perm_exact."""
        self.L.rvt_set_perm_exact.restype = C.c_int
        self.L.rvt_set_perm_exact.argtypes = [C.c_void_p, C.c_int]
        self._check(self.L.rvt_set_perm_exact(self.ctx, 1 if on else 0))

    def rand_seed(self, seed=1):
        """Restart the emulated glibc rand() stream the SKAT permutations draw from (srand semantics)."""
        self._check(self.L.rvt_rand_seed(self.ctx, int(seed)))

    # ---- related samples: FastLMM null + FamSKAT -----------------------------------------------------------
    def set_kinship(self, U, S):
        U = np.asfortranarray(U, dtype=np.float32)
        S = np.ascontiguousarray(S, dtype=np.float32)
        N = U.shape[0]
        assert U.shape == (N, N) and S.shape == (N,)
        fp = C.POINTER(C.c_float)
        self._check(self.L.rvt_set_kinship(self.ctx, N, U.ctypes.data_as(fp), S.ctypes.data_as(fp)))
        self.N = N

    def kinship_structure(self):
        """Share of the N x N rotation product that is computed (1 = dense eigenvectors; small for family structure)."""
        f = C.c_double(0.0)
        self.L.rvt_kinship_structure.restype = C.c_int
        self.L.rvt_kinship_structure.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        self._check(self.L.rvt_kinship_structure(self.ctx, C.byref(f)))
        return f.value

    def fit_fam_null(self, X, y):
        X = np.asfortranarray(X, dtype=np.float64)
        y = np.ascontiguousarray(y, dtype=np.float64)
        N, d = X.shape
        out = FamNull()
        self._check(self.L.rvt_fit_fam_null(self.ctx, N, d, _dp(X), _dp(y), C.byref(out)))
        self.d = getattr(self, "d", d)
        return out

    def run_fam_blocks(self, ptrs, Ms, ids=None, tests=16):
        """tests: TEST_FAMSKAT (16) | TEST_FAMCMC (32) | TEST_FAMZEGGINI (64)"""
        n = len(ptrs)
        arr_p = (C.c_void_p * n)(*[C.c_void_p(int(p)) for p in ptrs])
        arr_m = np.ascontiguousarray(Ms, dtype=np.int32)
        arr_id = np.ascontiguousarray(ids if ids is not None else np.arange(n), dtype=np.int64)
        out = (GeneResult * n)()
        self._check(self.L.rvt_run_fam_tests(self.ctx, n, arr_p, arr_m.ctypes.data_as(c_int_p),
                                             arr_id.ctypes.data_as(C.POINTER(C.c_int64)), int(tests), out))
        return list(out)

    # ---- MetaCov --------------------------------------------------------------------------------------------
    def cov_block(self, ptr, V):
        """Covariance band of one device block of V variants: (cov V x V with cov[h, j] valid for j >= h, xz V x d,
        zz d x d, polymorphic flags)."""
        d = self.d
        cov = np.full((V, V), np.nan, order="F")     # cov[h + j*V]
        xz = np.zeros((V, d))
        zz = np.zeros((d, d))
        poly = np.zeros(V, dtype=np.int32)
        self._check(self.L.rvt_cov_block(self.ctx, C.c_void_p(int(ptr)), V, _dp(cov), _dp(xz), _dp(zz),
                                         poly.ctypes.data_as(c_int_p)))
        return cov, xz, zz, poly

    def score_block(self, ptr, V):
        """MetaScore statistics of the V columns of a device block: dict of ok, U, V, effect, se, p arrays."""
        ok = np.zeros(V, dtype=np.int32)
        arr = [np.zeros(V) for _ in range(5)]
        self._check(self.L.rvt_score_block(self.ctx, C.c_void_p(int(ptr)), int(V), ok.ctypes.data_as(c_int_p),
                                           *[_dp(a) for a in arr]))
        return dict(ok=ok, U=arr[0], V=arr[1], effect=arr[2], se=arr[3], p=arr[4])

    def score_block_fam(self, ptr, V, binary=0):
        """MetaFamQtl (binary=1: MetaFamBinary) statistics of the V raw columns of a device block (after set_kinship +
        fit_fam_null [+ fam_binary_scale])."""
        ok = np.zeros(V, dtype=np.int32)
        arr = [np.zeros(V) for _ in range(4)]
        self._check(self.L.rvt_score_block_fam(self.ctx, C.c_void_p(int(ptr)), int(V), int(binary),
                                               ok.ctypes.data_as(c_int_p),
                                               *[_dp(a) for a in arr]))
        return dict(ok=ok, U=arr[0], V=arr[1], af=arr[2], p=arr[3])

    def fam_null_summary(self, d):
        covb = np.zeros(d)
        self._check(self.L.rvt_fam_null_summary(self.ctx, _dp(covb)))
        return covb

    def null_summary(self):
        """(beta, diag covB, sigma2) of the installed null model, as MetaScoreTest::PrintNullModel prints them."""
        beta, covb = np.zeros(self.d), np.zeros(self.d)
        s2 = C.c_double(0)
        self._check(self.L.rvt_null_summary(self.ctx, _dp(beta), _dp(covb), C.cast(C.byref(s2), c_double_p)))
        return beta, covb, s2.value

    def fam_binary_scale(self, n_case, n_ctrl):
        a, b = C.c_double(0), C.c_double(0)
        self._check(self.L.rvt_fam_binary_scale(self.ctx, int(n_case), int(n_ctrl), C.cast(C.byref(a), c_double_p),
                                                C.cast(C.byref(b), c_double_p)))
        return a.value, b.value

    def cov_rect(self, ptr, col0, H, W):
        """Heads [col0, col0+H) against markers [col0, col0+W): (cov H x W with cov[h, j] valid for j >= h, xz W x d,
        zz, polymorphic flags of the W markers)."""
        d = self.d
        cov = np.full((H, W), np.nan, order="F")
        xz = np.zeros((W, d))
        zz = np.zeros((d, d))
        poly = np.zeros(W, dtype=np.int32)
        self._check(self.L.rvt_cov_rect(self.ctx, C.c_void_p(int(ptr)), int(col0), int(H), int(W), _dp(cov), _dp(xz),
                                        _dp(zz), poly.ctypes.data_as(c_int_p)))
        return cov, xz, zz, poly

    def cov_band(self, ptr, ring, col0, H, W, halo, scale=1.0, band=None, fam_d=None):
        """The band of a sliding window on a block used as a ring (rvt_cov_band; fam_d = columns of X: rvt_cov_band_fam):
        logical column j = physical (col0 + j) mod ring (ring = 0: a linear range).  Returns (band H x (halo + 1) float32 with
        band[h, t] = float(value(h, h + t)) * scale, NaN beyond the window; xz W x d; zz; polymorphic flags of the W markers).
        `band`: an optional float32 array to write into (e.g. one registered with host_register)."""
        d = self.d if fam_d is None else fam_d
        W = min(W, H + halo)
        if band is None:
            band = np.empty((H, halo + 1), dtype=np.float32)
        assert band.dtype == np.float32 and band.flags.c_contiguous and band.size >= H * (halo + 1)
        xz = np.zeros((W, d))
        zz = np.zeros((d, d))
        poly = np.zeros(W, dtype=np.int32)
        fn = self.L.rvt_cov_band if fam_d is None else self.L.rvt_cov_band_fam
        self._check(fn(self.ctx, C.c_void_p(int(ptr)), int(ring), int(col0), int(H), int(W), int(halo), float(scale),
                       band.ctypes.data_as(C.POINTER(C.c_float)), _dp(xz), _dp(zz), poly.ctypes.data_as(c_int_p)))
        return band.reshape(-1)[:H * (halo + 1)].reshape(H, halo + 1), xz, zz, poly

    def cov_band_last_path(self):
        """0 fp64 product, 1 MXFP4 band on the column cache, 4 the same for mean-imputed columns (four products), 2 MXFP4 band on a
        copy made in the call, 11 / 12 the int8 instruction (rvt_cov_band_last_path)."""
        self.L.rvt_cov_band_last_path.restype = C.c_int
        self.L.rvt_cov_band_last_path.argtypes = [C.c_void_p]
        return int(self.L.rvt_cov_band_last_path(self.ctx))

    def cov_rect_fam(self, ptr, col0, H, W, d):
        """Family-mode heads x window rectangle (after set_kinship + fit_fam_null); d = columns of X."""
        cov = np.full((H, W), np.nan, order="F")
        xz = np.zeros((W, d))
        zz = np.zeros((d, d))
        poly = np.zeros(W, dtype=np.int32)
        self._check(self.L.rvt_cov_rect_fam(self.ctx, C.c_void_p(int(ptr)), int(col0), int(H), int(W), _dp(cov), _dp(xz),
                                            _dp(zz), poly.ctypes.data_as(c_int_p)))
        return cov, xz, zz, poly

    def cov_block_fam(self, ptr, V, d):
        """Family-mode covariance band (after set_kinship + fit_fam_null); d = columns of X."""
        cov = np.full((V, V), np.nan, order="F")
        xz = np.zeros((V, d))
        zz = np.zeros((d, d))
        poly = np.zeros(V, dtype=np.int32)
        self._check(self.L.rvt_cov_block_fam(self.ctx, C.c_void_p(int(ptr)), V, _dp(cov), _dp(xz), _dp(zz),
                                             poly.ctypes.data_as(c_int_p)))
        return cov, xz, zz, poly

    def upload_columns(self, ptr, col0, G):
        G = np.asfortranarray(G, dtype=np.float64)
        if G.ndim == 1:
            G = G.reshape(-1, 1, order="F")
        self._check(self.L.rvt_block_upload_columns(self.ctx, C.c_void_p(int(ptr)), int(col0), G.shape[1], _dp(G)))

    def move_columns(self, ptr, dst, src, n):
        self._check(self.L.rvt_block_move_columns(self.ctx, C.c_void_p(int(ptr)), int(dst), int(src), int(n)))

    def set_profiling(self, on):
        self._check(self.L.rvt_set_profiling(self.ctx, 1 if on else 0))

    def timing(self, reset=False):
        t = Timing()
        self._check(self.L.rvt_get_timing(self.ctx, C.byref(t), 1 if reset else 0))
        return t


class Group:
    """Several GPUs behind one caller (rvt_group_*): the null model on every member, the gene stream dealt in runs of 32,
    records back in submission order.  `devices`: list of HIP device indices (a device may repeat)."""

    def __init__(self, devices):
        self.L = load_library()
        self.g = C.c_void_p()
        ids = np.ascontiguousarray(devices, dtype=np.int32)
        rc = self.L.rvt_group_init(C.byref(self.g), len(ids), ids.ctypes.data_as(c_int_p))
        if rc != 0:
            raise RvtError("rvt_group_init failed (%d): no usable HIP device — the engine has no CPU fallback" % rc)

    def _check(self, rc):
        if rc != 0:
            raise RvtError("rvtests_amd error %d: %s" % (rc, self.L.rvt_group_last_error(self.g).decode()))

    def close(self):
        if self.g:
            self.L.rvt_group_destroy(self.g)
            self.g = C.c_void_p()

    def host_register(self, arr):
        self._check(self.L.rvt_group_host_register(self.g, C.c_void_p(arr.ctypes.data), arr.nbytes))

    def host_unregister(self, arr):
        self._check(self.L.rvt_group_host_unregister(self.g, C.c_void_p(arr.ctypes.data)))

    def set_content(self, hint, lattice_denominator=0):
        """What the caller's fp64 blocks hold, on every member (rvt_group_set_content): hint as Engine.set_content_hint,
        lattice_denominator as Engine.set_dosage_lattice."""
        self._check(self.L.rvt_group_set_content(self.g, int(hint), int(lattice_denominator)))

    def fit_null(self, trait, X, y):
        X = np.asfortranarray(X, dtype=np.float64)
        y = np.ascontiguousarray(y, dtype=np.float64)
        N, d = X.shape
        beta = np.zeros(d)
        s2 = C.c_double(0.0)
        self._check(self.L.rvt_group_fit_null(self.g, int(trait), N, d, _dp(X), _dp(y), _dp(beta),
                                              C.cast(C.byref(s2), c_double_p)))
        return beta, s2.value

    def submit_gene(self, gene_id, G, af, tests=TEST_ALL, params=None):
        G = np.asfortranarray(G, dtype=np.float64)
        af = np.ascontiguousarray(af, dtype=np.float64)
        prm = params or Params.default()
        self._check(self.L.rvt_group_submit_gene(self.g, int(gene_id), G.shape[1], _dp(G), _dp(af), int(tests),
                                                 C.byref(prm)))

    def submit_gene_i8(self, gene_id, G8, tests=TEST_ALL, params=None):
        G8 = np.asfortranarray(G8, dtype=np.int8)
        prm = params or Params.default()
        self._check(self.L.rvt_group_submit_gene_i8(self.g, int(gene_id), G8.shape[1],
                                                    G8.ctypes.data_as(C.POINTER(C.c_int8)), int(tests), C.byref(prm),
                                                    None))

    def submit_gene_bed(self, gene_id, bed, M, tests=TEST_ALL, params=None):
        """PLINK 2-bit codes, SNP-major (rvt_group_submit_gene_bed)."""
        bed = np.ascontiguousarray(bed, dtype=np.uint8)
        prm = params or Params.default()
        self.L.rvt_group_submit_gene_bed.restype = C.c_int
        self.L.rvt_group_submit_gene_bed.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p,
                                                     c_double_p]
        self._check(self.L.rvt_group_submit_gene_bed(self.g, int(gene_id), int(M), bed.ctypes.data_as(C.c_void_p),
                                                     int(tests), C.byref(prm), None))

    def submit_gene_bgen(self, gene_id, blocks, layout, tests=TEST_ALL, params=None):
        """One gene as uncompressed BGEN probability blocks (rvt_group_submit_gene_bgen); file sample i = analysis row i
        unless a sample map was installed on the members."""
        prm = params or Params.default()
        M, keep, ptr, blen = Engine._bgen_args(blocks)
        self.L.rvt_group_submit_gene_bgen.restype = C.c_int
        self.L.rvt_group_submit_gene_bgen.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                                      C.c_uint32, C.c_void_p, c_double_p]
        self._check(self.L.rvt_group_submit_gene_bgen(self.g, int(gene_id), M, ptr, blen, int(layout), int(tests),
                                                      C.byref(prm), None))

    def score_block_host(self, G):
        """--meta score over the group: G is the host N x V matrix of consecutive sites (rvt_group_score_block_host)."""
        G = np.asfortranarray(G, dtype=np.float64)
        N, V = G.shape
        ok = np.zeros(V, dtype=np.int32)
        outs = [np.zeros(V) for _ in range(5)]
        self.L.rvt_group_score_block_host.restype = C.c_int
        self.L.rvt_group_score_block_host.argtypes = [C.c_void_p, C.c_int64, C.c_int, c_double_p, c_int_p] + [c_double_p] * 5
        self._check(self.L.rvt_group_score_block_host(self.g, N, V, _dp(G), ok.ctypes.data_as(c_int_p),
                                                      *[_dp(o) for o in outs]))
        return (ok,) + tuple(outs)

    def cov_band_host(self, G, d, halo, chunk=0):
        """--meta cov over the group: band[h, t] = value of head h and marker h + t (rvt_group_cov_band_host)."""
        G = np.asfortranarray(G, dtype=np.float64)
        N, V = G.shape
        band = np.zeros((V, halo + 1))
        xz = np.zeros((V, d))
        zz = np.zeros((d, d))
        poly = np.zeros(V, dtype=np.int32)
        self.L.rvt_group_cov_band_host.restype = C.c_int
        self.L.rvt_group_cov_band_host.argtypes = [C.c_void_p, C.c_int64, C.c_int, c_double_p, C.c_int, C.c_int, c_double_p,
                                                   c_double_p, c_double_p, c_int_p]
        self._check(self.L.rvt_group_cov_band_host(self.g, N, V, _dp(G), int(halo), int(chunk), _dp(band), _dp(xz), _dp(zz),
                                                   poly.ctypes.data_as(c_int_p)))
        return band, xz, zz, poly

    def collect(self, cap=4096):
        out = (GeneResult * cap)()
        n = C.c_int(0)
        self._check(self.L.rvt_group_collect(self.g, out, cap, C.byref(n)))
        return list(out[: n.value])

    def collect_ready(self, cap=4096):
        out = (GeneResult * cap)()
        n = C.c_int(0)
        self._check(self.L.rvt_group_collect_ready(self.g, out, cap, C.byref(n)))
        return list(out[: n.value])

    def set_kinship(self, U, S):
        U = np.asfortranarray(U, dtype=np.float32)
        S = np.ascontiguousarray(S, dtype=np.float32)
        fp = C.POINTER(C.c_float)
        self._check(self.L.rvt_group_set_kinship(self.g, U.shape[0], U.ctypes.data_as(fp), S.ctypes.data_as(fp)))

    def fit_fam_null(self, X, y):
        X = np.asfortranarray(X, dtype=np.float64)
        y = np.ascontiguousarray(y, dtype=np.float64)
        out = FamNull()
        self._check(self.L.rvt_group_fit_fam_null(self.g, X.shape[0], X.shape[1], _dp(X), _dp(y), C.byref(out)))
        return out

    def run_fam_tests_host(self, genes, tests=16, ids=None):
        """genes: list of host N x M arrays."""
        Gs = [np.asfortranarray(G, dtype=np.float64) for G in genes]
        n = len(Gs)
        ptrs = (c_double_p * n)(*[_dp(G) for G in Gs])
        Ms = np.ascontiguousarray([G.shape[1] for G in Gs], dtype=np.int32)
        arr_id = np.ascontiguousarray(ids if ids is not None else np.arange(n), dtype=np.int64)
        out = (GeneResult * n)()
        self._check(self.L.rvt_group_run_fam_tests_host(self.g, n, ptrs, Ms.ctypes.data_as(c_int_p),
                                                        arr_id.ctypes.data_as(C.POINTER(C.c_int64)), int(tests), out))
        return list(out)

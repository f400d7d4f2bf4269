"""ctypes binding of liblgcn.so (C ABI: include/lgcn.h).  Fails loudly when the library is absent."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblgcn.so")

MAX_REL = 16
REL_IDENT, REL_CSR, REL_RANGE, REL_RANGE16 = 0, 1, 2, 3
F_GN1, F_RELU1, F_GEMM2, F_GN2, F_RES, F_RELU2 = 1, 2, 4, 8, 16, 32
MMA_F32, MMA_BF16X3, MMA_BF16, MMA_F16X2 = 0, 1, 2, 3
MMA_NAMES = {"f32": MMA_F32, "bf16x3": MMA_BF16X3, "bf16": MMA_BF16, "f16x2": MMA_F16X2}


class LgcnError(RuntimeError):
    pass


class Rel(C.Structure):
    _fields_ = [("src", C.c_void_p), ("wp", C.c_void_p), ("mode", C.c_int32), ("ridx", C.c_int32)]


class PairsJob(C.Structure):      # lgcn_pairs_job_t
    _fields_ = [
        ("agt_ctrs", C.c_void_p), ("agt_off", C.c_void_p), ("ctx_ctrs", C.c_void_p), ("ctx_off", C.c_void_p),
        ("n_scenes", C.c_int32), ("legacy_offsets", C.c_int32), ("n_agt", C.c_int64), ("n_ctx", C.c_int64),
        ("dist_th", C.c_float), ("pad_", C.c_int32),
        ("hi", C.c_void_p), ("wi", C.c_void_p), ("cap", C.c_int64),
        ("n_pairs", C.c_void_p), ("rowptr", C.c_void_p), ("ws", C.c_void_p),
    ]


class Index(C.Structure):         # lgcn_index_t
    _fields_ = [
        ("idx_local", C.c_void_p), ("n_elem", C.c_int64),
        ("seg_off", C.c_void_p), ("seg_base", C.c_void_p), ("n_seg", C.c_int32), ("n_rel", C.c_int32),
        ("u_off", C.c_int64 * MAX_REL), ("v_off", C.c_int64 * MAX_REL), ("n_edges", C.c_int64 * MAX_REL),
        ("n_nodes", C.c_int64),
        ("rowptr", C.c_void_p), ("col", C.c_void_p), ("cnt", C.c_void_p), ("uv", C.c_void_p),
        ("jobs", C.c_void_p), ("n_jobs", C.c_int32), ("pad_", C.c_int32),
        ("clear_word", C.c_void_p),
    ]


class PredReg(C.Structure):       # lgcn_pred_reg_t
    _fields_ = [
        ("h", C.c_void_p * 8), ("w", C.c_void_p * 8), ("b", C.c_void_p * 8),
        ("ctrs", C.c_void_p), ("wd", C.c_void_p), ("bd", C.c_void_p), ("reg", C.c_void_p), ("hd", C.c_void_p),
        ("n_act", C.c_int64), ("n_mod", C.c_int32), ("np2", C.c_int32),
    ]


class AggMlp(C.Structure):
    _fields_ = [
        ("n_rows", C.c_int64), ("n_rel", C.c_int32), ("n_rel_csr", C.c_int32),
        ("flags", C.c_int32), ("eps", C.c_float), ("mma", C.c_int32), ("tile_rb", C.c_int32),
        ("rel", Rel * MAX_REL),
        ("rowptr", C.c_void_p), ("col", C.c_void_p),
        ("x4_a", C.c_void_p), ("x4_b", C.c_void_p), ("x4_c", C.c_void_p), ("w4", C.c_void_p),
        ("gn1_g", C.c_void_p), ("gn1_b", C.c_void_p),
        ("wp2", C.c_void_p), ("gn2_g", C.c_void_p), ("gn2_b", C.c_void_p),
        ("res", C.c_void_p), ("out", C.c_void_p), ("out_pre", C.c_void_p),
        ("out_mid", C.c_void_p), ("out_pre2", C.c_void_p),
        ("ch_wq", C.c_void_p), ("ch_gq_g", C.c_void_p), ("ch_gq_b", C.c_void_p), ("ch_wu", C.c_void_p),
        ("ch_u_out", C.c_void_p), ("ch_wv", C.c_void_p), ("ch_v_out", C.c_void_p),
    ]


LC_UNITS = 15


class LaneConv(C.Structure):      # lgcn_laneconv_t
    _fields_ = [
        ("n_rows", C.c_int64), ("x", C.c_void_p), ("wp", C.c_void_p * LC_UNITS),
        ("col", C.c_void_p), ("plan", C.c_void_p),
        ("rows_per_block", C.c_int32), ("cap", C.c_int32), ("n_units", C.c_int32), ("n_groups", C.c_int32),
        ("gstart", C.c_int32 * (LC_UNITS + 1)),
        ("gn1_g", C.c_void_p), ("gn1_b", C.c_void_p), ("wp2", C.c_void_p), ("gn2_g", C.c_void_p), ("gn2_b", C.c_void_p),
        ("eps", C.c_float), ("mma", C.c_int32), ("part", C.c_void_p), ("out", C.c_void_p), ("waves", C.c_int32),
    ]


class AttFused(C.Structure):      # lgcn_att_fused_t
    _fields_ = ([("agts", C.c_void_p), ("n_agt", C.c_int64), ("agt_ctrs", C.c_void_p), ("ctx_ctrs", C.c_void_p),
                 ("hi", C.c_void_p), ("wi", C.c_void_p), ("rowptr", C.c_void_p), ("cap", C.c_int64)]
                + [(n, C.c_void_p) for n in ("wpq", "gq", "bq", "wpc0q", "wd0", "bd0", "wpd2", "gd", "btd", "wpc0e", "V",
                                             "gc", "btc", "wpagt", "wpc1", "gn", "bn", "wplin", "gl", "bl")]
                + [("eps", C.c_float), ("mma", C.c_int32), ("targets_per_block", C.c_int32), ("out", C.c_void_p)])


_P, _I, _L, _F = C.c_void_p, C.c_int, C.c_int64, C.c_float

# name -> (restype, argtypes); every symbol include/lgcn.h declares
SIGNATURES = {
    "lgcn_version": (C.c_int, []),
    "lgcn_strerror": (C.c_char_p, [_I]),
    "lgcn_graph_gather": (C.c_int, [_P, _L, _P, _P, _I, _P, _P, _P]),
    "lgcn_csr_rowptr_elems": (C.c_int64, [_L, _I]),
    "lgcn_csr_ws_elems": (C.c_int64, [_L, _I]),
    "lgcn_csr_build": (C.c_int, [_P, _P, _P, _I, _L, _P, _P, _P, _P]),
    "lgcn_pairs_ws_elems": (C.c_int64, [_L, _I]),
    "lgcn_pairs_build": (C.c_int, [_P, _P, _P, _P, _I, _L, _L, _F, _I, _P, _P, _L, _P, _P, _P, _P]),
    "lgcn_pairs_build_multi": (C.c_int, [_P, _I, _P]),
    "lgcn_widen_i32": (C.c_int, [_P, _P, _L, _P, _P]),
    "lgcn_packed_bytes": (C.c_int64, [_I, _I]),
    "lgcn_pack_weight": (C.c_int, [_P, _I, _I, _I, _I, _P, _P]),
    "lgcn_pack_weight_t": (C.c_int, [_P, _I, _I, _P, _P]),
    "lgcn_pack_weight_batch": (C.c_int, [_P, _I, _I, _P]),
    "lgcn_agg_mlp": (C.c_int, [C.POINTER(AggMlp), _P]),
    "lgcn_lc_config": (C.c_int, [_I, _I, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "lgcn_lc_plan_elems": (C.c_int64, [_L, _I, _I]),
    "lgcn_lc_part_elems": (C.c_int64, [_L, _I, _I]),
    "lgcn_lc_plan_build": (C.c_int, [_P, _P, _L, _I, _I, _I, _I, C.POINTER(C.c_int32), _P, _P]),
    "lgcn_laneconv_fwd": (C.c_int, [C.POINTER(LaneConv), _P]),
    "lgcn_agg_mlp_pair": (C.c_int, [C.POINTER(AggMlp), C.POINTER(AggMlp), _P]),
    "lgcn_agg_mlp_multi": (C.c_int, [C.POINTER(C.POINTER(AggMlp)), _I, _P]),
    "lgcn_gn_bwd": (C.c_int, [_P, _P, _P, _P, _L, _F, _P, _P, _P, _P, _P, _P]),
    "lgcn_gn_fwd": (C.c_int, [_P, _P, _P, _P, _L, _F, _I, _P, _P]),
    "lgcn_gn_cl": (C.c_int, [_P, _L, _I, _I, _P, _P, _F, _P, _I, _I, _I, _P, _P]),
    "lgcn_gn_cl_bwd": (C.c_int, [_P, _P, _P, _P, _L, _I, _I, _F, _P, _P, _P, _P]),
    "lgcn_wgrad": (C.c_int, [C.POINTER(AggMlp), _P, _P, _P, _I, _P]),
    "lgcn_gather_rows": (C.c_int, [_P, _P, _P, _L, _P, _P]),
    "lgcn_gather_sum": (C.c_int, [_P, _P, _P, _L, _P, _P]),
    "lgcn_pair_add": (C.c_int, [_P, _P, _P, _P, _P, _P, _L, _P, _P]),
    "lgcn_att_fused": (C.c_int, [C.POINTER(AttFused), _P]),
    "lgcn_check_finite": (C.c_int, [_P, _L, _P, _L, _P, _I, _P]),
    "lgcn_mapnet_input": (C.c_int, [_P, _P, _L] + [_P] * 10 + [_F, _I, _P, _P]),
    "lgcn_att_pairs": (C.c_int, [_P, _P, _P, _P, _P, _L] + [_P] * 10 + [_F, _I, _P, _P]),
    "lgcn_conv_packed_bytes": (C.c_int64, [_I, _I, _I]),
    "lgcn_conv_pack_weight": (C.c_int, [_P, _I, _I, _I, _P, _P]),
    "lgcn_conv1d_gn": (C.c_int, [_P, _L, _I, _I, _P, _I, _I, _I, _P, _P, _F, _P, _I, _I, _P, _P]),
    "lgcn_res1d_gn": (C.c_int, [_P, _L, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _F, _P, _P]),
    "lgcn_res1d_pair_gn": (C.c_int, [_P, _L, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _F, _P, _P]),
    "lgcn_pred_reg": (C.c_int, [C.POINTER(PredReg), _P]),
    "lgcn_pred_final": (C.c_int, [_P, _P, _P, _P, _P, _P, _L, _I, _I, _P, _P, _P]),
    "lgcn_scan_ws_elems": (C.c_int64, [_L]),
    "lgcn_bool_square_bound": (C.c_int, [_P, _P, _L, _P, _P, _P]),
    "lgcn_bool_square": (C.c_int, [_P, _P, _L, _P, _P, _P, _P, _P]),
    "lgcn_bool_square_compact": (C.c_int, [_P, _P, _P, _L, _P, _P, _P]),
    "lgcn_cross_edges": (C.c_int, [_P, _P, _P, _L, _I, _P, _L, _P, _L, _P, _L, _F, _P, _P, _P]),
    "lgcn_index_uv_elems": (C.c_int64, [_L]),
    "lgcn_index_cnt_words": (C.c_int64, [_L, _I]),
    "lgcn_index_build": (C.c_int, [_P, _P]),
    "lgcn_att_pairs_ws": (C.c_int, [_P, _P, _P, _P, _P, _L] + [_P] * 10 + [_F, _I, _I, _P, _P]),
    "lgcn_att_pairs_wi": (C.c_int, [_P, _P, _P, _P, _P, _L] + [_P] * 10 + [_F, _I, _I, _P, _P]),
    "lgcn_pack_weight_kperm": (C.c_int, [_P, _I, _I, _P, _P]),
    "lgcn_pred_loss_fwd": (C.c_int, [_P, _P, _P, _P, _L, _I, _I, _F, _F, _F, _F, _F, _P, _P, _P, _P]),
    "lgcn_pred_loss_bwd": (C.c_int, [_P, _P, _P, _P, _L, _I, _I, _F, _F, _P, _P, _P, _P, _P, _P]),
}

_lib = None


def load():
    """Load liblgcn.so once; raise LgcnError with build instructions if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LgcnError(
            "liblgcn.so not found at %s -- the HIP extension is required (no CPU fallback). "
            "Build it with `python -c 'import __graft_entry__ as g; g.build()'` or "
            "`make -C lanegcn-1_amd/csrc`." % LIB_PATH
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().lgcn_strerror(int(rc)).decode()
        raise LgcnError("%s failed: %s (code %d)" % (what, msg, rc))

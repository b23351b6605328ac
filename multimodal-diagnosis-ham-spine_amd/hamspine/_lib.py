"""ctypes binding of libhamspine_hip.so (the C ABI declared in include/hamspine.h).

The product path has no CPU fallback: if the shared library is missing, or a tensor that reaches a
kernel wrapper is not on the HIP device, we raise.  (The CPU oracle lives under /oracle and is only
used by tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke().)
"""
import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libhamspine_hip.so")
HEADER_PATH = os.path.normpath(os.path.join(_HERE, "..", "..", "include", "hamspine.h"))

HS_OK = 0
HS_F32, HS_BF16 = 0, 1
A_KC, A_RC, A_CONV, A_DGRAD = 0, 1, 2, 3
B_KC, B_RC, B_WDGRAD, B_CONV = 0, 1, 2, 3
ACT_NONE, ACT_RELU, ACT_GELU = 0, 1, 2
MUL_NONE, MUL_GELU_GRAD, MUL_RELU_MASK = 0, 1, 2
CAST_MAX, ADAM_MAX = 48, 32

i32, i64, u64, f32, vp = C.c_int32, C.c_int64, C.c_uint64, C.c_float, C.c_void_p


class HamspineError(RuntimeError):
    pass


class ConvGeom(C.Structure):
    _fields_ = [(n, i32) for n in (
        "N", "H", "W", "C", "P", "Q", "K", "R", "S", "stride", "pad",
        "row_pitch", "img_pitch", "qstep", "no_bounds")]


class GemmParams(C.Structure):
    _fields_ = [
        ("dtype", i32), ("a_kind", i32), ("b_kind", i32),
        ("M", i32), ("N", i32), ("K", i32),
        ("A", vp), ("B", vp),
        ("a_elems", i64), ("b_elems", i64),
        ("lda", i32), ("ldb", i32),
        ("g", ConvGeom),
        ("batch", i32), ("batch_inner", i32),
        ("a_bs0", i64), ("a_bs1", i64), ("b_bs0", i64), ("b_bs1", i64),
        ("d_bs0", i64), ("d_bs1", i64),
        ("split_k", i32), ("splitk_ws", vp),
        ("D", vp), ("ldd", i32), ("out_dtype", i32),
        ("alpha", f32), ("bias", vp), ("act", i32),
        ("D_preact", vp), ("residual", vp), ("ldr", i32),
        ("dropout_p", f32), ("dropout_seed", u64),
        ("mul_mode", i32), ("mul_src", vp), ("ldm", i32),
        ("accumulate", i32),
        ("seg_rows", i32), ("D_seg", vp * 2), ("colscale", vp), ("residual_before_act", i32), ("colstats", vp),
        ("rowsum_a", vp), ("rowsum_seg", vp * 2),
        ("bnb_x", vp), ("bnb_scale", vp), ("bnb_shift", vp), ("bnb_mean", vp), ("bnb_invstd", vp), ("bnb_partials", vp),
        ("bn_finish", vp),
        ("bnb_y", vp),
        ("bnb_finish", vp),
    ]


class BnParams(C.Structure):
    _fields_ = [
        ("dtype", i32), ("C", i32), ("M", i64), ("training", i32), ("relu", i32),
        ("eps", f32), ("momentum", f32),
        ("x", vp), ("residual", vp), ("y", vp), ("gamma", vp), ("beta", vp),
        ("running_mean", vp), ("running_var", vp), ("save_mean", vp), ("save_invstd", vp),
        ("scale", vp), ("shift", vp), ("ws", vp), ("ws_bytes", i64), ("partial_rows", i32),
        ("stats_done", i32),
    ]


class BnBwdParams(C.Structure):
    _fields_ = [
        ("dtype", i32), ("C", i32), ("M", i64), ("training", i32), ("relu", i32),
        ("dy", vp), ("y", vp), ("x", vp), ("gamma", vp), ("save_mean", vp), ("save_invstd", vp),
        ("dx", vp), ("dres", vp), ("dgamma", vp), ("dbeta", vp), ("ws", vp), ("ws_bytes", i64),
        ("scale", vp), ("shift", vp), ("partial_rows", i32), ("sums_done", i32),
    ]


class AttnDesc(C.Structure):
    _fields_ = [
        ("dtype", i32), ("B", i32), ("H", i32), ("Lq", i32), ("Lk", i32), ("hd", i32),
        ("q_bs", i64), ("k_bs", i64), ("v_bs", i64), ("o_bs", i64),
        ("q_ld", i32), ("k_ld", i32), ("v_ld", i32), ("o_ld", i32),
        ("scale", f32), ("dropout_p", f32), ("seed", u64), ("key_mask", vp),
    ]


class ConvBn(C.Structure):
    _fields_ = [
        ("Cin", i32), ("Cout", i32), ("R", i32), ("stride", i32), ("pad", i32),
        ("w", vp), ("gamma", vp), ("beta", vp), ("running_mean", vp), ("running_var", vp),
        ("dw", vp), ("dgamma", vp), ("dbeta", vp),
    ]


class ResblockDesc(C.Structure):
    _fields_ = [
        ("dtype", i32), ("N", i32), ("H", i32), ("W", i32), ("training", i32),
        ("eps", f32), ("momentum", f32), ("n_main", i32), ("main", ConvBn * 3),
        ("has_ds", i32), ("ds", ConvBn), ("inference", i32),
    ]


class StemDesc(C.Structure):
    _fields_ = [
        ("dtype", i32), ("N", i32), ("H", i32), ("W", i32), ("training", i32),
        ("eps", f32), ("momentum", f32), ("cb", ConvBn), ("inference", i32),
    ]


class Linear(C.Structure):
    _fields_ = [("in_f", i32), ("out_f", i32), ("w", vp), ("b", vp), ("dw", vp), ("db", vp)]


class Norm(C.Structure):
    _fields_ = [("gamma", vp), ("beta", vp), ("dgamma", vp), ("dbeta", vp)]


class BertLayerDesc(C.Structure):
    _fields_ = [
        ("dtype", i32), ("B", i32), ("L", i32), ("hidden", i32), ("heads", i32), ("inter", i32),
        ("ln_eps", f32), ("hidden_dropout", f32), ("attn_dropout", f32), ("seed", u64),
        ("attention_mask", vp),
        ("q", Linear), ("k", Linear), ("v", Linear), ("ao", Linear), ("ln1", Norm),
        ("inter_l", Linear), ("out_l", Linear), ("ln2", Norm),
    ]


RESNET_MAX_BLOCKS, RESNET_MAX_TAPS, BERT_MAX_LAYERS = 24, 4, 24


class ResnetDesc(C.Structure):
    _fields_ = [("stem", StemDesc), ("n_blocks", i32), ("blocks", ResblockDesc * RESNET_MAX_BLOCKS), ("n_taps", i32),
                ("tap_block", i32 * RESNET_MAX_TAPS)]


class ResnetPlan(C.Structure):
    _fields_ = [("saved_bytes", i64), ("ws_bytes", i64), ("tap_offset", i64 * RESNET_MAX_TAPS),
                ("tap_C", i32 * RESNET_MAX_TAPS), ("tap_H", i32 * RESNET_MAX_TAPS), ("tap_W", i32 * RESNET_MAX_TAPS)]


class BertDesc(C.Structure):
    _fields_ = [
        ("dtype", i32), ("B", i32), ("L", i32), ("hidden", i32), ("vocab", i32), ("max_pos", i32), ("n_types", i32),
        ("pad_id", i32), ("ln_eps", f32), ("embed_dropout", f32), ("seed", u64),
        ("word", vp), ("pos", vp), ("type0", vp), ("gamma", vp), ("beta", vp),
        ("dword", vp), ("dpos", vp), ("dtype0", vp), ("dgamma", vp), ("dbeta", vp),
        ("n_layers", i32), ("layers", BertLayerDesc * BERT_MAX_LAYERS),
    ]


_lib = None


def lib():
    """Load the shared library once; fail loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HamspineError(
                f"{LIB_PATH} not found: build it with `python __graft_entry__.py build` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        _lib = C.CDLL(LIB_PATH)
        _declare(_lib)
    return _lib


def _declare(l):
    P = C.POINTER
    l.hs_last_error.restype = C.c_char_p
    l.hs_gemm.argtypes = [P(GemmParams), vp]
    l.hs_gemm_splitk_ws_bytes.argtypes = [P(GemmParams)]
    l.hs_gemm_splitk_ws_bytes.restype = i64
    l.hs_gemm_stat_rows.argtypes = [P(GemmParams)]
    l.hs_gemm_bn_finish_rows.argtypes = [P(GemmParams)]
    l.hs_gemm_bn_finish_rows.restype = i32
    l.hs_gemm_bnb_finish_rows.argtypes = [P(GemmParams)]
    l.hs_gemm_bnb_finish_rows.restype = i32
    l.hs_gemm_tile_rows.argtypes = [P(GemmParams)]
    l.hs_gemm_tile_rows.restype = i32
    l.hs_gemm_suggest_split.argtypes = [i32] * 4
    l.hs_batchnorm_fwd.argtypes = [P(BnParams), vp]
    l.hs_batchnorm_bwd.argtypes = [P(BnBwdParams), vp]
    l.hs_batchnorm_ws_bytes.argtypes = [i64, i32, i32]
    l.hs_batchnorm_ws_bytes.restype = i64
    l.hs_layernorm_fwd.argtypes = [i32, vp, vp, vp, vp, vp, vp, i64, i32, f32, vp]
    l.hs_layernorm_bwd.argtypes = [i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i64, i32, vp]
    l.hs_layernorm_bwd_pre.argtypes = [i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, f32, u64, vp, i64, i64, i32, vp]
    l.hs_layernorm_bwd_ws_bytes.argtypes = [i64, i32]
    l.hs_layernorm_bwd_ws_bytes.restype = i64
    l.hs_maxpool_fwd.argtypes = [i32, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]
    l.hs_maxpool_bwd.argtypes = [i32, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]
    l.hs_mean_tokens_fwd.argtypes = [i32, vp, vp, i32, i32, i32, i32, vp]
    l.hs_mean_tokens_bwd.argtypes = [i32, vp, vp, i32, i32, i32, i32, vp]
    l.hs_pack_image.argtypes = [i32, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]
    l.hs_pack_stem_weight.argtypes = [i32, vp, vp, i32, i32, i32, i32, vp]
    l.hs_unpack_stem_wgrad.argtypes = [vp, vp, i32, i32, i32, i32, vp]
    l.hs_cast_f32_to_bf16_multi.argtypes = [i32, P(vp), P(vp), P(i64), vp]
    l.hs_axpby.argtypes = [i32, i32, vp, vp, vp, i64, f32, f32, vp]
    l.hs_transpose_bf16.argtypes = [vp, vp, i32, i32, i64, i64, vp]
    l.hs_transpose_bf16_multi.argtypes = [i32, P(vp), P(vp), P(i32), P(i32), P(i64), P(i64), vp]
    l.hs_dropout.argtypes = [i32, vp, vp, i64, f32, u64, vp]
    l.hs_relu_fwd.argtypes = [i32, vp, vp, i64, vp]
    l.hs_relu_bwd.argtypes = [i32, vp, vp, vp, i64, vp]
    l.hs_gelu_bwd.argtypes = [i32, vp, vp, vp, i64, vp]
    l.hs_mul_dev_scalar.argtypes = [vp, vp, vp, i64, vp]
    l.hs_colsum.argtypes = [i32, vp, i64, i32, i32, vp, vp, i64, i32, vp]
    l.hs_colsum_ws_bytes.argtypes = [i64, i32]
    l.hs_colsum_ws_bytes.restype = i64
    l.hs_softmax_fwd.argtypes = [i32, vp, vp, vp, vp, i64, i32, i32, i32, i32, f32, u64, vp]
    l.hs_softmax_bwd.argtypes = [i32, vp, vp, vp, i64, i32, i32, i32, f32, u64, vp]
    l.hs_bert_embed_fwd.argtypes = [i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, f32, f32, u64, vp]
    l.hs_bert_embed_bwd.argtypes = [i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]
    l.hs_cross_entropy.argtypes = [vp, vp, vp, f32, i32, i32, vp, vp, vp, vp]
    l.hs_adam_step_multi.argtypes = [i32, P(vp), P(vp), P(vp), P(vp), P(i64), f32, f32, f32, f32, f32, i32, i32, f32, vp]
    l.hs_adam_step_multi_shadow.argtypes = [i32, P(vp), P(vp), P(vp), P(vp), P(vp), P(i64), f32, f32, f32, f32, f32, i32, i32, f32, vp]
    l.hs_weight_shadow_set.argtypes = [vp, vp]
    l.hs_wgrad_group_begin.argtypes = [vp]
    l.hs_wgrad_group_end.argtypes = [vp]
    l.hs_weight_shadow_clear.argtypes = []
    l.hs_weight_shadow_clear.restype = None
    l.hs_sgd_step_multi.argtypes = [i32, P(vp), P(vp), P(vp), P(i64), f32, f32, f32, i32, i32, f32, vp]
    l.hs_attention_query.argtypes = [P(AttnDesc), P(i64), P(i64)]
    l.hs_attention_fwd.argtypes = [P(AttnDesc), vp, vp, vp, vp, vp, i64, vp, i64, vp]
    l.hs_attention_bwd.argtypes = [P(AttnDesc), vp, vp, vp, vp, vp, vp, vp, vp, i64, vp, i64, vp]
    l.hs_resblock_query.argtypes = [P(ResblockDesc), P(i64), P(i64)]
    l.hs_resblock_fwd.argtypes = [P(ResblockDesc), vp, vp, vp, i64, vp, i64, vp]
    l.hs_resblock_bwd.argtypes = [P(ResblockDesc), vp, vp, vp, vp, vp, i64, vp, i64, vp]
    l.hs_stem_query.argtypes = [P(StemDesc), P(i64), P(i64)]
    l.hs_stem_fwd.argtypes = [P(StemDesc), vp, vp, vp, i64, vp, i64, vp]
    l.hs_stem_bwd.argtypes = [P(StemDesc), vp, vp, vp, i64, vp, i64, vp]
    l.hs_bert_layer_query.argtypes = [P(BertLayerDesc), P(i64), P(i64)]
    l.hs_bert_layer_fwd.argtypes = [P(BertLayerDesc), vp, vp, vp, i64, vp, i64, vp]
    l.hs_bert_layer_bwd.argtypes = [P(BertLayerDesc), vp, vp, vp, vp, i64, vp, i64, vp]
    l.hs_abi_sizeof.argtypes = [i32]
    l.hs_abi_sizeof.restype = i64
    l.hs_resnet_query.argtypes = [P(ResnetDesc), P(ResnetPlan)]
    l.hs_resnet_fwd.argtypes = [P(ResnetDesc), vp, vp, i64, vp, i64, vp]
    l.hs_resnet_debug_offsets.argtypes = [P(ResnetDesc), P(i64), i32]
    l.hs_resnet_bwd.argtypes = [P(ResnetDesc), P(vp), vp, i64, vp, i64, vp]
    l.hs_bert_query.argtypes = [P(BertDesc), P(i64), P(i64), P(i64)]
    l.hs_bert_fwd.argtypes = [P(BertDesc), vp, vp, vp, i64, vp, i64, vp]
    l.hs_bert_bwd.argtypes = [P(BertDesc), vp, vp, vp, vp, i64, vp, i64, vp]
    l.hs_linear_fwd.argtypes = [i32, vp, i64, i32, P(Linear), vp, vp, i32, i32, i32, vp, vp, i32, f32, u64, vp]
    l.hs_linear_bwd.argtypes = [i32, vp, i64, i32, P(Linear), vp, vp, i32, vp, i32, i32, i32, vp, i32, vp, vp, i64, vp]
    l.hs_linear_bwd_ws_bytes.argtypes = [i64, i32, i32, i32]
    l.hs_linear_bwd_ws_bytes.restype = i64
    l.hs_prof_dump.argtypes = [C.c_char_p]
    l.hs_set_overlap.argtypes = [i32]
    l.hs_kl_rows.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, f32, vp]
    l.hs_lstm_cell_fwd.argtypes = [vp, i32, vp, i32, vp, vp, vp, vp, i32, i32, vp]
    l.hs_lstm_cell_bwd.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, i32, vp]
    l.hs_gru_cell_fwd.argtypes = [vp, i32, vp, i32, vp, vp, vp, i32, i32, vp]
    l.hs_gru_cell_bwd.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, vp]
    l.hs_concat2_t.argtypes = [i32, vp, i32, vp, i32, vp, i64, vp]
    l.hs_split2_t.argtypes = [i32, vp, vp, i32, vp, i32, i64, vp]
    l.hs_prof_calibrate.argtypes = [vp, i32, C.POINTER(C.c_float)]
    l.hs_tta_expand.argtypes = [vp, vp, i64, i32, i32, P(i32), i32, vp]
    l.hs_group_mean.argtypes = [vp, vp, i32, i64, vp]
    l.hs_repeat.argtypes = [i32, vp, vp, i64, i32, vp]
    l.hs_gradcam.argtypes = [i32, vp, vp, vp, i32, i32, i32, vp]
    l.hs_stage_images_u8.argtypes = [vp, vp, i32, i32, i32, P(f32), P(f32), vp]
    l.hs_set_overlap.restype = None
    l.hs_grad_milestones.argtypes = [i32, P(vp), P(vp)]
    l.hs_pointwise_stat_rows.argtypes = [i64, i32, i32]
    l.hs_pointwise_stat_rows.restype = i32
    l.hs_pointwise_fwd.argtypes = [vp, i64, i32, i32, vp, i32, vp, i32, vp, vp]
    l.hs_measure_build.argtypes = []
    l.hs_measure_build.restype = i32
    l.hs_set_wgrad_nt.argtypes = [i32]
    l.hs_set_wgrad_nt.restype = None
    l.hs_dwconv_ws_bytes.argtypes = [i32] * 5
    l.hs_dwconv_ws_bytes.restype = i64
    l.hs_dwconv_fwd.argtypes = [i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, i64, vp]
    l.hs_dwconv_bwd.argtypes = [i32, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, i64, vp]
    l.hs_layerscale_fwd.argtypes = [i32, vp, vp, vp, i32, vp, vp, i64, i32, vp]
    l.hs_layerscale_bwd.argtypes = [i32, vp, vp, vp, vp, i32, vp, vp, vp, i64, i64, i32, vp]
    l.hs_layerscale_ws_bytes.argtypes = [i64, i32]
    l.hs_layerscale_ws_bytes.restype = i64
    l.hs_patchify_fwd.argtypes = [i32, vp, vp, i32, i32, i32, i32, i32, i32, vp]
    l.hs_patchify_bwd.argtypes = [i32, vp, vp, i32, i32, i32, i32, i32, i32, vp]


ABI_STRUCTS = None


def abi_structs():
    """(number for hs_abi_sizeof, ctypes mirror) of every struct of the C ABI"""
    return [(0, ConvGeom), (1, GemmParams), (2, BnParams), (3, BnBwdParams), (4, AttnDesc), (5, ConvBn), (6, ResblockDesc),
            (7, StemDesc), (8, Linear), (9, Norm), (10, BertLayerDesc), (11, ResnetDesc), (12, ResnetPlan), (13, BertDesc)]


def check(status, what=""):
    if status != HS_OK:
        msg = lib().hs_last_error().decode("utf-8", "replace")
        raise HamspineError(f"{what} failed (status {status}): {msg}")


def exported_symbols():
    """Names declared in include/hamspine.h (used by the CPU-side ABI test)."""
    txt = open(HEADER_PATH).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(hs_[a-z0-9_]+)\s*\(", txt)))

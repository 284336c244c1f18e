"""
ctypes binding of libtocvp.so (include/tocvp.h) + thin tensor-level wrappers.

PyTorch is used only for device memory and the current HIP stream; every computation on the hot
path is one of the hand-written HIP kernels behind the C-ABI.  There is NO fallback: if the
library is missing or a call fails, this module raises.
"""

import ctypes
import math
import os
import struct
import weakref

import torch

from . import build as _build
from .precision import knob as _knob

_c_float_p = ctypes.c_void_p
_LIB = None

ACT_NONE, ACT_RELU, ACT_GELU = 0, 1, 2
ACT_GATE = 3          # split GEMM epilogue: `residual` is a gate (output zeroed where it is <= 0), see tocvp.h

_SIGNATURES = {
    "tocvp_version": (ctypes.c_int, []),
    "tocvp_strerror": (ctypes.c_char_p, [ctypes.c_int]),
    "tocvp_gemm_f32": (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
        ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
        ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "tocvp_split_weights_bf16": (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
        ctypes.c_void_p]),
    "tocvp_gemm_bf16split_f32": (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
        ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
        ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
        ctypes.c_void_p]),
    "tocvp_split_weights_frag_bf16": (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
        ctypes.c_void_p]),
    "tocvp_gemm_bf16wfrag_f32": (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
        ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
        ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
        ctypes.c_int, ctypes.c_void_p]),
    "tocvp_split_weights_frag_f16": (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "tocvp_gemm_f16wfrag_f32": (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
        ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
        ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "tocvp_tail_taps_f16x3_bytes": (ctypes.c_size_t, []),
    "tocvp_pack_tail_taps_f16x3": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "tocvp_conv5x5_dec_f16x3_tail_f32": (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
        ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "tocvp_dec_tail_sum_f32": (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
        ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "tocvp_absmax_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_void_p]),
    "tocvp_dec_tail_sum_placed_f32": (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
        ctypes.c_long, ctypes.c_long, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
        ctypes.c_void_p]),
    "tocvp_dec_tail_placed_f32": (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
        ctypes.c_void_p, ctypes.c_long, ctypes.c_long, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int,
        ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "tocvp_mlp_f16x3_fused_ws_bytes": (ctypes.c_size_t, []),
    "tocvp_mlp_f16x3_fused_f32": (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
        ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
        ctypes.c_size_t, ctypes.c_void_p]),
    "tocvp_gemm_f16chunk_f32": (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
        ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "tocvp_conv3x3_up2_f16x3_f32": (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
        ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "tocvp_gemm_f16mid_f32": (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
        ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "tocvp_mha_qk16_rows_f32": (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
        ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float,
        ctypes.c_void_p, ctypes.c_void_p]),
    "tocvp_mha_one_query_f32": (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
        ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float,
        ctypes.c_void_p, ctypes.c_void_p]),
    "tocvp_mha_qk16_rows_split_f16": (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
        ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float,
        ctypes.c_void_p, ctypes.c_void_p]),
    "tocvp_mha_one_query_split_f16": (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
        ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float,
        ctypes.c_void_p, ctypes.c_void_p]),
    "tocvp_mha_planes_f16": (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
        ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float,
        ctypes.c_void_p, ctypes.c_void_p]),
    "tocvp_copy4d_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_long, ctypes.c_long, ctypes.c_long, ctypes.c_void_p,
                                        ctypes.c_long, ctypes.c_long, ctypes.c_long, ctypes.c_int, ctypes.c_int,
                                        ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "tocvp_clamp01_rows_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_long,
                                              ctypes.c_long, ctypes.c_void_p]),
    "tocvp_gemm_wfrag_ws_bytes": (ctypes.c_size_t, []),
    "tocvp_gemm_f16wfrag_ws_f32": (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
        ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
        ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "tocvp_xattn_collapsed_f32": (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p,
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
        ctypes.c_float, ctypes.c_void_p]),
    "tocvp_layernorm_split_bf16": (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
        ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_void_p]),
    "tocvp_mha_split_bf16": (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
        ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
        ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p]),
    "tocvp_layernorm_f32": (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
        ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_void_p]),
    "tocvp_mha_f32": (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
        ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
        ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p]),
    "tocvp_mha_qk16_f32": (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
        ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
        ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p]),
    "tocvp_slot_attn_ws_init": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "tocvp_slot_attn_ws_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int]),
    "tocvp_slot_attn_iter_f32": (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
        ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float,
        ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "tocvp_slot_attn_iter_planes_f32": (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
        ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t,
        ctypes.c_void_p]),
    "tocvp_gru_gates_f32": (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
        ctypes.c_int, ctypes.c_void_p]),
    "tocvp_pos_embed_f32": (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
        ctypes.c_void_p]),
    "tocvp_conv5x5_in3_f32": (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
        ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "tocvp_pack_conv_weights_f32": (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
        ctypes.c_void_p]),
    "tocvp_conv5x5_f32": (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
        ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
        ctypes.c_int, ctypes.c_void_p]),
    "tocvp_dec_tapsum_f32": (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "tocvp_dec_tail_f32": (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
        ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
        ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "tocvp_text_embed_f32": (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
        ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float,
        ctypes.c_void_p]),
    "tocvp_split_conv_weights_bf16": (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "tocvp_split_conv_weights_frag_bf16": (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "tocvp_conv5x5_bf16x3_f32": (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
        ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "tocvp_conv_weights_f16f8_bytes": (ctypes.c_size_t, [ctypes.c_int]),
    "tocvp_split_conv_weights_f16f8": (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "tocvp_conv5x5_f16f8_f32": (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
        ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "tocvp_conv_weights_wino_f16x3_bytes": (ctypes.c_size_t, []),
    "tocvp_split_conv_weights_wino_f16x3": (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "tocvp_conv5x5_dec_wino_f16x3_f32": (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "tocvp_conv_weights_dec_f16x3_bytes": (ctypes.c_size_t, []),
    "tocvp_split_conv_weights_dec_f16x3": (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "tocvp_conv5x5_dec_f16x3_f32": (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
        ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
        ctypes.c_void_p]),
    "tocvp_bmm_f32": (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_long, ctypes.c_int,
        ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_long, ctypes.c_int,
        ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_long, ctypes.c_int, ctypes.c_int,
        ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_void_p]),
    "tocvp_gemm_tn_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                         ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                         ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "tocvp_gemm_tn_bf16x3_multi_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                                       ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                                       ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "tocvp_gemm_tn_bf16x3_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                                ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                                ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "tocvp_attn_bwd_f32": (ctypes.c_int, [ctypes.c_void_p] * 10 + [ctypes.c_int] * 5 + [ctypes.c_float, ctypes.c_void_p]),
    "tocvp_softmax_rows_f32": (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_void_p,
        ctypes.c_int, ctypes.c_void_p]),
    "tocvp_softmax_bwd_f32": (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_float,
        ctypes.c_void_p]),
    "tocvp_act_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_int,
                                     ctypes.c_void_p]),
    "tocvp_act_bwd_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long,
                                         ctypes.c_int, ctypes.c_void_p]),
    "tocvp_dropout_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long,
                                         ctypes.c_float, ctypes.c_void_p]),
    "tocvp_axpby_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_float,
                                       ctypes.c_float, ctypes.c_void_p]),
    "tocvp_colsum_partial_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                                ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "tocvp_layernorm_bwd_f32": (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
        ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_void_p]),
    "tocvp_embedding_bwd_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                               ctypes.c_int, ctypes.c_void_p]),
    "tocvp_mse_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                     ctypes.c_void_p, ctypes.c_long, ctypes.c_float, ctypes.c_void_p]),
    "tocvp_sqnorm_partial_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_long,
                                                ctypes.c_void_p]),
    "tocvp_dec_tail_grad_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                               ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                               ctypes.c_void_p]),
    "tocvp_conv3x3_t4_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                            ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                            ctypes.c_void_p]),
    "tocvp_dec_class_reduce_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                                  ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                  ctypes.c_int, ctypes.c_void_p]),
    "tocvp_adam_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                      ctypes.c_long, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "tocvp_clip_scale_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p]),
    "tocvp_metrics_ws_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int]),
    "tocvp_psnr_ssim_f32": (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
        ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t,
        ctypes.c_void_p]),
    "tocvp_conv3x3_f32": (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
        ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
        ctypes.c_int, ctypes.c_void_p]),
    "tocvp_conv3x3_f16x3_f32": (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
        ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
        ctypes.c_int, ctypes.c_void_p]),
    "tocvp_conv5x5_f16x3_f32": (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
        ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "tocvp_slot_composite_f32": (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
        ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "tocvp_bilinear_resize_f32": (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
        ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "tocvp_rmsnorm_f32": (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_float,
        ctypes.c_void_p]),
    "tocvp_embedding_f32": (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
        ctypes.c_void_p]),
    "tocvp_mha_bias_f32": (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
        ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
        ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "tocvp_slot_init_f32": (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
        ctypes.c_int, ctypes.c_void_p]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)


class TocvpError(RuntimeError):
    pass


class TocvpRangeError(TocvpError):
    """
    An operand left the range of an fp16-plane arithmetic (|activation| < 255, |weight| < 63).
    ``owner`` = (module, attribute name) of the arithmetic knob that governs the failing call (set by
    the modules through ``range_owner`` / ``gemm_precision(owner=...)``), or None.
    """

    def __init__(self, msg, owner=None):
        super().__init__(msg)
        self.owner = owner


class LaunchTimer:
    """
    Optional per-launch device timing of selected kernels with HIP events recorded on the stream
    the kernel is enqueued on (torch's current stream).  bench.py uses it to measure the average
    launch duration of the dominant kernel inside the timed region; it is off by default.
    """

    def __init__(self, only=("conv5x5",)):
        self.records = {}          # name -> list of (start_event, stop_event, work_units)
        self.only = tuple(only)    # name prefixes that are timed (events perturb short kernels: keep the
                                   # timed region of bench.py to the dominant kernel, time the rest apart)

    def wrap(self, name, units, fn):
        if not name.startswith(self.only):
            return fn()
        start = torch.cuda.Event(enable_timing=True)
        stop = torch.cuda.Event(enable_timing=True)
        start.record()
        out = fn()
        stop.record()
        self.records.setdefault(name, []).append((start, stop, units))
        return out

    def summary(self):
        """ name -> dict(launches, total_ms, units) ; call after torch.cuda.synchronize() """
        out = {}
        for name, recs in self.records.items():
            ms = sum(a.elapsed_time(b) for a, b, _ in recs)
            out[name] = {"launches": len(recs), "total_ms": ms, "units": sum(u for _, _, u in recs)}
        return out


TIMER = None   # set to a LaunchTimer() to time launches (conv5x5_*, gemm_*, mha_*, slot_attn_*)


def _timed(name, units, fn):
    """ ``name`` may be a callable (evaluated only when a LaunchTimer is installed: the f-string of every launch's name costs
    host time on a host-bound step) """
    if TIMER is None:
        return fn()
    return TIMER.wrap(name() if callable(name) else name, units, fn)


def lib():
    """ Load libtocvp.so once; raise loudly when it is missing (no CPU / torch fallback). """
    global _LIB
    if _LIB is None:
        path = _build.LIB_PATH
        if not os.path.exists(path):
            raise TocvpError(
                f"{path} not found: build the HIP extension first "
                f"(python -m textocvp_amd.build, or __graft_entry__.build())")
        handle = ctypes.CDLL(path)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)   # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _LIB = handle
    return _LIB


def _check(code, what):
    if code != 0:
        msg = lib().tocvp_strerror(code).decode()
        raise TocvpError(f"{what} failed: {msg} ({code})")


# Host side of a launch: the evaluation step at 8 sequences is ~2400 launches and HOST-bound (53 ms of Python against
# 62 ms until the device is done, scripts/host_profile.py), so the per-launch helpers are the cheap forms: the raw current
# stream from torch's C binding (torch.cuda.current_stream() builds a Stream object: 7 us of the ~20 us per launch), plain
# integers for pointers (ctypes converts them under the c_void_p argtypes).
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    if _raw_stream is not None and _raw_device is not None:
        return _raw_stream(_raw_device())
    return torch.cuda.current_stream().cuda_stream


def _ptr(t):
    return t.data_ptr() if t is not None else None


def _dev_f32(t, name):
    if t.device.type != "cuda":
        raise TocvpError(f"{name} must live on the GPU (got {t.device}); there is no CPU path")
    if t.dtype != torch.float32:
        raise TocvpError(f"{name} must be float32 (got {t.dtype})")
    return t


# --------------------------------------------------------------------------------------------
# GEMM arithmetic mode: "fp32" (exact fp32 MFMA), "bf16x3" or "bf16x6" (split-bf16 operands)
# --------------------------------------------------------------------------------------------
_GEMM_PRECISION = "fp32"
_NSPLIT = {"bf16x3": 2, "bf16x6": 3, "f16x3": 22}     # 22 = two fp16 planes
_SPLIT_CACHE = {}


_RANGE_OWNER = None          # (module, attr) whose arithmetic knob governs the kernels launched right now


class range_owner:
    """ context manager: range errors raised inside name ``(module, attr)`` as the knob to change """

    def __init__(self, module, attr):
        self.owner = (module, attr)

    def __enter__(self):
        global _RANGE_OWNER
        self.prev, _RANGE_OWNER = _RANGE_OWNER, self.owner
        return self

    def __exit__(self, *a):
        global _RANGE_OWNER
        _RANGE_OWNER = self.prev


class check_range:
    """ context manager: every fp16-plane kernel launched inside verifies its operands (slow: syncs) """

    def __init__(self, on=True):
        self.on = bool(on)

    def __enter__(self):
        global _CHECK_RANGE
        self.prev, _CHECK_RANGE = _CHECK_RANGE, self.on
        return self

    def __exit__(self, *a):
        global _CHECK_RANGE
        _CHECK_RANGE = self.prev


class gemm_precision:
    """
    context manager selecting the arithmetic of ``linear`` calls issued inside it; ``owner`` =
    (module, attr) of the knob the mode came from (named by range errors, see TocvpRangeError)
    """

    def __init__(self, mode, owner=None):
        if mode not in ("fp32", "bf16x3", "bf16x6", "f16x3"):
            raise ValueError(f"unknown GEMM precision {mode!r}")
        self.mode, self.owner = mode, owner

    def __enter__(self):
        global _GEMM_PRECISION, _RANGE_OWNER
        self.prev, _GEMM_PRECISION = _GEMM_PRECISION, self.mode
        self.prev_owner = _RANGE_OWNER
        if self.owner is not None:
            _RANGE_OWNER = self.owner
        return self

    def __exit__(self, *a):
        global _GEMM_PRECISION, _RANGE_OWNER
        _GEMM_PRECISION, _RANGE_OWNER = self.prev, self.prev_owner


_WFRAG = os.environ.get("TOCVP_GEMM_WFRAG", "1") != "0"   # W in MFMA-fragment order (bypasses LDS)
# chunk-resident persistent f16x3 GEMM (gemm_f16c.hip) for plane inputs with N % 512 == 0, K % 128 == 0, from
# _GEMM_CHUNK_MIN_TILES tiles of 128 x 512 up (one workgroup per CU: fewer tiles leave CUs idle and the 64 x 64 / 128 x 128
# kernels win; 9600 x 512 x 2048 = 75 tiles: 118 vs 103 us, 9600 x 1536 x 512 = 225 tiles: 55 vs 63 us).  TOCVP_GEMM_CHUNK=0: off
_GEMM_CHUNK = os.environ.get("TOCVP_GEMM_CHUNK", "1") != "0"
_GEMM_CHUNK_MIN_TILES = int(os.environ.get("TOCVP_GEMM_CHUNK_MIN_TILES", "192"))
# mid-size form of the chunk kernel (64 x 256 tiles, two workgroups per CU) for plane-input products (TOCVP_GEMM_MID=0: off)
_GEMM_MID = os.environ.get("TOCVP_GEMM_MID", "1") != "0"
# from _GEMM_MID_MIN_TILES tiles of 64 x 256 (fewer leave CUs idle: 2400 x 512 x 2048 = 76 tiles runs 34.5 us against 31.3 us
# on the two-operand planes kernel, 2400 x 1536 x 512 = 228 tiles 16.8 against 21.1 us); no upper row limit since the
# XCD-aware tile order (38400 x 1536 x 512 188 vs 207 us against the 256 x 128-tile planes kernel)
_GEMM_MID_MIN_TILES = int(os.environ.get("TOCVP_GEMM_MID_MIN_TILES", "192"))
# f16x3 pre-scales activations by 2^8 and weights by 2^10 into the fp16 range (gemm_bf16.hip, Elem<true>):
# fp32-class inside these bounds, saturating outside.  TOCVP_CHECK_RANGE=1 verifies every call (slow: syncs).
F16X3_ACT_RANGE, F16X3_WEIGHT_RANGE = 255.0, 63.0
_CHECK_RANGE = os.environ.get("TOCVP_CHECK_RANGE", "0") != "0"


def absmax(t):
    """ max |t| of a device fp32 tensor on the library's own reduction (tocvp_absmax_f32); NaN reads as inf """
    _dev_f32(t, "absmax operand")
    tc = t if t.is_contiguous() else t.contiguous()
    word = torch.empty(1, device=t.device, dtype=torch.int32)
    _check(lib().tocvp_absmax_f32(_ptr(tc), tc.numel(), _ptr(word), _stream()), "tocvp_absmax_f32")
    return struct.unpack("f", struct.pack("i", int(word.item())))[0]


def _copy_plan(src, dst):
    """ (sizes of <= 3 leading dims, src strides, dst strides, L) for tocvp_copy4d_f32, or None if the views do not fit """
    shape = list(src.shape)
    ss, ds = list(src.stride()), list(dst.stride())
    # the innermost run that is contiguous on BOTH sides
    L, d = 1, len(shape)
    while d > 0 and ss[d - 1] == L and ds[d - 1] == L:
        L *= shape[d - 1]
        d -= 1
    dims = [(shape[i], ss[i], ds[i]) for i in range(d) if shape[i] != 1]
    merged = []
    for n, a, b in dims:                                     # merge neighbours that are contiguous with each other
        if merged and merged[-1][1] == n * a and merged[-1][2] == n * b:
            m = merged.pop()
            merged.append((m[0] * n, a, b))
        else:
            merged.append((n, a, b))
    if len(merged) > 3 or L % 4 or any(a % 4 or b % 4 for _, a, b in merged):
        return None
    while len(merged) < 3:
        merged.insert(0, (1, 0, 0))
    return merged, L


# TOCVP_LIB_COPIES=0: leave the index copies to torch (dst.copy_ / .contiguous() / torch.stack) -- A/B switch
_LIB_COPIES = os.environ.get("TOCVP_LIB_COPIES", "1") != "0"
_COPY_PLANS = {}


def copy_strided(src, dst):
    """ dst[...] = src[...] (same shape, fp32, same device) through tocvp_copy4d_f32; returns dst.  Views whose layout the
    kernel does not take (more than three strided dimensions, runs that are not multiples of 4 floats, bases that are not
    16-byte aligned) are copied by torch (index copies are data movement, not arithmetic). """
    if not _LIB_COPIES:
        dst.copy_(src)
        return dst
    key = (src.shape, src.stride(), dst.stride())
    plan = _COPY_PLANS.get(key)
    if plan is None:
        _dev_f32(src, "copy source"), _dev_f32(dst, "copy destination")
        assert src.shape == dst.shape, (src.shape, dst.shape)
        plan = _copy_plan(src, dst) if src.numel() else ([(0, 0, 0)] * 3, 0)
        if plan is None or plan[1] >= 2 ** 31 or any(n >= 2 ** 31 for n, _, _ in plan[0]):
            # a layout the kernel does not take (more than three strided dimensions, runs or strides that are not
            # multiples of 4 floats -- odd image sizes, permuted dataloader layouts -- or extents beyond 32 bits):
            # pure data movement, left to torch's own device copy
            plan = "torch"
        if len(_COPY_PLANS) < 4096:
            _COPY_PLANS[key] = plan
    if src.dtype != torch.float32 or dst.dtype != torch.float32 or not src.is_cuda or not dst.is_cuda:
        raise TocvpError("copy_strided: fp32 tensors on the GPU only")
    if plan == "torch" or (src.data_ptr() | dst.data_ptr()) & 15:       # 16-byte loads / stores need aligned bases
        dst.copy_(src)
        return dst
    (n0, a0, b0), (n1, a1, b1), (n2, a2, b2) = plan[0]
    if n0 * n1 * n2 == 0:
        return dst
    code = lib().tocvp_copy4d_f32(src.data_ptr(), a0, a1, a2, dst.data_ptr(), b0, b1, b2, n0, n1, n2, plan[1], _stream())
    if code:
        _check(code, "tocvp_copy4d_f32")
    return dst


def contiguous(t):
    """ t if it is contiguous, else a contiguous copy made by the library's strided copy (no torch kernel) """
    if t.is_contiguous():
        return t
    if not _LIB_COPIES:
        return t.contiguous()
    return copy_strided(t, torch.empty(t.shape, device=t.device, dtype=t.dtype))


def stack1(tensors):
    """ torch.stack(tensors, dim=1) of equally shaped contiguous (B, ...) fp32 tensors through the strided copy """
    if not _LIB_COPIES:
        return torch.stack(tensors, dim=1)
    B = tensors[0].shape[0]
    out = torch.empty((B, len(tensors)) + tuple(tensors[0].shape[1:]), device=tensors[0].device, dtype=torch.float32)
    for i, t_ in enumerate(tensors):
        copy_strided(t_, out[:, i])
    return out


def clamp01_rows(src):
    """ src (R, ...) fp32 whose rows are contiguous (any row stride) -> contiguous clamp(src, 0, 1) """
    _dev_f32(src, "clamp01 operand")
    R = src.shape[0]
    assert R == 0 or src[0].is_contiguous()
    out = torch.empty(src.shape, device=src.device, dtype=torch.float32)
    row_len = out[0].numel() if R else 0
    _check(lib().tocvp_clamp01_rows_f32(_ptr(src), src.stride(0) if R > 1 else row_len, _ptr(out), R, row_len, _stream()),
           "tocvp_clamp01_rows_f32")
    return out


def _check_f16_range(amax, what, owner=None):
    """ checked pass (TOCVP_CHECK_RANGE=1 / check_range()): the fp16-plane arithmetic saturates at |x| = 255.9 """
    if not amax < F16X3_ACT_RANGE:                          # also trips on NaN
        raise TocvpRangeError(
            f"{what} out of the fp16-plane range: |x| max {amax:.4g} (< {F16X3_ACT_RANGE}); "
            f"select the bf16 / fp32 arithmetic for this model (setup_model.calibrate_precision)",
            owner if owner is not None else _RANGE_OWNER)


def _check_f16_weight_range(w, what):
    """ |w| < 63 for an fp16-plane weight image; skipped while a HIP graph is being captured (no sync there) """
    if torch.cuda.is_current_stream_capturing():
        return
    wmax = absmax(w)
    if not wmax < F16X3_WEIGHT_RANGE:
        raise TocvpRangeError(
            f"{what} weight out of the fp16-plane range: |w| max {wmax:.4g} (< {F16X3_WEIGHT_RANGE}); "
            f"select the bf16 / fp32 arithmetic for this model (setup_model.calibrate_precision)",
            _RANGE_OWNER)


class SplitAct:
    """
    An activation already split into operand planes by its producer: ``planes`` is (rows, P, D)
    bf16 (P = 2 / 3) or fp16 (P = 2, values pre-scaled by 2^8), ``shape`` the logical fp32 shape
    (..., D).  Consumed as the A operand of a split GEMM.
    """

    __slots__ = ("planes", "shape")

    def __init__(self, planes, shape):
        self.planes, self.shape = planes, tuple(shape)

    @property
    def nsplit(self):
        """ arithmetic code of the planes: 2 / 3 bf16 planes, 22 = two fp16 planes of 2^8 x (f16x3) """
        return 22 if self.planes.dtype == torch.float16 else self.planes.shape[1]


def _alloc_planes(rows, nsplit, D, device):
    """ (rows, planes, D) operand planes for arithmetic code ``nsplit`` """
    if nsplit == 22:
        return torch.empty((rows, 2, D), device=device, dtype=torch.float16)
    return torch.empty((rows, nsplit, D), device=device, dtype=torch.bfloat16)


def active_nsplit():
    """ planes per operand of the GEMM arithmetic selected by the enclosing gemm_precision() """
    return _NSPLIT.get(_GEMM_PRECISION, 0) if _WFRAG else 0


def _split_weight(w, nsplit, frag=False):
    """ (N, K) fp32 -> cached operand planes: bf16 (N, nsplit, K) or fragment order; nsplit 22: fp16 planes of 2^10 w in
    fragment order (rebuilt when the weight changes) """
    key = (id(w), nsplit, frag)
    hit = _SPLIT_CACHE.get(key)
    # the weakref guards against id()/address reuse after the original weight was freed
    if hit is not None and hit[0]() is w and hit[1] == (w._version, w.data_ptr()):
        return hit[2]
    if len(_SPLIT_CACHE) > 4096:
        for k_ in [k_ for k_, v in _SPLIT_CACHE.items() if v[0]() is None]:
            del _SPLIT_CACHE[k_]
    N, K = w.shape
    if nsplit == 22:
        out = torch.empty((N, 2, K), device=w.device, dtype=torch.float16)
        _check(lib().tocvp_split_weights_frag_f16(_ptr(w), _ptr(out), N, K, _stream()),
               "tocvp_split_weights_frag_f16")
        _SPLIT_CACHE[key] = (weakref.ref(w), (w._version, w.data_ptr()), out)
        return out
    out = torch.empty((N, nsplit, K), device=w.device, dtype=torch.bfloat16)
    if frag:
        _check(lib().tocvp_split_weights_frag_bf16(_ptr(w), _ptr(out), N, K, nsplit, _stream()),
               "tocvp_split_weights_frag_bf16")
    else:
        _check(lib().tocvp_split_weights_bf16(_ptr(w), _ptr(out), N, K, nsplit, _stream()),
               "tocvp_split_weights_bf16")
    _SPLIT_CACHE[key] = (weakref.ref(w), (w._version, w.data_ptr()), out)
    return out


# split-K of the skinny f16x3 GEMMs (gemm_bf16.hip, SK = true): TOCVP_GEMM_KSPLIT=0 turns it off.  The workspace
# (arrival counters + accumulator records, 16.8 MB) is per device AND per stream -- two streams may run skinny GEMMs
# at the same time (decoder overlap) -- and must start zeroed; the kernels leave the counters at zero.
_GEMM_KSPLIT = os.environ.get("TOCVP_GEMM_KSPLIT", "1") != "0"
_GEMM_KSPLIT_MAX_ROWS = 4096          # above this the tile count alone fills the CUs (the library would not split)
_KSPLIT_WS = {}


def _ksplit_workspace(device, stream=None):
    """ (tensor, pointer, bytes) of the zero-initialised split-K workspace of this device and stream """
    key = (device.index, torch.cuda.current_stream(device).cuda_stream if stream is None else stream)
    rec = _KSPLIT_WS.get(key)
    if rec is None:
        nbytes = lib().tocvp_gemm_wfrag_ws_bytes()
        wk = torch.zeros(nbytes // 4, device=device, dtype=torch.float32)
        rec = _KSPLIT_WS[key] = (wk, ctypes.c_void_p(wk.data_ptr()), nbytes)
    return rec


# fused predictor MLP (csrc/mlp_fused.hip): relu(x W1^T + b1) W2^T + b2 + residual in ONE launch, the hidden activation
# never reaches HBM.  Bit-identical to linear(linear(x, W1, b1, RELU, out_split=22), W2, b2, residual) on whole tiles;
# the tiles of a partly filled last round of workgroups are cut along the hidden dimension through a per-stream workspace
# (deterministic; those rows differ from the uncut sum in the last bits).  TOCVP_MLP_FUSED=0 keeps the two GEMMs.
# Isolated, 38400 x 512 x 2048 x 512 (scripts/probes/mlp_fused_check.hip): 516 vs 636 us; below ~11000 rows a tile's
# 8 MB weight stream (one workgroup per CU) no longer hides behind its products and the two GEMMs are faster.
_MLP_FUSED = os.environ.get("TOCVP_MLP_FUSED", "1") != "0"
_MLP_FUSED_MIN_ROWS = int(os.environ.get("TOCVP_MLP_FUSED_MIN_ROWS", "11000"))
_MLP_WS = {}


def _mlp_workspace(device):
    key = (device.index, _stream())
    rec = _MLP_WS.get(key)
    if rec is None:
        nbytes = lib().tocvp_mlp_f16x3_fused_ws_bytes()
        wk = torch.zeros(nbytes // 4, device=device, dtype=torch.float32)
        rec = _MLP_WS[key] = (wk, ctypes.c_void_p(wk.data_ptr()), nbytes)
    return rec


def mlp_fused_ok(x, w1, w2):
    """ shapes / mode the fused MLP kernel takes: fp16-plane input of 512 features, hidden width a multiple of 128 """
    return (_MLP_FUSED and not _CHECK_RANGE and isinstance(x, SplitAct) and x.nsplit == 22 and x.shape[-1] == 512
            and tuple(w1.shape[1:]) == (512,) and w1.shape[0] % 128 == 0 and tuple(w2.shape) == (512, w1.shape[0])
            and x.planes.shape[0] >= _MLP_FUSED_MIN_ROWS and x.planes.shape[0] * 2048 < 2 ** 32 and _WFRAG)


def mlp_fused(x, w1, b1, w2, b2, residual=None):
    """
    x: SplitAct of fp16 planes (rows, 2, 512); w1 (Hd, 512), b1 (Hd), w2 (512, Hd), b2 (512) in nn.Linear layout;
    residual (..., 512) fp32 or None -> relu(x w1^T + b1) w2^T + b2 + residual, shape x.shape.
    """
    assert mlp_fused_ok(x, w1, w2)
    _dev_f32(w1, "w1"), _dev_f32(w2, "w2")
    M, Hd = x.planes.shape[0], w1.shape[0]
    w1c = w1 if w1.is_contiguous() else w1.contiguous()
    w2c = w2 if w2.is_contiguous() else w2.contiguous()
    f1, f2 = _split_weight(w1c, 22, frag=True), _split_weight(w2c, 22, frag=True)
    r2 = None
    if residual is not None:
        r2 = residual.reshape(-1, 512)
        assert r2.shape[0] == M and r2.is_contiguous()
    out = torch.empty((M, 512), device=w1.device, dtype=torch.float32)
    _, wk_ptr, wk_bytes = _mlp_workspace(w1.device)
    _timed(lambda: f"mlp_fused_{M}x512x{Hd}", 4.0 * M * 512 * Hd, lambda: _check(
        lib().tocvp_mlp_f16x3_fused_f32(_ptr(x.planes), _ptr(f1), _ptr(b1), _ptr(f2), _ptr(b2), _ptr(r2), 512, _ptr(out),
                                        512, M, 512, Hd, wk_ptr, wk_bytes, _stream()),
        "tocvp_mlp_f16x3_fused_f32"))
    return out.reshape(x.shape)


# --------------------------------------------------------------------------------------------
# tensor-level wrappers
# --------------------------------------------------------------------------------------------

def linear(x, weight, bias=None, act=ACT_NONE, residual=None, rowvec=None, rv_div=1, rv_flip=False,
           out=None, precision=None, out_split=0, chunk_ok=True):
    """
    y = act(x W^T + bias + rowvec[idx(row)]) + residual over the last axis of ``x``.
    x: (..., K) contiguous fp32 tensor or a SplitAct; weight: (N, K) in nn.Linear layout.
    out_split = 2 / 3 / 22: return a SplitAct (operand planes) for a following split GEMM.
    chunk_ok = False keeps a plane-input product off the persistent chunk-resident kernel (one workgroup per CU with 128 KB
    of LDS: it cannot share a CU with another stream's workgroups -- the predictor under the overlapped decoder).
    """
    _dev_f32(weight, "weight")
    N, K = weight.shape
    w = weight if weight.is_contiguous() else weight.contiguous()
    pre_split = isinstance(x, SplitAct)
    if pre_split:
        assert x.shape[-1] == K, (x.shape, weight.shape)
        lead, x2, M = x.shape[:-1], x.planes, x.planes.shape[0]
        nsplit = x.nsplit
    else:
        _dev_f32(x, "x")
        assert x.shape[-1] == K, (weight.shape, x.shape)
        lead = x.shape[:-1]
        x2 = x.reshape(-1, K)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        M = x2.shape[0]
        nsplit = _NSPLIT.get(_GEMM_PRECISION if precision is None else precision, 0)
    r2 = None
    if residual is not None:
        r2 = residual.reshape(-1, N)
        assert r2.shape[0] == M and r2.is_contiguous()
    rv_mod = 1
    if rowvec is not None:
        assert rowvec.is_contiguous() and rowvec.shape[-1] == N
        rv_mod = rowvec.numel() // N
    frag_ok = nsplit and K % 64 == 0 and N % 32 == 0 and _WFRAG
    if nsplit == 22 and not frag_ok:
        nsplit = 0                            # f16x3 exists in fragment-order form only: fp32 MFMA
    if (pre_split or out_split) and not frag_ok:
        raise TocvpError("split activations need the fragment-order split GEMM (K, N % 32 == 0)")
    if out_split:
        assert out is None and out_split == nsplit
        out = _alloc_planes(M, nsplit, N, w.device)
    elif out is None:
        out = torch.empty((M, N), device=w.device, dtype=torch.float32)
    if frag_ok and nsplit == 22 and _CHECK_RANGE:
        if not pre_split:
            _check_f16_range(absmax(x2), f"f16x3 GEMM ({M}x{N}x{K}) activation")
        _check_f16_weight_range(w, f"f16x3 GEMM ({N}x{K})")
    bn = 512 if N % 512 == 0 else 384
    use_chunk = (frag_ok and pre_split and nsplit == 22 and _GEMM_CHUNK and chunk_ok and rowvec is None and N % bn == 0 and
                 K % 128 == 0 and ((M + 127) // 128) * (N // bn) >= _GEMM_CHUNK_MIN_TILES and
                 act in (ACT_NONE, ACT_RELU, ACT_GELU))
    if not pre_split and nsplit and M * 4 * K >= 2 ** 32:
        # the split kernels address A with 32-bit byte offsets as well: more than 2^32 bytes of fp32 activations (1 048 576
        # rows at K = 1024 -- configs[3] decoded in ONE call, as the reference's evaluator does, hands the MLPPatchDecoder
        # 2.85 M rows) came back wrong behind the limit in rounds 1-3, silently (scripts/probes/gemm_big_m_fp32.py; the
        # exact-fp32 kernel is not affected): row blocks
        if rowvec is not None:
            raise TocvpError("linear: a row-periodic vector with more than 2^32 bytes of activations is not supported")
        step = ((2 ** 32 - 1) // (4 * K)) // 128 * 128
        for r0 in range(0, M, step):
            mb = min(step, M - r0)
            sub = linear(x2[r0:r0 + mb], weight, bias, act=act, residual=None if r2 is None else r2[r0:r0 + mb],
                         precision=precision, out_split=out_split, chunk_ok=chunk_ok, out=None if out_split else out[r0:r0 + mb])
            if out_split:
                out[r0:r0 + mb].copy_(sub.planes)
        return SplitAct(out, (*lead, N)) if out_split else out.reshape(*lead, N)
    if pre_split and not use_chunk and rowvec is None and M * (4 if nsplit == 22 else 2 * nsplit) * K >= 2 ** 32:
        # the plane-input kernels address the A operand with 32-bit byte offsets (LDS-DMA sources): more than 2^32 bytes
        # of planes (1 048 576 rows at K = 1024) go through in row blocks -- found with scripts/probes/gemm_chunk_big_m.py:
        # rows behind the limit came back wrong, silently
        step = ((2 ** 32 - 1) // ((4 if nsplit == 22 else 2 * nsplit) * K)) // 128 * 128
        for r0 in range(0, M, step):
            mb = min(step, M - r0)
            sub = linear(SplitAct(x2[r0:r0 + mb], (mb, K)), weight, bias, act=act,
                         residual=None if r2 is None else r2[r0:r0 + mb], precision=precision, out_split=out_split,
                         chunk_ok=chunk_ok, out=None if out_split else out[r0:r0 + mb])
            if out_split:
                out[r0:r0 + mb].copy_(sub.planes)
        return SplitAct(out, (*lead, N)) if out_split else out.reshape(*lead, N)
    if use_chunk:
        # A chunks resident in LDS, weights streamed in fragment order (gemm_f16c.hip); 32-bit DMA offsets: row blocks
        ws = _split_weight(w, 22, frag=True)
        step = ((2 ** 32 - 1) // (4 * K)) // 128 * 128
        for r0 in range(0, M, step):
            mb = min(step, M - r0)
            _timed(lambda: f"gemm_split{nsplit}_{mb}x{N}x{K}", 2.0 * mb * N * K, lambda: _check(
                lib().tocvp_gemm_f16chunk_f32(x2[r0:].data_ptr(), _ptr(ws), _ptr(bias),
                                              None if r2 is None else r2[r0:].data_ptr(), N, out[r0:].data_ptr(),
                                              int(bool(out_split)), N, mb, N, K, int(act), _stream()),
                "tocvp_gemm_f16chunk_f32"))
    elif (frag_ok and pre_split and nsplit == 22 and _GEMM_MID and rowvec is None and N % 256 == 0 and K % 128 == 0 and
            ((M + 63) // 64) * (N // 256) >= _GEMM_MID_MIN_TILES and M * 4 * K < 2 ** 32 and
            act in (ACT_NONE, ACT_RELU, ACT_GELU)):
        # 64 x 256 tiles, A chunks by LDS-DMA, weights streamed (gemm_f16c.hip, mid-size form)
        ws = _split_weight(w, 22, frag=True)
        _timed(lambda: f"gemm_split{nsplit}_{M}x{N}x{K}", 2.0 * M * N * K, lambda: _check(
            lib().tocvp_gemm_f16mid_f32(_ptr(x2), _ptr(ws), _ptr(bias), _ptr(r2), N, _ptr(out), int(bool(out_split)), N, M, N,
                                        K, int(act), _stream()),
            "tocvp_gemm_f16mid_f32"))
    elif frag_ok and nsplit == 22 and not pre_split and _GEMM_KSPLIT and M <= _GEMM_KSPLIT_MAX_ROWS:
        # few output tiles (small batches): split-K over idle CUs through a per-stream workspace
        ws = _split_weight(w, nsplit, frag=True)
        st = _stream()
        _, wk_ptr, wk_bytes = _ksplit_workspace(w.device, st)
        _timed(lambda: f"gemm_split{nsplit}_{M}x{N}x{K}", 2.0 * M * N * K, lambda: _check(
            lib().tocvp_gemm_f16wfrag_ws_f32(_ptr(x2), K, _ptr(ws), _ptr(bias), _ptr(r2), N, _ptr(rowvec),
                                             int(rv_div), int(rv_mod), int(bool(rv_flip)), _ptr(out),
                                             int(bool(out_split)), N, M, N, K, int(act), wk_ptr,
                                             wk_bytes, st),
            "tocvp_gemm_f16wfrag_ws_f32"))
    elif frag_ok:
        ws = _split_weight(w, nsplit, frag=True)
        _timed(lambda: f"gemm_split{nsplit}_{M}x{N}x{K}", 2.0 * M * N * K, lambda: _check(
            lib().tocvp_gemm_bf16wfrag_f32(_ptr(x2), int(pre_split), K, _ptr(ws), nsplit,
                                           _ptr(bias), _ptr(r2), N, _ptr(rowvec), int(rv_div),
                                           int(rv_mod), int(bool(rv_flip)), _ptr(out),
                                           int(bool(out_split)), N, M, N, K, int(act), _stream()),
            "tocvp_gemm_bf16wfrag_f32"))
    elif nsplit and K % 32 == 0:
        ws = _split_weight(w, nsplit)
        _check(lib().tocvp_gemm_bf16split_f32(_ptr(x2), K, _ptr(ws), nsplit, _ptr(bias), _ptr(r2), N,
                                              _ptr(rowvec), int(rv_div), int(rv_mod),
                                              int(bool(rv_flip)), _ptr(out), N, M, N, K, int(act),
                                              _stream()), "tocvp_gemm_bf16split_f32")
    else:
        _check(lib().tocvp_gemm_f32(_ptr(x2), K, _ptr(w), _ptr(bias), _ptr(r2), N, _ptr(rowvec),
                                    int(rv_div), int(rv_mod), int(bool(rv_flip)), _ptr(out), N, M, N,
                                    K, int(act), _stream()), "tocvp_gemm_f32")
    if out_split:
        return SplitAct(out, (*lead, N))
    return out.reshape(*lead, N)


def layer_norm(x, gamma, beta, eps, add=None, split=0):
    """
    LayerNorm over the last axis; ``add`` (R, D) is added row-periodically before the norm.
    split = 2 / 3 / 22: return a SplitAct (operand planes) instead of an fp32 tensor.
    """
    _dev_f32(x, "x")
    D = x.shape[-1]
    x2 = x.reshape(-1, D)
    if not x2.is_contiguous():
        x2 = x2.contiguous()
    rows = x2.shape[0]
    add_rows = 0
    if add is not None:
        assert add.is_contiguous() and add.shape[-1] == D
        add_rows = add.numel() // D
    if split:
        y = _alloc_planes(rows, split, D, x.device)
        _check(lib().tocvp_layernorm_split_bf16(_ptr(x2), _ptr(add), add_rows, _ptr(gamma),
                                                _ptr(beta), _ptr(y), int(split), rows, D, float(eps),
                                                _stream()), "tocvp_layernorm_split_bf16")
        return SplitAct(y, x.shape)
    y = torch.empty_like(x2)
    _check(lib().tocvp_layernorm_f32(_ptr(x2), _ptr(add), add_rows, _ptr(gamma), _ptr(beta),
                                     _ptr(y), rows, D, float(eps), _stream()),
           "tocvp_layernorm_f32")
    return y.reshape(x.shape)


def bmm_f32(A, B, C, M, N, Kd, lda, ldb, ldc, transA=False, transB=False, batch=(1, 1), sA=(0, 0), sB=(0, 0),
            sC=(0, 0), alpha=1.0, accumulate=False):
    """
    C[b1, b2] (M x N) = alpha * op(A[b1, b2]) (M x Kd) * op(B[b1, b2]) (Kd x N) on the exact fp32 MFMA (tocvp_bmm_f32):
    A, B, C are device fp32 tensors whose data_ptr() is element (0, 0) of slice (0, 0); ld* leading dimensions and
    s* = (stride of b1, stride of b2) in elements address the slices in place.
    """
    for t_ in (A, B, C):
        _dev_f32(t_, "bmm operand")
    _check(lib().tocvp_bmm_f32(_ptr(A), int(lda), int(sA[0]), int(sA[1]), int(bool(transA)), _ptr(B), int(ldb),
                               int(sB[0]), int(sB[1]), int(bool(transB)), _ptr(C), int(ldc), int(sC[0]), int(sC[1]),
                               int(batch[0]), int(batch[1]), int(M), int(N), int(Kd), float(alpha),
                               int(bool(accumulate)), _stream()), "tocvp_bmm_f32")
    return C


def xattn_operands(G, HT):
    """
    G (B * 128, 512), HT (B * 512, 128) fp32 -> their fp16 operand planes (2^10 w = hi + lo) in MFMA-fragment
    order, as tocvp_xattn_collapsed_f32 reads them.  Checked mode verifies |w| < 63 like every fp16-plane weight.
    """
    _dev_f32(G, "G"), _dev_f32(HT, "HT")
    outs = []
    for w in (G, HT):
        assert w.is_contiguous()
        if _CHECK_RANGE:
            _check_f16_weight_range(w, "collapsed cross-attention operand")
        N, Kd = w.shape
        out = torch.empty((N, 2, Kd), device=w.device, dtype=torch.float16)
        _check(lib().tocvp_split_weights_frag_f16(_ptr(w), _ptr(out), N, Kd, _stream()), "tocvp_split_weights_frag_f16")
        outs.append(out)
    return tuple(outs)


def xattn_collapsed(x, gamma, beta, eps, Gf, Hf, bias, heads, Lt, scale):
    """
    z = x + CrossAttention(LayerNorm(x), caption) with the projections folded into the caption operands
    (csrc/xattn.hip): x (B, Tq, 512) fp32 contiguous -> (B, Tq, 512).
    """
    _dev_f32(x, "x")
    B, Tq, E = x.shape
    LP = 16 if Lt <= 16 else (32 if Lt <= 32 else 64)   # caption slots per head the operands were padded to
    assert x.is_contiguous() and Gf.shape[0] == B * heads * LP and Hf.shape[0] == B * E and Lt <= 64
    if _CHECK_RANGE:                               # |LayerNorm(x)| <= sqrt(E) max|gamma| + max|beta|
        _check_f16_range(absmax(gamma) * E ** 0.5 + absmax(beta),
                         "collapsed cross-attention: bound of the LayerNorm output")
    y = torch.empty_like(x)
    _timed(lambda: f"xattn_{B}x{Tq}x{Lt}", 4.0 * B * Tq * E * heads * LP, lambda: _check(
        lib().tocvp_xattn_collapsed_f32(_ptr(x), _ptr(gamma), _ptr(beta), float(eps), _ptr(Gf), _ptr(Hf), _ptr(bias),
                                        _ptr(y), B, Tq, E, int(heads), int(Lt), float(scale), _stream()),
        "tocvp_xattn_collapsed_f32"))
    return y


# matrix products of the attention kernel (Q K^T and P V): "f16x3" (split fp16 operands on the f16 matrix
# cores, fp32-class, |q|, |k|, |v| < 255) or "fp32" (exact fp32 MFMA); the softmax is fp32 either way
_ATTN_QK16 = _knob("TOCVP_ATTN_QK", "f16x3") != "fp32"


# sequence lengths of 128 n + 1 (ViT: patches + class token): last query row in its own launch (TOCVP_MHA_TAIL_ROW=0: off)
_MHA_TAIL_ROW = os.environ.get("TOCVP_MHA_TAIL_ROW", "1") != "0"


def mha(q, k, v, heads, scale, key_len=None, out_split=0, bias=None):
    """
    q: (B, Tq, E) view with unit last stride (may be a column slice of a fused projection);
    k, v: (B, Tk, E) likewise.  Returns (B, Tq, E) contiguous fp32, or a SplitAct if out_split.
    """
    B, Tq, E = q.shape
    Tk = k.shape[1]
    dh = E // heads
    for name, t in (("q", q), ("k", k), ("v", v)):
        _dev_f32(t, name)
        assert t.stride(2) == 1 and t.stride(0) == t.shape[1] * t.stride(1), (name, t.stride())
    if key_len is not None:
        assert key_len.dtype == torch.int32 and key_len.is_cuda and key_len.numel() == B
    if bias is not None:
        assert not out_split and bias.is_contiguous() and tuple(bias.shape) == (heads, Tq, Tk)
        o = torch.empty((B, Tq, E), device=q.device, dtype=torch.float32)
        _check(lib().tocvp_mha_bias_f32(_ptr(q), q.stride(1), _ptr(k), k.stride(1), _ptr(v),
                                        v.stride(1), _ptr(o), E, B, heads, Tq, Tk, dh, float(scale),
                                        _ptr(key_len), _ptr(bias), _stream()), "tocvp_mha_bias_f32")
        return o
    if out_split == 22 and _ATTN_QK16 and not _CHECK_RANGE:
        # both products on the f16 matrix cores, O as fp16 operand planes (the split the consuming GEMM would make)
        o = _alloc_planes(B * Tq, 22, E, q.device)
        tail = _MHA_TAIL_ROW and Tq > 128 and Tq % 128 == 1 and Tk <= 1024 and B * heads >= 256

        def planes_out():
            _check(lib().tocvp_mha_qk16_rows_split_f16(_ptr(q), q.stride(1), _ptr(k), k.stride(1), _ptr(v), v.stride(1),
                                                       _ptr(o), B, heads, Tq, Tq - 1 if tail else Tq, Tk, dh, float(scale),
                                                       _ptr(key_len), _stream()), "tocvp_mha_qk16_rows_split_f16")
            if tail:
                _check(lib().tocvp_mha_one_query_split_f16(_ptr(q), q.stride(1), _ptr(k), k.stride(1), _ptr(v), v.stride(1),
                                                           _ptr(o), B, heads, Tq, Tq - 1, Tk, dh, float(scale),
                                                           _ptr(key_len), _stream()), "tocvp_mha_one_query_split_f16")
        _timed(lambda: f"mha_{B}x{heads}x{Tq}x{Tk}x{dh}", 4.0 * B * heads * Tq * Tk * dh, planes_out)
        return SplitAct(o, (B, Tq, E))
    if out_split:
        o = _alloc_planes(B * Tq, out_split, E, q.device)
        _check(lib().tocvp_mha_split_bf16(_ptr(q), q.stride(1), _ptr(k), k.stride(1), _ptr(v),
                                          v.stride(1), _ptr(o), int(out_split), B, heads, Tq, Tk, dh,
                                          float(scale), _ptr(key_len), _stream()),
               "tocvp_mha_split_bf16")
        return SplitAct(o, (B, Tq, E))
    o = torch.empty((B, Tq, E), device=q.device, dtype=torch.float32)
    if _ATTN_QK16 and _CHECK_RANGE:
        _check_f16_range(max(absmax(q), absmax(k), absmax(v)),
                         "attention q / k / v (f16x3 products)", owner=("kernels", "_ATTN_QK16"))
    if _ATTN_QK16 and _MHA_TAIL_ROW and Tq > 128 and Tq % 128 == 1 and Tk <= 1024 and B * heads >= 256:
        # one row past a multiple of the 128-query tile (256 patches + class token): the tile kernel on the first Tq - 1
        # rows, the last row on its own (a third tile would stage every key and value again for that one row)
        def two():
            _check(lib().tocvp_mha_qk16_rows_f32(_ptr(q), q.stride(1), _ptr(k), k.stride(1), _ptr(v), v.stride(1), _ptr(o), E,
                                                 B, heads, Tq, Tq - 1, Tk, dh, float(scale), _ptr(key_len), _stream()),
                   "tocvp_mha_qk16_rows_f32")
            _check(lib().tocvp_mha_one_query_f32(_ptr(q), q.stride(1), _ptr(k), k.stride(1), _ptr(v), v.stride(1), _ptr(o), E,
                                                 B, heads, Tq, Tq - 1, Tk, dh, float(scale), _ptr(key_len), _stream()),
                   "tocvp_mha_one_query_f32")
        _timed(lambda: f"mha_{B}x{heads}x{Tq}x{Tk}x{dh}", 4.0 * B * heads * Tq * Tk * dh, two)
        return o
    fn = lib().tocvp_mha_qk16_f32 if _ATTN_QK16 else lib().tocvp_mha_f32
    _timed(lambda: f"mha_{B}x{heads}x{Tq}x{Tk}x{dh}", 4.0 * B * heads * Tq * Tk * dh, lambda: _check(
        fn(_ptr(q), q.stride(1), _ptr(k), k.stride(1), _ptr(v), v.stride(1), _ptr(o), E, B, heads, Tq, Tk,
           dh, float(scale), _ptr(key_len), _stream()), "tocvp_mha_f32"))
    return o


# self-attention on q / k / v operand planes written by the projection's epilogue (csrc/attn_planes.hip, round 5): the kernel
# copies the planes instead of splitting / transposing fp32 rows per query block.  TOCVP_MHA_PLANES=0: fp32 hand-over
_MHA_PLANES = os.environ.get("TOCVP_MHA_PLANES", "1") != "0"


def mha_planes_ok(heads, E):
    """ the plane-input attention serves head dim 64 under the default f16x3 arithmetic, outside the range-checked pass
    (a plane-producing epilogue saturates without a check of its own: the checked pass keeps fp32 hand-overs) """
    return (_MHA_PLANES and _ATTN_QK16 and not _CHECK_RANGE and active_nsplit() == 22 and E == heads * 64 and E % 64 == 0)


def mha_planes(q, q_col, k, k_col, v, v_col, B, Tq, Tk, heads, scale, key_len=None, out_split=0):
    """
    q: SplitAct whose planes (B * Tq, 2, ldq) hold the query projection in columns q_col .. q_col + E; k, v likewise with
    (B * Tk, 2, ld) planes (the three may share one fused projection).  Returns (B, Tq, E) fp32, or a SplitAct (fp16 planes)
    if out_split == 22.  Bit-identical to ``mha`` on the fp32 values the planes encode.
    """
    dh = 64
    E = heads * dh
    for name, t, rows in (("q", q, B * Tq), ("k", k, B * Tk), ("v", v, B * Tk)):
        pl = t.planes
        assert pl.dtype == torch.float16 and pl.is_cuda and pl.is_contiguous() and pl.shape[0] == rows and pl.shape[1] == 2, \
            (name, pl.dtype, tuple(pl.shape), rows)
    if key_len is not None:
        assert key_len.dtype == torch.int32 and key_len.is_cuda and key_len.numel() == B
    dev = q.planes.device
    o = osp = None
    if out_split:
        assert out_split == 22
        osp = _alloc_planes(B * Tq, 22, E, dev)
    else:
        o = torch.empty((B, Tq, E), device=dev, dtype=torch.float32)
    qp, kp, vp = q.planes.data_ptr() + 2 * q_col, k.planes.data_ptr() + 2 * k_col, v.planes.data_ptr() + 2 * v_col
    ldq, ldk, ldv = q.planes.shape[2], k.planes.shape[2], v.planes.shape[2]

    def run():
        _check(lib().tocvp_mha_planes_f16(qp, ldq, kp, ldk, vp, ldv, _ptr(o), E, _ptr(osp), B, heads, Tq, Tq, Tk, dh, float(scale),
                                          _ptr(key_len), _stream()), "tocvp_mha_planes_f16")
    _timed(lambda: f"mha_{B}x{heads}x{Tq}x{Tk}x{dh}", 4.0 * B * heads * Tq * Tk * dh, run)
    return SplitAct(osp, (B, Tq, E)) if out_split else o


def slot_attn_workspace(B, N, device):
    """ workspace of the slot-attention kernel for THIS (B, N): ticket words zeroed once (tocvp_slot_attn_ws_init);
    every iteration leaves them zero again, so the same tensor serves all iterations of that shape """
    need = lib().tocvp_slot_attn_ws_bytes(B, N)
    ws = torch.empty((need + 3) // 4, device=device, dtype=torch.float32)
    _check(lib().tocvp_slot_attn_ws_init(_ptr(ws), ws.numel() * 4, _stream()), "tocvp_slot_attn_ws_init")
    return ws


def slot_attn_iter(q, k, v, scale, eps, attn_out=None, ws=None):
    """ q (B, Ks, D); k, v (B, N, D) views with row stride ldkv -> updates (B, Ks, D). """
    B, Ks, D = q.shape
    N = k.shape[1]
    _dev_f32(q, "q"), _dev_f32(k, "k"), _dev_f32(v, "v")
    assert q.is_contiguous()
    assert k.stride(2) == 1 and v.stride(2) == 1 and k.stride(1) == v.stride(1)
    assert k.stride(0) == N * k.stride(1) and v.stride(0) == N * v.stride(1)
    if ws is None:
        ws = slot_attn_workspace(B, N, q.device)
    assert ws.numel() * 4 >= lib().tocvp_slot_attn_ws_bytes(B, N)
    if _CHECK_RANGE:            # q, k and v are split into fp16 planes of 2^8 x inside the kernel (saturating)
        _check_f16_range(max(absmax(q), absmax(k), absmax(v)), "slot attention q / k / v")
    upd = torch.empty((B, Ks, D), device=q.device, dtype=torch.float32)
    # units = algorithmic HBM bytes: k and v read once (B * 2 * N * D * sizeof)
    _timed(lambda: f"slot_attn_{B}x{Ks}x{N}x{D}", 2.0 * B * N * D * 4, lambda: _check(
        lib().tocvp_slot_attn_iter_f32(_ptr(q), _ptr(k), _ptr(v), k.stride(1), _ptr(upd),
                                       _ptr(attn_out), B, Ks, N, D, float(scale), float(eps),
                                       _ptr(ws), ws.numel() * 4, _stream()),
        "tocvp_slot_attn_iter_f32"))
    return upd


def slot_attn_iter_planes(q, kv_planes, scale, eps, attn_out=None, ws=None):
    """
    q (B, Ks, D) fp32; kv_planes (B, N, 2, 2 D) fp16 operand planes of the fused [k | v] projection
    (linear(..., out_split=22)) -> updates (B, Ks, D).
    """
    B, Ks, D = q.shape
    N = kv_planes.shape[1]
    _dev_f32(q, "q")
    assert q.is_contiguous() and kv_planes.is_contiguous() and kv_planes.dtype == torch.float16
    assert tuple(kv_planes.shape) == (B, N, 2, 2 * D), (kv_planes.shape, (B, N, 2, 2 * D))
    if ws is None:
        ws = slot_attn_workspace(B, N, q.device)
    assert ws.numel() * 4 >= lib().tocvp_slot_attn_ws_bytes(B, N)
    upd = torch.empty((B, Ks, D), device=q.device, dtype=torch.float32)
    _timed(lambda: f"slot_attn_{B}x{Ks}x{N}x{D}", 2.0 * B * N * D * 4, lambda: _check(
        lib().tocvp_slot_attn_iter_planes_f32(_ptr(q), _ptr(kv_planes), _ptr(upd), _ptr(attn_out), B, Ks, N, D,
                                              float(scale), float(eps), _ptr(ws), ws.numel() * 4, _stream()),
        "tocvp_slot_attn_iter_planes_f32"))
    return upd


def gru_gates(gi, gh, h):
    rows, D = h.reshape(-1, h.shape[-1]).shape
    out = torch.empty_like(h)
    _check(lib().tocvp_gru_gates_f32(_ptr(gi), _ptr(gh), _ptr(h), _ptr(out), rows, D, _stream()),
           "tocvp_gru_gates_f32")
    return out


def pos_embed(proj_weight, proj_bias, H, W):
    """ SoftPositionEmbed addend (H, W, C) from the 1x1-conv parameters (C,4,1,1), (C,). """
    C = proj_weight.shape[0]
    w = proj_weight.reshape(C, 4).contiguous()
    out = torch.empty((H, W, C), device=w.device, dtype=torch.float32)
    _check(lib().tocvp_pos_embed_f32(_ptr(w), _ptr(proj_bias), _ptr(out), H, W, C, _stream()),
           "tocvp_pos_embed_f32")
    return out


_CONV_FRAG = {}


def conv_frag_weights(wp):
    """
    Packed conv weights (taps..., Cout, Cin) fp32 -> the fp16 operand planes of 2^10 w in MFMA-fragment order that the f16x3
    convolutions of conv3x3.hip read straight from L2 (tocvp_split_weights_frag_f16 on the (taps x Cout, Cin) matrix); cached
    per packed tensor, rebuilt when it changes.
    """
    key = id(wp)
    hit = _CONV_FRAG.get(key)
    if hit is not None and hit[0]() is wp and hit[1] == (wp._version, wp.data_ptr()):
        return hit[2]
    if len(_CONV_FRAG) > 1024:
        for k_ in [k_ for k_, v in _CONV_FRAG.items() if v[0]() is None]:
            del _CONV_FRAG[k_]
    _dev_f32(wp, "conv weights")
    assert wp.is_contiguous()
    Cin = wp.shape[-1]
    rows = wp.numel() // Cin
    out = torch.empty((rows, 2, Cin), device=wp.device, dtype=torch.float16)
    _check(lib().tocvp_split_weights_frag_f16(_ptr(wp), _ptr(out), rows, Cin, _stream()), "tocvp_split_weights_frag_f16")
    _CONV_FRAG[key] = (weakref.ref(wp), (wp._version, wp.data_ptr()), out)
    return out


def pack_conv_weights(w):
    """ (Cout, Cin, k, k) -> (k*k, Cout, Cin) """
    Cout, Cin, k, _ = w.shape
    w = w.contiguous()
    out = torch.empty((k * k, Cout, Cin), device=w.device, dtype=torch.float32)
    _check(lib().tocvp_pack_conv_weights_f32(_ptr(w), _ptr(out), Cout, Cin, k, _stream()),
           "tocvp_pack_conv_weights_f32")
    return out


def conv5x5_in3(x, w, bias):
    """ x: (n, 3, H, W) view whose images are contiguous planes; returns NHWC (n, H, W, Cout). """
    n, C, H, W = x.shape
    assert C == 3 and x.stride(3) == 1 and x.stride(2) == W and x.stride(1) == H * W
    Cout = w.shape[0]
    y = torch.empty((n, H, W, Cout), device=x.device, dtype=torch.float32)
    _check(lib().tocvp_conv5x5_in3_f32(_ptr(x), x.stride(0) if n > 1 else 3 * H * W,
                                       _ptr(w.contiguous()), _ptr(bias), _ptr(y), n, H, W, Cout,
                                       _stream()), "tocvp_conv5x5_in3_f32")
    return y


def conv5x5(x, wp, bias, relu=True, out=None, precision="fp32"):
    """
    NHWC (n, H, W, Cin) -> (n, H, W, Cout) with packed weights (25, Cout, Cin).
    precision "f16x3": split fp16 operands (fp32-class, |x| < 255; Cin, Cout % 32 == 0), else fp32 MFMA.
    """
    n, H, W, Cin = x.shape
    Cout = wp.shape[1]
    assert x.is_contiguous() and wp.shape[0] == 25 and wp.shape[2] == Cin
    if out is None:
        out = torch.empty((n, H, W, Cout), device=x.device, dtype=torch.float32)
    split = precision == "f16x3" and Cin % 32 == 0 and Cout % 32 == 0 and H % 8 == 0
    if split and _CHECK_RANGE:
        _check_f16_range(absmax(x), "conv5x5 (f16x3) input")
        _check_f16_weight_range(wp, "conv5x5 (f16x3)")
    def run():
        if split:
            _check(lib().tocvp_conv5x5_f16x3_f32(_ptr(x), _ptr(conv_frag_weights(wp)), _ptr(bias), _ptr(out), n, H, W, Cin,
                                                 Cout, int(bool(relu)), _stream()),
                   "tocvp_conv5x5_f16x3_f32")
            return
        _check(lib().tocvp_conv5x5_f32(_ptr(x), None, 0, _ptr(wp), _ptr(bias), _ptr(out), n, H, W,
                                       Cin, Cout, int(bool(relu)), _stream()), "tocvp_conv5x5_f32")
    if TIMER is not None:
        TIMER.wrap(f"conv5x5_{Cin}_{Cout}", n, run)
    else:
        run()
    return out


def conv5x5_collapsed(cpos, S, wp, bias, relu=True, out=None):
    """
    Decoder layer 1 fed by the analytically collapsed layer 0:
    cpos (H, W, Cin) = conv0(pos)+bias0, S (n, 25, Cin) per-slot tap sums -> (n, H, W, Cout).
    """
    H, W, Cin = cpos.shape
    n = S.shape[0]
    Cout = wp.shape[1]
    assert cpos.is_contiguous() and S.is_contiguous() and S.shape[1:] == (25, Cin)
    if out is None:
        out = torch.empty((n, H, W, Cout), device=cpos.device, dtype=torch.float32)
    def run():
        _check(lib().tocvp_conv5x5_f32(_ptr(cpos), _ptr(S), 1, _ptr(wp), _ptr(bias), _ptr(out), n, H,
                                       W, Cin, Cout, int(bool(relu)), _stream()),
               "tocvp_conv5x5_f32")
    if TIMER is not None:
        TIMER.wrap(f"conv5x5_{Cin}_{Cout}", n, run)
    else:
        run()
    return out


def dec_tapsum(w):
    """ (Cout, Cin, 5, 5) -> (25, Cout, Cin) border-class tap sums """
    Cout, Cin = w.shape[:2]
    out = torch.empty((25, Cout, Cin), device=w.device, dtype=torch.float32)
    _check(lib().tocvp_dec_tapsum_f32(_ptr(w.contiguous()), _ptr(out), Cout, Cin, _stream()),
           "tocvp_dec_tapsum_f32")
    return out


def _tail_outputs(F, K, H, W, dev, out):
    """
    (imgs, recons, masks, clamped or None) and their frame strides for the decoder tail kernels.  ``out`` = optional
    (imgs, recons, masks[, clamped]) views to write into: each frame contiguous, frames any distance apart (e.g.
    ``full.view(B, P, 3, H, W)[:, t]``: frame b of this call lands at row b * P + t of the full tensor).
    """
    if out is not None:
        imgs, recons, masks = out[:3]
        clamped = out[3] if len(out) > 3 else None
        assert imgs.shape == (F, 3, H, W) and recons.shape == (F, K, 3, H, W) and masks.shape == (F, K, 1, H, W)
        for t_ in (imgs, recons, masks) + ((clamped,) if clamped is not None else ()):
            _dev_f32(t_, "tail output")
            assert t_[0].is_contiguous() if F else True, "every frame of a tail output must be contiguous"
        if clamped is not None:
            assert clamped.shape == imgs.shape and (F <= 1 or clamped.stride(0) == imgs.stride(0))
    else:
        imgs = torch.empty((F, 3, H, W), device=dev, dtype=torch.float32)
        recons = torch.empty((F, K, 3, H, W), device=dev, dtype=torch.float32)
        masks = torch.empty((F, K, 1, H, W), device=dev, dtype=torch.float32)
        clamped = None
    fs = [t_.stride(0) if F > 1 else t_[0].numel() if F else 0 for t_ in (imgs, recons, masks)]
    return imgs, recons, masks, clamped, fs


def dec_tail(x, w, bias, F, K, out=None):
    """
    x: (F*K, H, W, Cin) NHWC -> recons_imgs (F,3,H,W), recons (F,K,3,H,W), masks (F,K,1,H,W).
    ``out`` = optional (imgs, recons, masks[, clamped]) views to write into (see _tail_outputs).
    """
    n, H, W, Cin = x.shape
    assert n == F * K and x.is_contiguous()
    dev = x.device
    imgs, recons, masks, clamped, fs = _tail_outputs(F, K, H, W, dev, out)
    ws = torch.empty(9 * Cin * 4, device=dev, dtype=torch.float32)
    _check(lib().tocvp_dec_tail_placed_f32(_ptr(x), _ptr(w.contiguous()), _ptr(bias), _ptr(imgs), _ptr(recons),
                                           _ptr(masks), _ptr(clamped), fs[0], fs[1], fs[2], F, K, H, W, Cin,
                                           _ptr(ws), ws.numel() * 4, _stream()), "tocvp_dec_tail_placed_f32")
    return imgs, recons, masks


def pack_tail_taps_f16x3(w):
    """ tail Conv2d(64 -> 4, k = 3) weight (4, 64, 3, 3) -> tap-matrix image of the folded tail (conv_f16x3.hip) """
    _dev_f32(w, "tail weight")
    assert tuple(w.shape) == (4, 64, 3, 3)
    if _CHECK_RANGE:
        _check_f16_weight_range(w, "folded decoder tail")
    out = torch.empty(lib().tocvp_tail_taps_f16x3_bytes() // 2, device=w.device, dtype=torch.float16)
    _check(lib().tocvp_pack_tail_taps_f16x3(_ptr(w.contiguous()), _ptr(out), _stream()), "tocvp_pack_tail_taps_f16x3")
    return out


def conv5x5_dec_f16x3_tail(x, wf, bias, taps, relu=True, out=None, pm_in=False, planes=False):
    """
    Last hidden decoder layer with the tail folded into its epilogue (tocvp_conv5x5_dec_f16x3_tail_f32): returns the
    (n, 36, H, W) tap products; ``x`` as for conv5x5_dec_f16x3 (NHWC, or pass-major / operand planes with pm_in).
    """
    n, H, W, Cin = x.shape
    assert x.is_contiguous() and Cin == 64
    planes = bool(planes) and bool(pm_in) and not _CHECK_RANGE   # as the producing call decided
    if _CHECK_RANGE:
        _check_f16_range(absmax(x), "conv5x5_dec_f16x3 (folded tail) input")
    if out is None:
        out = torch.empty((n, 36, H, W), device=x.device, dtype=torch.float32)

    def run():
        _check(lib().tocvp_conv5x5_dec_f16x3_tail_f32(_ptr(x), _ptr(wf), _ptr(bias), _ptr(taps), _ptr(out), n, H, W,
                                                      int(bool(relu)), int(bool(pm_in)) | (8 if planes else 0),
                                                      _stream()), "tocvp_conv5x5_dec_f16x3_tail_f32")
    if TIMER is not None:
        # work units of the launch timer are slot images of 0.839 GFLOP (one 5 x 5 layer); the folded tail adds its
        # 2 x 36 x 64 x H x W = 18.9 MFLOP per 64 x 64 slot image to this launch
        TIMER.wrap("conv5x5_64_64", n * (1.0 + (2.0 * 36 * 64) / (2.0 * 25 * 64 * 64)), run)
    else:
        run()
    return out


def dec_tail_sum(products, bias, F, K, out=None):
    """ tap products (F*K, 36, H, W) -> recons_imgs (F,3,H,W), recons (F,K,3,H,W), masks (F,K,1,H,W); ``out`` as for
    dec_tail """
    n, T, H, W = products.shape
    assert n == F * K and T == 36 and products.is_contiguous()
    imgs, recons, masks, clamped, fs = _tail_outputs(F, K, H, W, products.device, out)
    _check(lib().tocvp_dec_tail_sum_placed_f32(_ptr(products), _ptr(bias), _ptr(imgs), _ptr(recons), _ptr(masks),
                                               _ptr(clamped), fs[0], fs[1], fs[2], F, K, H, W, _stream()),
           "tocvp_dec_tail_sum_placed_f32")
    return imgs, recons, masks


def text_embed(tokens, tok_emb, pos_emb, gamma, beta, eps):
    B, L = tokens.shape
    D = tok_emb.shape[1]
    assert tokens.dtype == torch.int64 and tokens.is_cuda and tokens.is_contiguous()
    out = torch.empty((B, L, D), device=tokens.device, dtype=torch.float32)
    _check(lib().tocvp_text_embed_f32(_ptr(tokens), _ptr(tok_emb), _ptr(pos_emb), _ptr(gamma),
                                      _ptr(beta), _ptr(out), B, L, D, tok_emb.shape[0], float(eps),
                                      _stream()), "tocvp_text_embed_f32")
    return out


def slot_init(mu, sigma, noise):
    """ mu, sigma: (..., D) with D elements; noise: (B, K, D) on device -> mu + sigma * noise """
    D = noise.shape[-1]
    _dev_f32(noise, "noise")
    noise = noise.contiguous()
    out = torch.empty_like(noise)
    _check(lib().tocvp_slot_init_f32(_ptr(mu), _ptr(sigma), _ptr(noise), _ptr(out),
                                     noise.numel() // D, D, _stream()), "tocvp_slot_init_f32")
    return out


def split_conv_weights_bf16(w):
    """ (Cout, Cin, 5, 5) fp32 -> (25, Cout, 2*Cin) bf16: [hi | lo] split operands """
    Cout, Cin = w.shape[:2]
    out = torch.empty((25, Cout, 2 * Cin), device=w.device, dtype=torch.bfloat16)
    _check(lib().tocvp_split_conv_weights_bf16(_ptr(w.contiguous()), _ptr(out), Cout, Cin,
                                               _stream()), "tocvp_split_conv_weights_bf16")
    return out


def split_conv_weights_frag_bf16(w):
    """ (64, 64, 5, 5) fp32 -> MFMA-fragment-order split weights (25*2*2*2*2*64*8,) bf16 """
    Cout, Cin = w.shape[:2]
    out = torch.empty((25 * Cout * Cin * 2,), device=w.device, dtype=torch.bfloat16)
    _check(lib().tocvp_split_conv_weights_frag_bf16(_ptr(w.contiguous()), _ptr(out), Cout, Cin,
                                                    _stream()), "tocvp_split_conv_weights_frag_bf16")
    return out


_CONV_WD = os.environ.get("TOCVP_CONV_WD", "1") != "0"


def conv5x5_bf16x3(x, wsplit, bias, relu=True, out=None, collapsed=None, wfrag=None, gate=None):
    """
    64->64 5x5 conv with split-bf16 (bf16x3) operands, fp32 NHWC in/out.
    collapsed=(cpos, S): layer-1 mode fed by the analytically collapsed decoder layer 0.
    gate (n, H, W, Cout): the output is zeroed where gate <= 0 (data gradient through the ReLU of the layer
    below, fused into the store); needs wfrag, excludes relu / collapsed.
    """
    if collapsed is not None:
        cpos, S = collapsed
        H, W, Cin = cpos.shape
        n = S.shape[0]
        assert cpos.is_contiguous() and S.is_contiguous() and S.shape[1:] == (25, Cin)
        xin, aux, mode, dev = cpos, S, 1, cpos.device
    else:
        n, H, W, Cin = x.shape
        assert x.is_contiguous()
        xin, aux, mode, dev = x, None, 0, x.device
    Cout = wsplit.shape[1]
    assert wsplit.dtype == torch.bfloat16 and wsplit.shape == (25, Cout, 2 * Cin)
    if not _CONV_WD and gate is None:
        wfrag = None
    if out is None:
        out = torch.empty((n, H, W, Cout), device=dev, dtype=torch.float32)
    code = int(bool(relu))
    if gate is not None:
        assert collapsed is None and not relu and wfrag is not None
        assert gate.is_contiguous() and gate.shape == out.shape and gate.dtype == torch.float32
        aux, code = gate, 2

    def run():
        _check(lib().tocvp_conv5x5_bf16x3_f32(_ptr(xin), _ptr(aux), mode, _ptr(wsplit), _ptr(wfrag),
                                              _ptr(bias), _ptr(out), n, H, W, Cin, Cout,
                                              code, _stream()), "tocvp_conv5x5_bf16x3_f32")
    if TIMER is not None:
        TIMER.wrap(f"conv5x5_{Cin}_{Cout}", n, run)
    else:
        run()
    return out


def split_conv_weights_f16f8(w):
    """ (64, 64, 5, 5) fp32 -> (wf16, wf8) fragment-order weight images of the f16+fp8 hybrid conv """
    Cout, Cin = w.shape[:2]
    _check_f16_weight_range(w, "decoder conv (f16f8)")
    wf16 = torch.empty(lib().tocvp_conv_weights_f16f8_bytes(0), device=w.device, dtype=torch.uint8)
    wf8 = torch.empty(lib().tocvp_conv_weights_f16f8_bytes(1), device=w.device, dtype=torch.uint8)
    _check(lib().tocvp_split_conv_weights_f16f8(_ptr(w.contiguous()), _ptr(wf16), _ptr(wf8), Cout, Cin,
                                                _stream()), "tocvp_split_conv_weights_f16f8")
    return wf16, wf8


def conv5x5_f16f8(x, wimgs, bias, relu=True, out=None, collapsed=None, pm_in=False, pm_out=False):
    """
    64->64 5x5 conv with hybrid f16 + fp8 split operands (tocvp_conv5x5_f16f8_f32), fp32 in/out.
    wimgs = split_conv_weights_f16f8(weight); collapsed=(cpos, S): layer-1 mode (see conv5x5_bf16x3).
    x / out are (n, H, W, 64) NHWC; with pm_in / pm_out the same buffer holds the pass-major layout
    (n, 4, H, W, 16) used between consecutive decoder layers (the tensor shape stays (n, H, W, 64)).
    """
    if collapsed is not None:
        cpos, S = collapsed
        H, W, Cin = cpos.shape
        n = S.shape[0]
        assert cpos.is_contiguous() and S.is_contiguous() and S.shape[1:] == (25, Cin)
        xin, aux, mode, dev = cpos, S, 1, cpos.device
    else:
        n, H, W, Cin = x.shape
        assert x.is_contiguous()
        xin, aux, mode, dev = x, None, 0, x.device
    wf16, wf8 = wimgs
    Cout = bias.shape[0]
    if _CHECK_RANGE:    # layer-1 mode: |relu(cpos + S[cls])| <= max|cpos| + max|S|
        _check_f16_range(absmax(xin) + (absmax(aux) if mode == 1 else 0.0),
                         "conv5x5_f16f8 input")
    if out is None:
        out = torch.empty((n, H, W, Cout), device=dev, dtype=torch.float32)

    def run():
        _check(lib().tocvp_conv5x5_f16f8_f32(_ptr(xin), _ptr(aux), mode, _ptr(wf16), _ptr(wf8),
                                             _ptr(bias), _ptr(out), n, H, W, Cin, Cout,
                                             int(bool(relu)), int(bool(pm_in)) | (int(bool(pm_out)) << 1),
                                             _stream()), "tocvp_conv5x5_f16f8_f32")
    if TIMER is not None:
        TIMER.wrap(f"conv5x5_{Cin}_{Cout}", n, run)
    else:
        run()
    return out


def split_conv_weights_dec_f16x3(w):
    """ (64, 64, 5, 5) fp32 -> fragment-order [Wh | Wl] fp16 weight image of the f16x3 decoder conv """
    Cout, Cin = w.shape[:2]
    # checked once per weight version (the image is cached by the caller): saturation is never silent
    _check_f16_weight_range(w, "decoder conv (f16x3)")
    wf = torch.empty(lib().tocvp_conv_weights_dec_f16x3_bytes(), device=w.device, dtype=torch.uint8)
    _check(lib().tocvp_split_conv_weights_dec_f16x3(_ptr(w.contiguous()), _ptr(wf), Cout, Cin, _stream()),
           "tocvp_split_conv_weights_dec_f16x3")
    return wf




# output transform of the vertical Winograd F(4, 5) (csrc/conv_wino.hip): rows p^a over the points 0, 1, -1, 2, -2, 1/2, -1/2, inf
_WINO_AT = [[1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 0.0],
            [0.0, 1.0, -1.0, 2.0, -2.0, 0.5, -0.5, 0.0],
            [0.0, 1.0, 1.0, 4.0, 4.0, 0.25, 0.25, 0.0],
            [0.0, 1.0, -1.0, 8.0, -8.0, 0.125, -0.125, 1.0]]


def split_conv_weights_wino_f16x3(w):
    """
    (64, 64, 5, 5) fp32 -> (wf, coef): the fragment-order fp16 planes of the eight transformed weight rows of the
    Winograd decoder conv, each scaled by the largest power of two that keeps it below 2^14 (chosen here from the
    weights at hand: nothing saturates, whatever their range), and the 32 output-transform coefficients
    AT[a][xi] / (16 * scale[xi]) as a ctypes float array (kernel arguments of conv5x5_dec_wino).
    """
    Cout, Cin = w.shape[:2]
    w = w.contiguous()
    amax = torch.empty(8, device=w.device, dtype=torch.float32)
    _check(lib().tocvp_split_conv_weights_wino_f16x3(_ptr(w), None, None, _ptr(amax), Cout, Cin, _stream()),
           "tocvp_split_conv_weights_wino_f16x3 (absmax)")
    amax = [float(v) for v in amax.tolist()]
    if not all(math.isfinite(v) for v in amax):
        raise ValueError("decoder conv weights are not finite")
    scales = [2.0 ** max(-40, min(40, math.floor(math.log2(16384.0 / max(v, 1e-30))))) for v in amax]
    sc = (ctypes.c_float * 8)(*scales)
    wf = torch.empty(lib().tocvp_conv_weights_wino_f16x3_bytes(), device=w.device, dtype=torch.uint8)
    _check(lib().tocvp_split_conv_weights_wino_f16x3(_ptr(w), _ptr(wf), ctypes.cast(sc, ctypes.c_void_p), None, Cout, Cin,
                                                     _stream()), "tocvp_split_conv_weights_wino_f16x3")
    coef = (ctypes.c_float * 32)(*[_WINO_AT[a][q] / (16.0 * scales[q]) for a in range(4) for q in range(8)])
    return wf, coef


def conv5x5_dec_wino(x, wpack, bias, relu=True, out=None, collapsed=None, in_mode=2, out_mode=0, tail_taps=None,
                     auto_scale=False, gate=None):
    """
    64 -> 64 5x5 conv as vertical Winograd F(4, 5) with split-fp16 products (tocvp_conv5x5_dec_wino_f16x3_f32).
    wpack = split_conv_weights_wino_f16x3(weight).  in_mode 0: x is (n, 4, H, W, 16) fp32 holding 16 * activation (what
    out_mode 1 writes), 2: NHWC fp32; collapsed = (cpos, S): first layer.  out_mode 0 NHWC fp32, 1 the x 16 pass-major
    buffer, 2 fp16 operand planes for conv5x5_dec_f16x3(_tail) (never with the range check on), 3 with tail_taps: the
    (n, 36, H, W) tap products of the folded decoder tail.
    auto_scale (in_mode 2): max |x| is reduced on the device (tocvp_absmax_f32, no host round trip) and the kernel scales
    the input by the matching power of two -- inputs of any magnitude, e.g. data gradients; gate (out_mode 0): the output
    is zeroed where gate <= 0.
    """
    wf, coef = wpack
    amax = None
    if auto_scale:
        assert collapsed is None and in_mode == 2
        amax = torch.empty(1, device=x.device, dtype=torch.int32)
        _check(lib().tocvp_absmax_f32(_ptr(x), x.numel(), _ptr(amax), _stream()), "tocvp_absmax_f32")
    if gate is not None:
        assert out_mode == 0 and gate.is_contiguous() and gate.dtype == torch.float32
    if collapsed is not None:
        cpos, S = collapsed
        H, W, Cin = cpos.shape
        n = S.shape[0]
        assert cpos.is_contiguous() and S.is_contiguous() and S.shape[1:] == (25, Cin)
        xin, aux, mode, dev = cpos, S, 1, cpos.device
        if _CHECK_RANGE:
            _check_f16_range(absmax(cpos) + absmax(S), "conv5x5_dec_wino layer-0 bound")
    else:
        assert x.is_contiguous() and in_mode in (0, 2)
        if in_mode == 0:
            n, _, H, W, _ = x.shape
        else:
            n, H, W, _ = x.shape
        xin, aux, mode, dev = x, None, in_mode, x.device
        if _CHECK_RANGE and not auto_scale:
            _check_f16_range(absmax(xin) * (1.0 / 16.0 if in_mode == 0 else 1.0), "conv5x5_dec_wino input")
    assert not (out_mode == 2 and _CHECK_RANGE)
    assert (out_mode == 3) == (tail_taps is not None)
    if out is None:
        shape = {0: (n, H, W, 64), 1: (n, 4, H, W, 16), 2: (n, 4, H, W, 16), 3: (n, 36, H, W)}[out_mode]
        out = torch.empty(shape, device=dev, dtype=torch.float32)

    def run():
        _check(lib().tocvp_conv5x5_dec_wino_f16x3_f32(_ptr(xin), _ptr(aux), mode, _ptr(wf),
                                                      ctypes.cast(coef, ctypes.c_void_p), _ptr(bias), _ptr(tail_taps),
                                                      _ptr(out), n, H, W, int(bool(relu)), out_mode, _ptr(amax), _ptr(gate),
                                                      _stream()),
               "tocvp_conv5x5_dec_wino_f16x3_f32")
    if TIMER is not None:
        TIMER.wrap("conv5x5_64_64", n, run)
    else:
        run()
    return out


def conv5x5_dec_f16x3(x, wf, bias, relu=True, out=None, collapsed=None, pm_in=False, pm_out=False, planes=False):
    """
    64->64 5x5 conv with split-fp16 operands (tocvp_conv5x5_dec_f16x3_f32, fp32-class), fp32 in/out.
    wf = split_conv_weights_dec_f16x3(weight); collapsed / pm_in / pm_out as conv5x5_f16f8.
    planes: the pass-major buffers (pm_in / pm_out) hold the fp16 operand planes [Xh | Xl] per pixel and pass instead
    of 16 floats -- written by the producing layer's epilogue, moved into LDS by DMA in the consuming layer; same
    bytes, bit-identical results.  Never with the range check on (the planes cannot be inspected as fp32).
    """
    # (producer and consumer of a buffer are called with the same flag and see the same two switches)
    planes = bool(planes) and bool(pm_in or pm_out) and not _CHECK_RANGE
    if collapsed is not None:
        cpos, S = collapsed
        H, W, Cin = cpos.shape
        n = S.shape[0]
        assert cpos.is_contiguous() and S.is_contiguous() and S.shape[1:] == (25, Cin)
        xin, aux, mode, dev = cpos, S, 1, cpos.device
        if _CHECK_RANGE:    # |relu(cpos + S[cls])| <= max|cpos| + max|S|
            _check_f16_range(absmax(cpos) + absmax(S), "conv5x5_dec_f16x3 layer-0 bound")
    else:
        n, H, W, Cin = x.shape
        assert x.is_contiguous()
        xin, aux, mode, dev = x, None, 0, x.device
        if _CHECK_RANGE:
            _check_f16_range(absmax(xin), "conv5x5_dec_f16x3 input")
    Cout = bias.shape[0]
    if out is None:
        out = torch.empty((n, H, W, Cout), device=dev, dtype=torch.float32)

    def run():
        _check(lib().tocvp_conv5x5_dec_f16x3_f32(_ptr(xin), _ptr(aux), mode, _ptr(wf), _ptr(bias), _ptr(out),
                                                 n, H, W, Cin, Cout, int(bool(relu)),
                                                 int(bool(pm_in)) | (int(bool(pm_out)) << 1) |
                                                 (8 if planes else 0), _stream()),
               "tocvp_conv5x5_dec_f16x3_f32")
    if TIMER is not None:
        TIMER.wrap(f"conv5x5_{Cin}_{Cout}", n, run)
    else:
        run()
    return out


def psnr_ssim(preds, targets, clamp01=True, want_psnr=True, want_ssim=True):
    """ preds, targets (N, C, H, W) fp32 on device -> (psnr (N,), ssim (N,)) (None if not wanted) """
    _dev_f32(preds, "preds"), _dev_f32(targets, "targets")
    assert preds.shape == targets.shape and preds.dim() == 4
    preds, targets = preds.contiguous(), targets.contiguous()
    N, C, H, W = preds.shape
    ws = torch.empty(max(1, N * C * 2), device=preds.device, dtype=torch.float32)
    psnr = torch.empty(N, device=preds.device, dtype=torch.float32) if want_psnr else None
    ssim = torch.empty(N, device=preds.device, dtype=torch.float32) if want_ssim else None
    _check(lib().tocvp_psnr_ssim_f32(_ptr(preds), _ptr(targets), _ptr(psnr), _ptr(ssim), N, C, H, W,
                                     int(bool(clamp01)), _ptr(ws), ws.numel() * 4, _stream()),
           "tocvp_psnr_ssim_f32")
    return psnr, ssim


def conv3x3(x, wp, scale, shift, relu=True, upsample2=False, precision="fp32"):
    """
    NHWC (n, SH, SW, Cin) -> (n, H, W, Cout): 3x3 conv (pad 1) + per-channel scale/shift (+ReLU) on
    the (optionally nearest-x2-upsampled) input; wp packed (9, Cout, Cin).
    precision: "fp32" (exact fp32 MFMA) or "f16x3" (split fp16 operands, fp32-class, |x| < 255).
    """
    n, SH, SW, Cin = x.shape
    H, W = (2 * SH, 2 * SW) if upsample2 else (SH, SW)
    Cout = wp.shape[1]
    assert x.is_contiguous() and wp.shape[0] == 9 and wp.shape[2] == Cin
    y = torch.empty((n, H, W, Cout), device=x.device, dtype=torch.float32)
    if precision == "f16x3" and _CHECK_RANGE:
        _check_f16_range(absmax(x), "conv3x3 (f16x3) input")
        _check_f16_weight_range(wp, "conv3x3 (f16x3)")
    fn = lib().tocvp_conv3x3_f16x3_f32 if precision == "f16x3" else lib().tocvp_conv3x3_f32
    wk = conv_frag_weights(wp) if precision == "f16x3" else wp        # f16x3: fragment-order fp16 planes of the packed weights
    _check(fn(_ptr(x), _ptr(wk), _ptr(scale), _ptr(shift), _ptr(y), n, H, W, Cin, Cout, int(bool(relu)),
              int(bool(upsample2)), _stream()), "tocvp_conv3x3_" + precision)
    return y


def pack_conv3x3_up2_weights(weight):
    """
    (Cout, Cin, 3, 3) -> (4 phases, 4 taps, Cout, Cin) for conv3x3_up2: phase (a, b) = output pixel (2 y + a, 2 x + b) of
    "nearest x2 -> 3x3 conv"; its tap (i, j) reads source pixel (y + i + a - 1, x + j + b - 1) and holds the sum of the
    3x3 taps that fall on that pixel (summed in float64, rounded once).
    """
    rows = {(0, 0): (0,), (0, 1): (1, 2), (1, 0): (0, 1), (1, 1): (2,)}
    w = weight.detach().double()
    out = torch.empty((4, 4) + tuple(w.shape[:2]), device=w.device, dtype=torch.float64)
    for a in range(2):
        for b in range(2):
            for i in range(2):
                for j in range(2):
                    out[2 * a + b, 2 * i + j] = sum(w[:, :, dy, dx] for dy in rows[(a, i)] for dx in rows[(b, j)])
    return out.float().contiguous()


def conv3x3_up2(x, wphase, scale, shift, relu=True):
    """
    NHWC (n, SH, SW, Cin) -> (n, 2 SH, 2 SW, Cout): nearest x2 upsampling + 3x3 conv (pad 1) + per-channel scale / shift
    (+ ReLU) as four 2x2 phase convolutions over the source image (f16x3 arithmetic; tocvp_conv3x3_up2_f16x3_f32);
    wphase from pack_conv3x3_up2_weights.
    """
    n, SH, SW, Cin = x.shape
    Cout = wphase.shape[2]
    assert x.is_contiguous() and wphase.is_contiguous() and tuple(wphase.shape[:2]) == (4, 4) and wphase.shape[3] == Cin
    y = torch.empty((n, 2 * SH, 2 * SW, Cout), device=x.device, dtype=torch.float32)
    if _CHECK_RANGE:
        _check_f16_range(absmax(x), "conv3x3_up2 (f16x3) input")
        _check_f16_weight_range(wphase, "conv3x3_up2 (f16x3)")
    _timed(lambda: f"conv3x3_up2_{n}x{SH}x{SW}x{Cin}x{Cout}", 2.0 * n * SH * SW * 16 * Cin * Cout, lambda: _check(
        lib().tocvp_conv3x3_up2_f16x3_f32(_ptr(x), _ptr(conv_frag_weights(wphase)), _ptr(scale), _ptr(shift), _ptr(y), n, SH, SW, Cin, Cout,
                                          int(bool(relu)), _stream()), "tocvp_conv3x3_up2_f16x3_f32"))
    return y


def slot_composite(decoded, feat_dim=None):
    """
    decoded (B, K, N, ld) -> recons (B, N, F), masks (B, K, N); features in [..., :F], alpha at
    [..., F], F = feat_dim (default ld - 1; ld > F + 1 when the producing GEMM padded its width).
    """
    B, Ks, N, ld = decoded.shape
    F_ = ld - 1 if feat_dim is None else int(feat_dim)
    _dev_f32(decoded, "decoded")
    decoded = decoded.contiguous()
    recons = torch.empty((B, N, F_), device=decoded.device, dtype=torch.float32)
    masks = torch.empty((B, Ks, N), device=decoded.device, dtype=torch.float32)
    _check(lib().tocvp_slot_composite_f32(_ptr(decoded), _ptr(recons), _ptr(masks), B, Ks, N, F_, ld,
                                          _stream()), "tocvp_slot_composite_f32")
    return recons, masks


def bilinear_resize_nhwc_to_nchw(x, channels, out_h, out_w):
    """ x (n, SH, SW, Cs) NHWC, first ``channels`` channels -> (n, channels, out_h, out_w) """
    n, SH, SW, Cs = x.shape
    assert x.is_contiguous()
    y = torch.empty((n, channels, out_h, out_w), device=x.device, dtype=torch.float32)
    _check(lib().tocvp_bilinear_resize_f32(_ptr(x), _ptr(y), n, channels, Cs, SH, SW, out_h, out_w,
                                           _stream()), "tocvp_bilinear_resize_f32")
    return y


def rms_norm(x, gamma, eps):
    """ T5LayerNorm over the last axis """
    _dev_f32(x, "x")
    D = x.shape[-1]
    x2 = x.reshape(-1, D)
    if not x2.is_contiguous():
        x2 = x2.contiguous()
    y = torch.empty_like(x2)
    _check(lib().tocvp_rmsnorm_f32(_ptr(x2), _ptr(gamma), _ptr(y), x2.shape[0], D, float(eps),
                                   _stream()), "tocvp_rmsnorm_f32")
    return y.reshape(x.shape)


def embedding(ids, table):
    """ ids int64 (...,) on device, table (V, D) -> (..., D) """
    assert ids.dtype == torch.int64 and ids.is_cuda
    ids = ids.contiguous()
    V, D = table.shape
    out = torch.empty((*ids.shape, D), device=table.device, dtype=torch.float32)
    _check(lib().tocvp_embedding_f32(_ptr(ids), _ptr(table), _ptr(out), ids.numel(), D, V, _stream()),
           "tocvp_embedding_f32")
    return out

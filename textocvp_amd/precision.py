"""
Arithmetic selection.  Every module reads its own knob (TOCVP_DECODER_PRECISION, TOCVP_PREDICTOR_PRECISION,
TOCVP_ENCODER_PRECISION, TOCVP_ENCODER_GEMM_PRECISION, TOCVP_DECODER_MLP_PRECISION,
TOCVP_DECODER_CNN_PRECISION, TOCVP_ATTN_QK); ``TOCVP_PRECISION=fp32`` is the master switch that puts all of
them on the exact fp32 MFMA kernels (the all-fp32 mode of DESIGN.md section 3).
"""

import os

__all__ = ["knob"]


def knob(name, default):
    """ value of the environment knob ``name``; the master switch TOCVP_PRECISION=fp32 overrides the default """
    if name in os.environ:
        return os.environ[name]
    if os.environ.get("TOCVP_PRECISION", "") == "fp32":
        return "fp32"
    return default

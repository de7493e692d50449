"""Arithmetic of the contractions on the hot path -- a process-wide option, `--arithmetic` of main.py / test.py /
bench.py (reference semantics: every product an fp32 fma, `src/lib/models/networks/pose_dla_dcn.py` through cuDNN and
the DCNv2 extension).

  "split_bf16" (default)  float32 tensors in and out; every convolution, the heads and the DCNv2 forward / backward
                          contract on the bf16 matrix cores as a*b ~ ah*bh + ah*bl + al*bh (fp32 accumulate,
                          ~2^-16 relative error per product; the small 64-channel DCN maps stay on the exact
                          f32 MFMA).  Measured against the reference's golden network outputs: ~1e-5 of each head's
                          max-norm (bench.py `max_rel_err_vs_golden`), against the 1e-3 bar of the north star.
  "exact_f32"             the exact fp32 chain everywhere: library (MIOpen) convolutions, f32-MFMA DCNv2 forward
                          and backward, f32 heads.  The continuity point of the round-1 numbers (`exact_f32` in the
                          bench line).

Replaces the environment variables of round 2 (CP_CONV_MFMA, CP_DCN_BWD_F32, CP_DCN_FWD_F32): nothing on the call
path reads the environment."""
from . import _C

MODES = ("split_bf16", "exact_f32")
_mode = "split_bf16"


def configure(mode):
    """Select the arithmetic for every model built or called afterwards in this process."""
    global _mode
    if mode not in MODES:
        raise ValueError("arithmetic must be one of %s" % (MODES,))
    from .models.networks import conv3x3
    from .models.networks.DCNv2.dcn_v2 import DCN
    exact = mode == "exact_f32"
    conv3x3._ENABLED = not exact
    conv3x3._WGRAD = not exact
    DCN.train_contraction = "f32" if exact else "auto"
    DCN.infer_contraction = "f32" if exact else "auto"
    DCN.backward_flags = _C.DCN_BWD_EXACT_F32 if exact else 0
    _mode = mode


def current():
    return _mode


def describe():
    """What the bench line's `dtype` says."""
    if _mode == "exact_f32":
        return "f32 (exact fp32 fma chains: library convolutions, f32-MFMA DCNv2)"
    return ("f32 I/O; split-bf16 x3 contraction (a*b ~ ah*bh + ah*bl + al*bh on bf16 MFMA, ~2^-16 per product) in the "
            "convolutions, the heads and DCNv2; f32 accumulate")

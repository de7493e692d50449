"""ctypes binding of libcenterpoly_hip.so (the C ABI declared in include/centerpoly_hip.h).

There is NO CPU fallback: every op in this package goes through this library and
raises if it is missing or if a tensor is not on a HIP device.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int32, c_int64, c_size_t, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libcenterpoly_hip.so")

CP_OK = 0
REP = {"cartesian": 0, "polar": 1, "polar_fixed": 2}
L1_PLAIN, L1_POLAR, L1_POLAR_FIXED, L1_RELU20, L1_SMOOTH = 0, 1, 2, 3, 4
DCN_BWD_EXACT_F32, DCN_BWD_NARROW_TILES, DCN_BWD_ROUND1_KERNELS = 1, 2, 4
DCN_CONTRACTION = {"f32": 0, "bf16x3": 1, "bf16x3_region": 3}     # (+1 = "..._PREPARED": weights already in the workspace)


class DcnShape(Structure):
    _fields_ = [(n, c_int32) for n in ("B", "Cin", "H", "W", "Cout", "kh", "kw", "stride", "pad",
                                       "dil", "deformable_groups")]


class TargetShape(Structure):
    _fields_ = [(n, c_int32) for n in ("B", "max_objs", "nbr_points", "num_classes", "out_h", "out_w",
                                       "rep", "no_reorder_flip")]


class NativeLibraryMissing(ImportError):
    pass


class NativeError(RuntimeError):
    pass


_lib = None

ABI_VERSION = 3              # CP_ABI_VERSION of include/centerpoly_hip.h this binding was written against
_P = c_void_p
_SIGNATURES = {
    "cp_abi_version": (c_int32, []),
    "cp_strerror": (c_char_p, [c_int32]),
    "cp_build_arch": (c_char_p, []),
    "cp_dcn_v2_forward_workspace_bytes": (c_size_t, [POINTER(DcnShape)]),
    "cp_dcn_v2_forward_kernel": (c_int32, [POINTER(DcnShape), c_int32]),
    "cp_dcn_v2_forward": (c_int32, [POINTER(DcnShape), _P, _P, c_int64, _P, c_int64, c_int32, _P, _P,
                                    _P, _P, c_int32, c_int32, _P, _P, c_size_t, _P]),
    "cp_dcn_v2_forward_fused_supported": (c_int32, [POINTER(DcnShape)]),
    "cp_dcn_v2_forward_fused_workspace_bytes": (c_size_t, [POINTER(DcnShape)]),
    "cp_dcn_v2_forward_fused": (c_int32, [POINTER(DcnShape), _P, _P, _P, _P, _P, _P, _P, c_int32, c_int32, _P, _P, _P,
                                          c_size_t, _P]),
    "cp_dcn_v2_backward_workspace_bytes": (c_size_t, [POINTER(DcnShape)]),
    "cp_dcn_v2_backward": (c_int32, [POINTER(DcnShape), _P, _P, c_int64, _P, c_int64, c_int32, _P, _P,
                                     _P, _P, c_int64, _P, c_int64, _P, _P, c_int32, _P, c_size_t, _P]),
    "cp_upsample2x_add": (c_int32, [_P, _P, _P] + [c_int32] * 4 + [_P]),
    "cp_depthwise_up_forward": (c_int32, [_P, _P, _P, _P] + [c_int32] * 5 + [_P]),
    "cp_depthwise_up_backward": (c_int32, [_P, _P, _P, _P, _P] + [c_int32] * 5 + [_P]),
    "cp_bn_workspace_bytes": (c_size_t, [c_int32, c_int32, c_int64]),
    "cp_bn_act_forward_train": (c_int32, [_P] * 9 + [c_float, c_float, c_int32, c_int32, c_int32, c_int64,
                                                    _P, c_size_t, _P]),
    "cp_bn_act_backward": (c_int32, [_P] * 7 + [c_int32] + [_P] * 4 + [c_int32, c_int32, c_int64, _P,
                                                                       c_size_t, _P]),
    "cp_bias_act_inplace": (c_int32, [_P, _P, _P, c_int32, c_int32, c_int64, c_int32, _P]),
    "cp_channel_sum_accumulate": (c_int32, [_P, _P, c_int32, c_int32, c_int64, _P]),
    "cp_bias_relu_backward": (c_int32, [_P, _P, _P, _P, c_int32, c_int32, c_int64, _P]),
    "cp_preprocess_warp_normalize": (c_int32, [_P, c_int32, c_int32, _P, _P, _P, c_int32, c_int32, c_int32, _P, _P]),
    "cp_color_aug_workspace_bytes": (c_size_t, []),
    "cp_color_aug_normalize": (c_int32, [_P, c_int64, c_int32, _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    "cp_instance_masks": (c_int32, [_P, _P, c_int32, c_int32, c_int32, c_int32, _P, _P, _P]),
    "cp_polydet_post_process": (c_int32, [_P, _P, c_float, c_int32, c_int32, c_int32, _P, _P]),
    "cp_polydet_targets_workspace_bytes": (c_size_t, [POINTER(TargetShape)]),
    "cp_polydet_targets": (c_int32, [POINTER(TargetShape)] + [_P] * 19 + [_P, c_size_t, _P]),
    "cp_polydet_dense_targets": (c_int32, [POINTER(TargetShape), _P, _P, c_size_t, _P, _P, _P]),
    "cp_conv_direct_supported": (c_int32, [c_int32] * 5),
    "cp_conv_direct_wgrad_supported": (c_int32, [c_int32] * 5),
    "cp_conv_direct_wgrad": (c_int32, [_P, _P, _P] + [c_int32] * 8 + [_P]),
    "cp_conv_direct_forward": (c_int32, [_P, _P, _P, _P] + [c_int32] * 9 + [_P]),
    "cp_conv_direct_forward_ex": (c_int32, [_P, _P, _P, _P] + [c_int32] * 10 + [_P]),
    "cp_conv3x3_mfma_supported": (c_int32, [c_int32] * 4),
    "cp_conv3x3_mfma_weight_bytes": (c_size_t, [c_int32] * 2),
    "cp_conv3x3_mfma_prepare": (c_int32, [_P, c_int32, c_int32, c_int32, _P, _P]),
    "cp_conv3x3_mfma_forward": (c_int32, [_P, _P, _P, _P, _P] + [c_int32] * 6 + [_P]),
    "cp_conv_mfma_weight_bytes": (c_size_t, [c_int32] * 3),
    "cp_conv_mfma_prepare": (c_int32, [_P, c_int32, c_int32, c_int32, c_int32, _P, _P]),
    "cp_conv3x3_s2_wgrad_supported": (c_int32, [c_int32] * 4),
    "cp_conv3x3_s2_wgrad": (c_int32, [_P, _P, _P] + [c_int32] * 5 + [_P]),
    "cp_conv7x7_c3_supported": (c_int32, [c_int32] * 4),
    "cp_conv7x7_c3_weight_bytes": (c_size_t, [c_int32]),
    "cp_conv7x7_c3_prepare": (c_int32, [_P, c_int32, _P, _P]),
    "cp_conv7x7_c3_forward": (c_int32, [_P, _P, _P, _P] + [c_int32] * 6 + [_P]),
    "cp_conv_mfma_prepare_blocks": (c_int32, [c_int32] * 3),
    "cp_conv_mfma_prepare_batch": (c_int32, [_P, c_int32, c_int32, _P]),
    "cp_conv_mfma_forward": (c_int32, [_P, _P, c_int32, _P, _P, _P, _P] + [c_int32] * 6 + [_P]),
    "cp_conv_mfma_forward_strided": (c_int32, [_P, _P, c_int32, _P, _P, _P, _P] + [c_int32] * 7 + [_P]),
    "cp_dla_base_pair_supported": (c_int32, [c_int32] * 2),
    "cp_dla_base_pair_forward": (c_int32, [_P] * 6 + [c_int32] * 3 + [_P]),
    "cp_conv_mfma_forward_split": (c_int32, [_P, c_int32, _P, _P, _P, _P] + [c_int32] * 9 + [_P]),
    "cp_activation_split": (c_int32, [_P, _P] + [c_int32] * 4 + [_P]),
    "cp_activation_unsplit": (c_int32, [_P, _P] + [c_int32] * 4 + [_P]),
    "cp_conv_mfma_input_grad_relu_workspace_bytes": (c_size_t, [c_int32] * 4),
    "cp_conv_mfma_input_grad_relu": (c_int32, [_P] * 5 + [c_int32] * 6 + [_P, c_size_t, _P]),
    "cp_conv3x3_s2_input_grad": (c_int32, [_P, _P, _P, _P] + [c_int32] * 5 + [_P]),
    "cp_conv3x3_mfma_wgrad_supported": (c_int32, [c_int32] * 4),
    "cp_conv3x3_mfma_wgrad": (c_int32, [_P, _P, _P] + [c_int32] * 5 + [_P]),
    "cp_conv_mfma_wgrad": (c_int32, [_P, _P, _P] + [c_int32] * 6 + [_P]),
    "cp_heads_fused_w2_bytes": (c_size_t, [c_int32]),
    "cp_heads_fused_prepare_w2": (c_int32, [_P, c_int32, c_int32, _P, _P]),
    "cp_heads_fused_forward": (c_int32, [_P, _P, _P, _P, _P, _P, _P] + [c_int32] * 6 + [_P]),
    "cp_maxpool2x2_forward": (c_int32, [_P, _P] + [c_int32] * 4 + [_P]),
    "cp_maxpool2x2_backward": (c_int32, [_P, _P, _P] + [c_int32] * 4 + [_P]),
    "cp_conv1x1_act_forward": (c_int32, [_P, c_int64, _P, c_int32, _P, _P, _P, c_int32, c_int32, c_int32, c_int64, _P]),
    "cp_soft_nms": (c_int32, [_P, c_int32, c_int32, c_float, c_float, c_float, c_int32]),
    "cp_polydet_decode_workspace_bytes": (c_size_t, [c_int32] * 5),
    "cp_polydet_decode": (c_int32, [_P, _P, _P, _P] + [c_int32] * 7 + [_P, _P, _P, _P, c_size_t, _P]),
    "cp_polydet_decode_ex": (c_int32, [_P, _P, _P, _P] + [c_int32] * 8 + [_P, _P, _P, _P, c_size_t, _P]),
    "cp_sigmoid_focal_workspace_bytes": (c_size_t, [c_int64]),
    "cp_sigmoid_focal_forward": (c_int32, [_P, _P, c_int64, _P, _P, _P, c_size_t, _P]),
    "cp_sigmoid_focal_backward": (c_int32, [_P, _P, c_int64, _P, _P, _P, _P]),
    "cp_gather_l1_forward": (c_int32, [_P, _P, _P, _P, _P] + [c_int32] * 6 + [c_float, _P, _P]),
    "cp_gather_l1_backward": (c_int32, [_P, _P, _P, _P, _P] + [c_int32] * 6 + [c_float, _P, _P, _P]),
    "cp_mse_workspace_bytes": (c_size_t, []),
    "cp_mse_forward": (c_int32, [_P, _P, c_int64, _P, _P, c_size_t, _P]),
    "cp_mse_backward": (c_int32, [_P, _P, c_int64, _P, _P, _P]),
    "cp_dense_l1_workspace_bytes": (c_size_t, []),
    "cp_dense_l1_forward": (c_int32, [_P, _P, _P, c_int64, c_float, _P, _P, c_size_t, _P]),
    "cp_dense_l1_backward": (c_int32, [_P, _P, _P, c_int64, _P, _P, _P, _P]),
    "cp_poly_iou_order_workspace_bytes": (c_size_t, [c_int32] * 3),
    "cp_poly_iou_order_forward": (c_int32, [_P, _P, _P, _P] + [c_int32] * 6 + [_P, _P, _P, _P,
                                                                               c_size_t, _P]),
    "cp_poly_iou_order_backward": (c_int32, [_P, _P, _P, _P] + [c_int32] * 6 + [_P, _P, _P, _P,
                                                                                c_size_t, _P]),
}

EXPORTS = tuple(_SIGNATURES)


def lib():
    """The loaded library; raises NativeLibraryMissing when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeLibraryMissing(
                "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C centerpoly_amd/csrc`). centerpoly_amd has no CPU fallback." % LIB_PATH)
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        if l.cp_abi_version() != ABI_VERSION:
            raise NativeError("ABI version mismatch")
        _lib = l
    return _lib


def check(rc, what=""):
    if rc != CP_OK:
        raise NativeError("%s failed: %s (%d)" % (what, lib().cp_strerror(rc).decode(), rc))


def ptr(t):
    """Device pointer of a tensor (or NULL).  Refuses host tensors and non-contiguous views."""
    if t is None:
        return None
    if not t.is_cuda:
        raise NativeError("centerpoly_amd ops need HIP device tensors (got %s); there is no CPU "
                          "fallback" % t.device)
    if not t.is_contiguous():
        raise NativeError("tensor must be contiguous")
    return c_void_p(t.data_ptr())


def stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def workspace(nbytes, device):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


class ConvPrepareJob(ctypes.Structure):
    """cp_conv_prepare_job of include/centerpoly_hip.h."""
    _fields_ = [("weight", c_void_p), ("wperm", c_void_p), ("Cin", c_int32), ("Cout", c_int32), ("taps", c_int32),
                ("transposed", c_int32), ("first_block", c_int32), ("reserved", c_int32)]


class _ZeroPool(object):
    """Zero-initialised accumulators (the weight / bias gradients the kernels ADD into) carved out of one zero-filled
    block: one fill kernel per BLOCK bytes handed out instead of one per tensor (the training step asks for ~190 of
    them, most a few KB -- each a launch of its own).  A block is never reused: a fresh one is allocated when the
    current one is used up, and the caching allocator recycles a block only once every tensor carved from it has died,
    so a gradient that outlives the step keeps its memory -- and pins its whole block: tensors from `zeros` are meant to
    die within the step (autograd's grad accumulation copies or adopts them; a retained .grad keeps 32 MB alive, not
    more).  `release()` drops the current block (trainer teardown, train() / eval() switches)."""
    BLOCK = 32 << 20

    def __init__(self):
        self.block, self.off, self.key = None, 0, None

    def release(self):
        self.block, self.off, self.key = None, 0, None

    def take(self, shape, device):
        n = 1
        for d in shape:
            n *= int(d)
        nbytes = (n * 4 + 255) // 256 * 256
        if nbytes > self.BLOCK // 8 or n == 0:
            return torch.zeros(tuple(shape), dtype=torch.float32, device=device)
        dev = torch.device(device)
        if dev.type == "cuda" and dev.index is None:     # 'cuda' and 'cuda:0' are one device: one key, one block
            dev = torch.device("cuda", torch.cuda.current_device())
        key = (dev, torch.cuda.current_stream(dev).cuda_stream)
        if self.block is None or self.key != key or self.off + nbytes > self.BLOCK:
            self.block = torch.zeros(self.BLOCK // 4, dtype=torch.float32, device=device)
            self.off, self.key = 0, key
        t = self.block[self.off // 4: self.off // 4 + n].view(tuple(shape))
        self.off += nbytes
        return t


_zero_pool = _ZeroPool()


def zeros(shape, device):
    """float32 zeros on a HIP device for a kernel to accumulate into (see _ZeroPool).  Not to be retained past the step."""
    return _zero_pool.take(shape, device)


def release_zero_pool():
    """Drop the pool's current block (its memory returns to the allocator once the tensors carved from it have died)."""
    _zero_pool.release()


class _NoEvent(object):
    def record(self):
        pass


class KernelTimer(object):
    """Optional per-launch HIP-event timing on torch's current stream (where every kernel of
    this library is launched).  bench.py enables it around the timed region to get the
    dominant kernel's average launch duration; disabled (None) it costs nothing.
    `watch`: a set of keys -- only those launches get events (an event pair costs a few
    microseconds of stream time, which adds up over ~50 launches per step)."""

    def __init__(self, watch=None):
        self.records = {}
        self.watch = watch

    def start(self, key):
        if self.watch is not None and key not in self.watch:
            return _NoEvent()
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        self.records.setdefault(key, []).append(ev)
        ev[0].record()
        return ev[1]

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for key, evs in self.records.items():
            ms = [a.elapsed_time(b) for a, b in evs]
            out[key] = {"launches": len(ms), "avg_ms": sum(ms) / len(ms), "min_ms": min(ms)}
        return out


kernel_timer = None

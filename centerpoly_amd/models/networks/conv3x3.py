"""Host side of the split-bf16 MFMA 3x3 convolution (csrc/conv_mfma.hip, include/centerpoly_hip.h
cp_conv3x3_mfma_*): the dense 3x3 / stride 1 / pad 1 convolutions of the reference's networks
(src/lib/models/networks/pose_dla_dcn.py BasicBlock :38-66, heads :445-462, DCNv2/dcn_v2.py:137-145
conv_offset_mask; large_hourglass.py convolution :24-37, residual :55-81), which the reference hands
to cuDNN.  float32 in and out; forward and input gradient run on the bf16 matrix cores as three
products of split halves, and so does the weight gradient (cp_conv3x3_mfma_wgrad).

`conv_raw(conv, x)` is what every training call site uses for "conv without its bias";
`conv3x3_infer(x, conv, w, bias, residual, relu)` is the inference call with the fused epilogue.
Shapes the kernel does not take (stride 2, 1x1, 7x7, fewer than 24 input channels, tiny maps) go to the direct
kernel or the library.  `centerpoly_amd.arithmetic.configure("exact_f32")` turns the kernels off (library fp32)."""

import weakref

import torch
import torch.nn.functional as F

from ... import _C

_ENABLED = True              # set by centerpoly_amd.arithmetic.configure
_WGRAD = True
MIN_CIN = 24                 # the contraction steps over 32 input channels: fewer would mostly multiply zeros
MIN_WORKGROUPS = 128         # (of the narrowest tile form) below this the launch cannot fill the 256 CUs


def _workgroups(B, cout, H, W):
    """Workgroups of the narrowest tile form the library picks for small layers: 32 channels x 4 rows x 32 px."""
    return B * ((W + 31) // 32) * ((H + 3) // 4) * ((cout + 31) // 32)


def _fills(B, cin, cout, H, W):
    """Enough workgroups for the chip -- or, for deep inputs, enough for the in-workgroup K split (4 x 4 waves per
    workgroup) to keep it busy: the 27-channel offset convolutions of the deep DCN layers, 16-64 workgroups, still
    beat the library there (256 -> 27 @64x128: 22 us against 49)."""
    n = _workgroups(B, cout, H, W)
    return n >= MIN_WORKGROUPS or (cin >= 256 and n >= 16)


def usable_shape(x, cout):
    """True when a 3x3 / stride 1 / pad 1 convolution of x to `cout` channels is a launch of the MFMA kernel."""
    if not (_ENABLED and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4):
        return False
    B, cin, H, W = x.shape
    return cin >= MIN_CIN and bool(_C.lib().cp_conv3x3_mfma_supported(cin, cout, H, W)) \
        and _fills(B, cin, cout, H, W)


def usable(conv, x):
    """True when conv(x) (without bias) is a launch of the MFMA kernel: 3x3 / pad 1 (stride 1 or 2) or 1x1 (stride 1)."""
    k = conv.kernel_size
    if not (k in ((3, 3), (1, 1)) and conv.padding == (k[0] // 2, k[0] // 2) and conv.dilation == (1, 1)
            and conv.groups == 1 and conv.padding_mode == "zeros"):
        return False
    if conv.stride == (1, 1):
        return usable_shape(x, conv.out_channels)
    if conv.stride == (2, 2) and k == (3, 3) and x.dim() == 4:         # forward only; gradients from the library
        B, cin, H, W = x.shape
        return usable_shape(x, conv.out_channels) and _fills(B, cin, conv.out_channels, (H - 1) // 2 + 1, (W - 1) // 2 + 1)
    return False


class _WeightBank(object):
    """The permuted (split-bf16, fragment-ordered) forms of the model's convolution weights, refreshed by ONE launch per
    optimizer step (cp_conv_mfma_prepare_batch) instead of one launch per use -- a training step uses ~110 forms.
    A form enters the bank the first time a PARAMETER is prepared (temporaries -- weight slices, folded copies -- never
    do); `refresh()` (the trainer calls it right after optimizer.step()) permutes every registered form and records the
    parameters' versions; `get` serves a form only while the parameter is still at that version, so a weight changed
    behind the bank's back (load_state_dict, manual edits) simply takes the per-use path until the next refresh."""

    def __init__(self):
        # (id(weight), cin, cout, code) -> [weakref, wp or None, version, data_ptr, device]; wp is allocated by refresh()
        self.entries = {}
        self.table = None          # per device: (device tensor holding the job structs, njobs, total_blocks, keys in table order)

    def get(self, weight, cin, cout, code):
        if not isinstance(weight, torch.nn.Parameter):
            return None
        key = (id(weight), cin, cout, code)
        e = self.entries.get(key)
        if e is None or e[0]() is not weight:
            # (registration only: the buffer is allocated by the first refresh() -- a weight that is prepared once and
            # never refreshed, e.g. at inference, does not pay for a second copy of its permuted form)
            self.entries[key] = [weakref.ref(weight), None, -1, weight.data_ptr(), weight.device]
            self.table = None
            return None
        # `param.data = ...`, module.to(device), load_state_dict(assign=True) or parameter flattening re-point the storage
        # WITHOUT a version bump: the form is served only while the parameter still lives where the table read it
        if e[1] is None or e[2] != weight._version or e[3] != weight.data_ptr() or e[4] != weight.device:
            return None
        return e[1]

    def refresh(self):
        L = _C.lib()
        dead = [k for k, e in self.entries.items() if e[0]() is None]
        for k in dead:
            del self.entries[k]
        if dead:
            self.table = None
        if not self.entries:
            return
        for k, e in self.entries.items():                      # storage moved (or first refresh): new buffer, new table
            w = e[0]()
            if e[1] is None or e[3] != w.data_ptr() or e[4] != w.device:
                _, cin, cout, _code = k
                e[1] = torch.empty(L.cp_conv_mfma_weight_bytes(cin, cout, w.shape[2] * w.shape[3]), dtype=torch.uint8,
                                   device=w.device)
                e[3], e[4] = w.data_ptr(), w.device
                self.table = None
        if self.table is None:
            by_dev = {}
            for k, e in self.entries.items():
                by_dev.setdefault(e[4], []).append(k)          # keyed on the WEIGHT's device
            self.table = {}
            for dev, keys in by_dev.items():
                jobs = (_C.ConvPrepareJob * len(keys))()
                blk = 0
                for j, k in zip(jobs, keys):
                    w = self.entries[k][0]()
                    _, cin, cout, code = k
                    j.weight, j.wperm = w.data_ptr(), self.entries[k][1].data_ptr()
                    j.Cin, j.Cout, j.taps, j.transposed, j.first_block = cin, cout, w.shape[2] * w.shape[3], code, blk
                    blk += L.cp_conv_mfma_prepare_blocks(cin, cout, j.taps)
                host = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8)
                self.table[dev] = (host.to(dev), len(keys), blk, keys)
        for dev, (tab, n, blocks, keys) in self.table.items():
            with torch.cuda.device(dev):
                _C.check(L.cp_conv_mfma_prepare_batch(_C.ptr(tab), n, blocks, _C.stream()), "cp_conv_mfma_prepare_batch")
            for k in keys:
                e = self.entries[k]
                e[2] = e[0]()._version


_BANK = _WeightBank()


def refresh_weight_bank():
    """Permute every registered convolution weight in one launch (call after optimizer.step())."""
    if _ENABLED:
        _BANK.refresh()


def _prepare(weight, cin, cout, transposed):
    """The permuted form of `weight` ([cout or cin][...][k][k]) for the MFMA kernels: transposed False / True, or the
    kernel's mode code (6: the one-launch stride-2 input gradient)."""
    code = int(transposed) if not isinstance(transposed, bool) else (1 if transposed else 0)
    wp = _BANK.get(weight, cin, cout, code)
    if wp is not None:
        return wp
    L = _C.lib()
    taps = weight.shape[2] * weight.shape[3]
    wp = torch.empty(L.cp_conv_mfma_weight_bytes(cin, cout, taps), dtype=torch.uint8, device=weight.device)
    _C.check(L.cp_conv_mfma_prepare(_C.ptr(weight), cin, cout, taps, code, _C.ptr(wp), _C.stream()), "cp_conv_mfma_prepare")
    return wp


def _launch(x, wp, bias, residual, cout, relu, taps=9, stride=1):
    B, cin, H, W = x.shape
    out = torch.empty((B, cout, (H - 1) // stride + 1, (W - 1) // stride + 1), dtype=torch.float32, device=x.device)
    tag = ("conv3x3_fwd" if stride == 1 else "conv3x3s2_fwd") if taps == 9 else "conv1x1_fwd"
    end = _C.kernel_timer.start((tag, cin, cout, H, W, B)) if _C.kernel_timer is not None else None
    ptrs, chans = (_C.c_void_p * 1)(x.data_ptr()), (_C.c_int32 * 1)(cin)
    _C.check(_C.lib().cp_conv_mfma_forward_strided(ptrs, chans, 1, _C.ptr(wp), _C.ptr(bias), _C.ptr(residual),
                                                   _C.ptr(out), B, H, W, cout, taps, stride, 1 if relu else 0,
                                                   _C.stream()), "cp_conv_mfma_forward_strided")
    if end is not None:
        end.record()
    return out


def conv_infer(xs, owner, w, bias=None, residual=None, relu=False, conv=None, key="_mfma_wperm", x_split=False,
               out_split=False):
    """Inference: 3x3 / pad 1 or 1x1 (stride 1 or 2) convolution of the channel concatenation of `xs` -- read in place,
    no torch.cat -- with the (folded) weight `w`, + bias + residual + ReLU in the kernel's epilogue.  The permuted
    weights are cached on `owner` (under `key`) for as long as `w` is the same, unmodified tensor.  `conv`, when
    given, is the module whose geometry must be the kernel's.  Returns None when the shape is not the kernel's.
    x_split / out_split (one source, 3x3): the input / output is a SPLIT tensor -- the [hi | lo] bf16 planes of
    cp_conv_mfma_forward_split held in a float32 tensor of the logical shape (same bytes; only such a convolution may
    read it).  Bit-identical to the float32 route."""
    k = tuple(w.shape[2:])
    if not _ENABLED or k not in ((1, 1), (3, 3)):
        return None
    if (x_split or out_split) and (len(xs) != 1 or k != (3, 3) or (out_split and residual is not None)
                                   or (x_split and xs[0].shape[1] % 32) or (out_split and w.shape[0] % 8)):
        return None
    stride = 1
    if conv is not None:
        if not (conv.stride in ((1, 1), (2, 2)) and conv.padding == (k[0] // 2, k[0] // 2)
                and conv.dilation == (1, 1) and conv.groups == 1 and conv.padding_mode == "zeros"):
            return None
        stride = conv.stride[0]
    x0 = xs[0]
    if not all(x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[0] == x0.shape[0]
               and x.shape[2:] == x0.shape[2:] for x in xs) or len(xs) > 4:
        return None
    cs = [x.shape[1] for x in xs]
    cin, cout = sum(cs), w.shape[0]
    B, _, H, W = x0.shape
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    L = _C.lib()
    if cin != w.shape[1] or cin < MIN_CIN or (len(xs) > 1 and any(c % 32 for c in cs)) \
            or not all(L.cp_conv3x3_mfma_supported(c, cout, H, W) for c in cs) \
            or not _fills(B, cin, cout, Ho, Wo) or (residual is not None and not residual.is_contiguous()):
        return None
    taps = k[0] * k[1]
    cache = owner.__dict__.get(key)
    if cache is None or cache[0] is not w or cache[1] != w._version:
        wp = torch.empty(L.cp_conv_mfma_weight_bytes(cin, cout, taps), dtype=torch.uint8, device=w.device)
        _C.check(L.cp_conv_mfma_prepare(_C.ptr(w.contiguous()), cin, cout, taps, 0, _C.ptr(wp), _C.stream()),
                 "cp_conv_mfma_prepare")
        cache = (w, w._version, wp)
        owner.__dict__[key] = cache
    xs = [x.contiguous() for x in xs]
    out = torch.empty((B, cout, Ho, Wo), dtype=torch.float32, device=x0.device)
    ptrs = (_C.c_void_p * len(xs))(*[x.data_ptr() for x in xs])
    chans = (_C.c_int32 * len(xs))(*cs)
    tag = "conv3x3_fwd" if taps == 9 and stride == 1 else ("conv3x3s2_fwd" if taps == 9 else "conv1x1_fwd")
    end = _C.kernel_timer.start((tag, cin, cout, H, W, B)) if _C.kernel_timer is not None else None
    if x_split or out_split:
        if x_split and stride != 1:
            return None
        _C.check(L.cp_conv_mfma_forward_split(_C.ptr(xs[0]), 1 if x_split else 0, _C.ptr(cache[2]), _C.ptr(bias),
                                              _C.ptr(residual), _C.ptr(out), 1 if out_split else 0, B, cin, H, W, cout,
                                              taps, stride, 1 if relu else 0, _C.stream()), "cp_conv_mfma_forward_split")
    else:
        _C.check(L.cp_conv_mfma_forward_strided(ptrs, chans, len(xs), _C.ptr(cache[2]), _C.ptr(bias), _C.ptr(residual),
                                                _C.ptr(out), B, H, W, cout, taps, stride, 1 if relu else 0, _C.stream()),
                 "cp_conv_mfma_forward_strided")
    if end is not None:
        end.record()
    return out


def block_infer(x, conv1, wb1, conv2, wb2, skip):
    """relu(conv2(relu(conv1(x) + b1)) + b2 + skip) of a residual block at inference (pose_dla_dcn.py BasicBlock :38-66,
    large_hourglass.py residual :55-81) with the intermediate kept as split planes: conv1's epilogue writes the bf16
    halves conv2's staging would otherwise form from float32, conv2 stages them with 16-byte loads.  None when the pair
    is not two launches of the MFMA kernel (the caller then runs the float32 route)."""
    planes = wb1[0].shape[0]
    if not (_ENABLED and x.is_cuda and planes % 32 == 0 and tuple(wb1[0].shape[2:]) == (3, 3)
            and tuple(wb2[0].shape[2:]) == (3, 3) and conv2.stride == (1, 1)):
        return None
    B, _, H, W = x.shape
    s = conv1.stride[0]
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    if planes < MIN_CIN or not _fills(B, planes, planes, Ho, Wo) \
            or not _C.lib().cp_conv3x3_mfma_supported(planes, planes, Ho, Wo) or (skip is not None and not skip.is_contiguous()):
        return None                                  # conv2 would refuse the split tensor
    y = conv_infer([x], conv1, wb1[0], wb1[1], None, True, conv=conv1, out_split=True)
    if y is None:
        return None
    z = conv_infer([y], conv2, wb2[0], wb2[1], skip, True, conv=conv2, x_split=True)
    if z is None:
        raise _C.NativeError("block_infer: the second convolution refused a split tensor its first convolution wrote")
    return z


def conv3x3_infer(x, owner, w, bias=None, residual=None, relu=False, conv=None, key="_mfma_wperm"):
    """conv_infer for one input tensor (kept for the call sites that predate the 1x1 / multi-source form)."""
    return conv_infer([x], owner, w, bias, residual, relu, conv=conv, key=key)


def s2_input_grad(xshape, weight, go):
    """Input gradient of a 3x3 / stride 2 / pad 1 convolution: one launch of the MFMA convolution over grad_out with
    the four parity classes' accumulators (cp_conv3x3_s2_input_grad); None where the kernel does not take the shape."""
    L = _C.lib()
    B, cin, H, W = xshape
    cout = weight.shape[0]
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    if not (_ENABLED and go.is_cuda and go.dtype == torch.float32 and tuple(weight.shape[2:]) == (3, 3)
            and tuple(go.shape) == (B, cout, Ho, Wo) and cout >= MIN_CIN and L.cp_conv3x3_mfma_supported(cout, cin, Ho, Wo)
            and cin * H * W * 4 < 0x7FFFFFF0 and _fills(B, cout, cin, Ho, Wo)):
        return None
    go = go.contiguous()
    gx = torch.empty(xshape, dtype=torch.float32, device=go.device)
    end = _C.kernel_timer.start(("conv3x3s2_igrad", cout, cin, Ho, Wo, B)) if _C.kernel_timer is not None else None
    wp = _prepare(weight, cout, cin, 6)
    _C.check(L.cp_conv3x3_s2_input_grad(_C.ptr(go), _C.ptr(wp), None, _C.ptr(gx), B, cin, H, W, cout, _C.stream()),
             "cp_conv3x3_s2_input_grad")
    if end is not None:
        end.record()
    return gx


def stem_infer(x, conv, w, bias, relu):
    """Inference: a 7x7 / pad 3 convolution of a 3-channel image, stride 1 (DLA's base_layer) or 2 (the Hourglass stem),
    on the bf16 matrix cores (cp_conv7x7_c3_forward) with the (folded) weight `w`, + bias + ReLU in the epilogue; the
    permuted weights are cached on `conv` for as long as `w` is the same, unmodified tensor.  None when the module is
    not that convolution (or the arithmetic is exact_f32)."""
    if not (_ENABLED and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and conv.kernel_size == (7, 7)
            and conv.stride in ((1, 1), (2, 2)) and conv.padding == (3, 3) and conv.dilation == (1, 1) and conv.groups == 1
            and conv.in_channels == 3 and x.shape[1] == 3 and conv.padding_mode == "zeros"):
        return None
    L = _C.lib()
    B, _, H, W = x.shape
    cout, stride = w.shape[0], conv.stride[0]
    if not L.cp_conv7x7_c3_supported(cout, H, W, stride) or B > 65535:
        return None
    cache = conv.__dict__.get("_stem_wperm")
    if cache is None or cache[0] is not w or cache[1] != w._version:
        wp = torch.empty(L.cp_conv7x7_c3_weight_bytes(cout), dtype=torch.uint8, device=w.device)
        _C.check(L.cp_conv7x7_c3_prepare(_C.ptr(w.contiguous()), cout, _C.ptr(wp), _C.stream()), "cp_conv7x7_c3_prepare")
        cache = (w, w._version, wp)
        conv.__dict__["_stem_wperm"] = cache
    out = torch.empty((B, cout, (H - 1) // stride + 1, (W - 1) // stride + 1), dtype=torch.float32, device=x.device)
    _C.check(L.cp_conv7x7_c3_forward(_C.ptr(x.contiguous()), _C.ptr(cache[2]), _C.ptr(bias), _C.ptr(out), B, H, W, cout,
                                     stride, 1 if relu else 0, _C.stream()), "cp_conv7x7_c3_forward")
    return out


def s2_weight_grad(x, weight, go):
    """Weight gradient of a 3x3 / stride 2 / pad 1 convolution (cp_conv3x3_s2_wgrad); None where the kernel does not take
    the shape or the MFMA weight gradients are switched off."""
    L = _C.lib()
    B, cin, H, W = x.shape
    cout = weight.shape[0]
    if not (_ENABLED and _WGRAD and x.is_cuda and x.dtype == torch.float32 and tuple(weight.shape[2:]) == (3, 3)
            and cin >= MIN_CIN and L.cp_conv3x3_s2_wgrad_supported(cin, cout, H, W)
            and tuple(go.shape) == (B, cout, (H - 1) // 2 + 1, W // 2)):
        return None
    gw = _C.zeros(weight.shape, weight.device)
    end = _C.kernel_timer.start(("conv3x3s2_wgrad", cin, cout, H, W, B)) if _C.kernel_timer is not None else None
    _C.check(L.cp_conv3x3_s2_wgrad(_C.ptr(x), _C.ptr(go.contiguous()), _C.ptr(gw), B, cin, H, W, cout, _C.stream()),
             "cp_conv3x3_s2_wgrad")
    if end is not None:
        end.record()
    return gw


class _Conv3x3Fn(torch.autograd.Function):
    """Training: forward and input gradient on the matrix cores (the input gradient is the same kernel
    over grad_out with the transposed, flipped weights); weight gradient by cp_conv3x3_mfma_wgrad."""

    @staticmethod
    def forward(ctx, x, weight, stride=1):
        ctx.save_for_backward(x, weight)
        ctx.stride = stride
        cout, cin = weight.shape[0], weight.shape[1]
        return _launch(x, _prepare(weight, cin, cout, False), None, None, cout, False, weight.shape[2] * weight.shape[3],
                       stride)

    @staticmethod
    def backward(ctx, go):
        x, weight = ctx.saved_tensors
        go = go.contiguous()
        if ctx.stride != 1:                           # stride 2: forward and input gradient from the kernel
            gx = None
            if ctx.needs_input_grad[0]:
                gx = s2_input_grad(x.shape, weight, go)
                if gx is None:
                    gx = torch.nn.grad.conv2d_input(x.shape, weight, go, stride=ctx.stride, padding=1)
            gw = None
            if ctx.needs_input_grad[1]:
                gw = s2_weight_grad(x, weight, go)
                if gw is None:
                    gw = torch.nn.grad.conv2d_weight(x, weight.shape, go, stride=ctx.stride, padding=1)
            return gx, gw, None
        return grads(x, weight, go, ctx.needs_input_grad[0], ctx.needs_input_grad[1], min_k=MIN_CIN) + (None,)


class _Conv3x3SkipFn(torch.autograd.Function):
    """conv(x) together with x itself as a second output, for a block whose skip connection starts at the convolution's
    input (BasicBlock, src/lib/models/networks/pose_dla_dcn.py:32-60: `out += residual` with residual = x): the gradient
    arriving over the skip is added in the input-gradient kernel's epilogue (its `residual` operand) instead of by
    autograd's separate accumulation pass over the map."""

    @staticmethod
    def forward(ctx, x, weight):
        ctx.save_for_backward(x, weight)
        cout, cin = weight.shape[0], weight.shape[1]
        y = _launch(x, _prepare(weight, cin, cout, False), None, None, cout, False, weight.shape[2] * weight.shape[3])
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, go, gskip):
        x, weight = ctx.saved_tensors
        cout, cin = weight.shape[0], weight.shape[1]
        B, _, H, W = x.shape
        if go is None:                                  # (only the skip was used)
            return gskip, None
        go = go.contiguous()
        gx = None
        if ctx.needs_input_grad[0] and gskip is not None and gskip.is_contiguous() \
                and _C.lib().cp_conv3x3_mfma_supported(cout, cin, H, W):
            gx = _launch(go, _prepare(weight, cout, cin, True), None, gskip, cin, False, weight.shape[2] * weight.shape[3])
            _, gw = grads(x, weight, go, False, ctx.needs_input_grad[1], min_k=MIN_CIN)
            return gx, gw
        gx, gw = grads(x, weight, go, ctx.needs_input_grad[0], ctx.needs_input_grad[1], min_k=MIN_CIN)
        if gx is not None and gskip is not None:
            gx = gx + gskip
        return gx, gw


def conv_raw_skip(conv, x):
    """(conv(x) without bias, x) for a block whose skip connection is the convolution's own input: one autograd node
    that adds the skip's gradient inside the input-gradient launch (_Conv3x3SkipFn); None when the shape is not the
    MFMA kernel's or the stride is not 1."""
    if conv.bias is not None or conv.stride != (1, 1) or conv.kernel_size != (3, 3) or not usable(conv, x) \
            or conv.in_channels < MIN_CIN \
            or conv.out_channels < MIN_CIN or not torch.is_grad_enabled() or not x.requires_grad:
        return None
    return _Conv3x3SkipFn.apply(x.contiguous(), conv.weight)


class _ConcatConv1x1Fn(torch.autograd.Function):
    """Training: the 1x1 convolution of a channel concatenation (`Root`, src/lib/models/networks/pose_dla_dcn.py:148-166:
    conv(torch.cat(xs, 1))) without the concatenated copy: the forward reads the sources in place (as inference does),
    the backward runs one input-gradient launch per source over its slice of the weight -- each gradient leaves
    contiguous, where slices of the concatenation's gradient had to be copied before the next kernel could take them --
    and one weight-gradient launch per source into its slice."""

    @staticmethod
    def forward(ctx, weight, *xs):
        ctx.save_for_backward(weight, *xs)
        cout, cin = weight.shape[0], weight.shape[1]
        x0 = xs[0]
        B, _, H, W = x0.shape
        out = torch.empty((B, cout, H, W), dtype=torch.float32, device=x0.device)
        end = _C.kernel_timer.start(("conv1x1_fwd", cin, cout, H, W, B)) if _C.kernel_timer is not None else None
        n = len(xs)
        ptrs, chans = (_C.c_void_p * n)(*[x.data_ptr() for x in xs]), (_C.c_int32 * n)(*[x.shape[1] for x in xs])
        _C.check(_C.lib().cp_conv_mfma_forward(ptrs, chans, n, _C.ptr(_prepare(weight, cin, cout, False)), None, None,
                                               _C.ptr(out), B, H, W, cout, 1, 0, _C.stream()), "cp_conv_mfma_forward")
        if end is not None:
            end.record()
        return out

    @staticmethod
    def backward(ctx, go):
        weight, xs = ctx.saved_tensors[0], ctx.saved_tensors[1:]
        go = go.contiguous()
        gxs, gws, c0 = [], [], 0
        for i, x in enumerate(xs):
            c = x.shape[1]
            w_i = weight[:, c0:c0 + c].contiguous()
            gx, gw = grads(x, w_i, go, ctx.needs_input_grad[1 + i], ctx.needs_input_grad[0], min_k=MIN_CIN)
            gxs.append(gx)
            gws.append(gw)
            c0 += c
        gw = torch.cat(gws, 1) if ctx.needs_input_grad[0] else None
        return (gw,) + tuple(gxs)


def concat_conv1x1(conv, xs):
    """conv(torch.cat(xs, 1)) for a bias-free 1x1 convolution in training, the sources read in place
    (_ConcatConv1x1Fn); None when the shapes are not the MFMA kernel's."""
    if not (_ENABLED and conv.bias is None and conv.kernel_size == (1, 1) and conv.stride == (1, 1)
            and conv.padding == (0, 0) and conv.groups == 1 and 2 <= len(xs) <= 4 and torch.is_grad_enabled()):
        return None
    x0 = xs[0]
    if not all(x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[0] == x0.shape[0]
               and x.shape[2:] == x0.shape[2:] and x.shape[1] % 32 == 0 for x in xs):
        return None
    B, _, H, W = x0.shape
    cin, cout = sum(x.shape[1] for x in xs), conv.out_channels
    L = _C.lib()
    if cin != conv.in_channels or cout < MIN_CIN or not _fills(B, cin, cout, H, W) \
            or not all(L.cp_conv3x3_mfma_supported(x.shape[1], cout, H, W) and L.cp_conv3x3_mfma_supported(cout, x.shape[1], H, W)
                       for x in xs):
        return None
    return _ConcatConv1x1Fn.apply(conv.weight, *[x.contiguous() for x in xs])


def mfma_enabled():
    return _ENABLED


def grads(x, weight, go, want_x=True, want_w=True, min_k=1, residual=None):
    """(grad_x, grad_weight) of a stride-1 3x3 / pad 1 or 1x1 convolution from grad_out: the MFMA kernels where
    they take the shape (the input gradient contracts over Cout: below `min_k` output channels it goes to the
    library)."""
    L = _C.lib()
    cout, cin = weight.shape[0], weight.shape[1]
    taps, pad = weight.shape[2] * weight.shape[3], weight.shape[2] // 2
    B, _, H, W = x.shape
    gx = gw = None
    if want_x:                                          # (+ residual: a gradient of the same input to accumulate onto)
        if cout >= min_k and L.cp_conv3x3_mfma_supported(cout, cin, H, W) \
                and (residual is None or residual.is_contiguous()):
            gx = _launch(go, _prepare(weight, cout, cin, True), None, residual, cin, False, taps)
        else:
            gx = torch.nn.grad.conv2d_input(x.shape, weight, go, padding=pad)
            if residual is not None:
                gx = gx + residual
    if want_w:
        if _WGRAD and L.cp_conv3x3_mfma_wgrad_supported(cin, cout, H, W):
            gw = _C.zeros(weight.shape, weight.device)
            tag = "conv3x3_wgrad" if taps == 9 else "conv1x1_wgrad"
            end = _C.kernel_timer.start((tag, cin, cout, H, W, B)) if _C.kernel_timer is not None else None
            _C.check(L.cp_conv_mfma_wgrad(_C.ptr(x), _C.ptr(go), _C.ptr(gw), B, cin, H, W, cout, taps, _C.stream()),
                     "cp_conv_mfma_wgrad")
            if end is not None:
                end.record()
        else:
            gw = torch.nn.grad.conv2d_weight(x, weight.shape, go, padding=pad)
    return gx, gw


class _ConvBiasActFn(torch.autograd.Function):
    """Training: stride-1 convolution + bias (+ ReLU) with the bias / ReLU in the kernel's epilogue (no separate pass
    over the output); backward: ReLU mask + bias gradient in one pass (cp_bias_relu_backward) or a channel sum of
    grad_out (cp_channel_sum_accumulate), then the convolution's gradients."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu):
        cout, cin = weight.shape[0], weight.shape[1]
        y = _launch(x, _prepare(weight, cin, cout, False), bias, None, cout, relu, weight.shape[2] * weight.shape[3])
        ctx.relu = relu
        ctx.save_for_backward(x, weight, y if relu else None)
        return y

    @staticmethod
    def backward(ctx, go):
        x, weight, y = ctx.saved_tensors
        go = go.contiguous()
        B, C, H, W = go.shape
        L = _C.lib()
        gb = _C.zeros((C,), go.device)
        if ctx.relu:
            g = torch.empty_like(go)
            _C.check(L.cp_bias_relu_backward(_C.ptr(y), _C.ptr(go), _C.ptr(g), _C.ptr(gb), B, C, H * W, _C.stream()),
                     "cp_bias_relu_backward")
        else:
            g = go
            _C.check(L.cp_channel_sum_accumulate(_C.ptr(go), _C.ptr(gb), B, C, H * W, _C.stream()),
                     "cp_channel_sum_accumulate")
        gx, gw = grads(x, weight, g, ctx.needs_input_grad[0], ctx.needs_input_grad[1], min_k=MIN_CIN)
        return gx, gw, (gb if ctx.needs_input_grad[2] else None), None


def conv_bias_act(conv, x, relu):
    """conv(x) + bias (+ ReLU), differentiable, as ONE forward launch when the shape is the MFMA kernel's
    (stride 1); None otherwise (the caller falls back to conv_raw + its separate epilogue)."""
    if conv.bias is None or conv.stride != (1, 1) or not usable(conv, x):
        return None
    B, _, H, W = x.shape
    if (H * W) % 4 != 0 or B * conv.out_channels > 65535:          # (limits of the backward's epilogue kernels)
        return None
    return _ConvBiasActFn.apply(x.contiguous(), conv.weight, conv.bias, bool(relu))


class _HeadsFn(torch.autograd.Function):
    """The detection heads in training -- each Conv2d(3x3, bias) -> ReLU -> Conv2d(1x1, bias) on the SAME feature map
    (src/lib/models/networks/pose_dla_dcn.py:445-462, 480-483) -- as ONE autograd node: both forwards of every head with
    their epilogues in the kernels; backward per head: the 1x1 convolution's input gradient leaves the kernel already
    masked by the ReLU with the 3x3 bias gradient summed in its epilogue (cp_conv_mfma_input_grad_relu) -- no separate
    pass over the [B][head_conv][H][W] map (4 x 537 MB at the training size) -- and the heads' gradients of the shared
    input are accumulated by the input-gradient launches themselves (`residual` operand) instead of three passes of
    autograd's accumulation over the 134 MB map.  Arguments: x, then (w1, b1, w2, b2) per head; one output per head."""

    @staticmethod
    def forward(ctx, x, *params):
        n = len(params) // 4
        outs, saved = [], [x]
        for h in range(n):
            w1, b1, w2, b2 = params[4 * h:4 * h + 4]
            c1, cin, c2 = w1.shape[0], w1.shape[1], w2.shape[0]
            y = _launch(x, _prepare(w1, cin, c1, False), b1, None, c1, True, 9)
            outs.append(_launch(y, _prepare(w2, c1, c2, False), b2, None, c2, False, 1))
            saved += [w1, w2, y]
        ctx.save_for_backward(*saved)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gos):
        saved = ctx.saved_tensors
        x = saved[0]
        L = _C.lib()
        gx, gparams = None, []
        for h, go in enumerate(gos):
            w1, w2, y = saved[1 + 3 * h:4 + 3 * h]
            if go is None:                              # this head's output was not used
                gparams += [None, None, None, None]
                continue
            go = go.contiguous()
            B, c2, H, W = go.shape
            c1 = w1.shape[0]
            need = ctx.needs_input_grad[1 + 4 * h:5 + 4 * h]
            gb2 = _C.zeros((c2,), go.device)
            _C.check(L.cp_channel_sum_accumulate(_C.ptr(go), _C.ptr(gb2), B, c2, H * W, _C.stream()), "cp_channel_sum_accumulate")
            _, gw2 = grads(y, w2, go, False, need[2])
            g = torch.empty_like(y)
            gb1 = _C.zeros((c1,), go.device)
            end = _C.kernel_timer.start(("conv1x1_igrad_relu", c2, c1, H, W, B)) if _C.kernel_timer is not None else None
            ws = _C.workspace(L.cp_conv_mfma_input_grad_relu_workspace_bytes(B, c1, H, W), go.device)
            _C.check(L.cp_conv_mfma_input_grad_relu(_C.ptr(go), _C.ptr(_prepare(w2, c2, c1, True)), _C.ptr(y), _C.ptr(g),
                                                    _C.ptr(gb1), B, c1, H, W, c2, 1, _C.ptr(ws), ws.numel(), _C.stream()),
                     "cp_conv_mfma_input_grad_relu")
            if end is not None:
                end.record()
            gxh, gw1 = grads(x, w1, g, ctx.needs_input_grad[0], need[0], min_k=MIN_CIN, residual=gx)
            gx = gxh if gxh is not None else gx
            gparams += [gw1, gb1, gw2, gb2]
        return (gx,) + tuple(gparams)


def _head_ok(fc, x):
    c0, c2 = fc[0], fc[2]
    if not (c0.bias is not None and c2.bias is not None and c0.kernel_size == (3, 3) and c2.kernel_size == (1, 1)
            and c0.stride == (1, 1) and c2.stride == (1, 1) and usable(c0, x)):
        return False
    B, _, H, W = x.shape
    L = _C.lib()
    c1, co = c0.out_channels, c2.out_channels
    return bool(c2.padding == (0, 0) and c2.dilation == (1, 1) and c2.groups == 1 and c1 >= MIN_CIN
                and L.cp_conv3x3_mfma_supported(c1, co, H, W) and L.cp_conv3x3_mfma_supported(co, c1, H, W)
                and _fills(B, c1, co, H, W) and (H * W) % 4 == 0 and B * max(c1, co) <= 65535)


def heads_train(fcs, x):
    """[fc(x) for fc in fcs] for heads Sequential(Conv2d(3x3, bias), ReLU, Conv2d(1x1, bias)) over one feature map in
    training as ONE autograd node (_HeadsFn); None when a shape is not the MFMA kernel's."""
    if not fcs or not all(_head_ok(fc, x) for fc in fcs):
        return None
    params = []
    for fc in fcs:
        params += [fc[0].weight, fc[0].bias, fc[2].weight, fc[2].bias]
    return list(_HeadsFn.apply(x.contiguous(), *params))


def head_train(fc, x):
    """fc(x) for one head (see heads_train); None when a shape is not the MFMA kernel's."""
    out = heads_train([fc], x)
    return None if out is None else out[0]


def conv_raw(conv, x):
    """conv(x) WITHOUT its bias, differentiable: the MFMA kernel for its shapes, the library otherwise."""
    if usable(conv, x):
        return _Conv3x3Fn.apply(x.contiguous(), conv.weight, conv.stride[0])
    return F.conv2d(x, conv.weight, None, conv.stride, conv.padding, conv.dilation, conv.groups)

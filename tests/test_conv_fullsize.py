"""The split-bf16 convolution kernels at the launches bench.py times (too big for a float64 reference over the whole
map): size-independent properties through the C ABI -- impulse / zero-padding response, linearity, bit-identical
reruns, batch-split equality -- plus equality with the library's fp32 convolution on a random 32-row band (the band
with one halo row each side is convolved by F.conv2d and compared on its interior), the forward / weight-gradient
adjoint identity at the training shape, and one Hourglass-104 property test at 1024 x 2048 (BASELINE config 4)."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from centerpoly_amd import _C, synth

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 2e-5


def P(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _t(tag, shape, scale=1.0):
    return torch.from_numpy(synth.normal("convfull/" + tag, shape) * np.float32(scale)).to(DEV)


def _conv(x, w, bias=None, residual=None, relu=False):
    L = _C.lib()
    B, cin, H, W = x.shape
    cout = w.shape[0]
    wp = torch.empty(L.cp_conv3x3_mfma_weight_bytes(cin, cout), dtype=torch.uint8, device=DEV)
    _C.check(L.cp_conv3x3_mfma_prepare(P(w), cin, cout, 0, P(wp), _C.stream()), "prepare")
    out = torch.full((B, cout, H, W), float("nan"), device=DEV)
    _C.check(L.cp_conv3x3_mfma_forward(P(x), P(wp), P(bias), P(residual), P(out), B, cin, H, W, cout,
                                       1 if relu else 0, _C.stream()), "forward")
    return out


def _band_check(x, w, out, bias=None, rows=32, seed=0):
    """out[:, :, y0 : y0 + rows] against the library's fp32 convolution of the same rows (+ one halo row)."""
    H = x.shape[2]
    y0 = int(np.random.RandomState(seed).randint(1, H - rows - 1))
    ref = F.conv2d(x[:, :, y0 - 1:y0 + rows + 1], w, bias, padding=(0, 1))
    got = out[:, :, y0:y0 + rows]
    assert tuple(ref.shape) == tuple(got.shape)
    err = (got - ref).abs().max().item() / ref.abs().max().item()
    assert err <= TOL, (y0, err)


# (B, Cin, Cout, H, W): heads' 3x3 at training batch, a DLA level-3 block, the Hourglass' 256 -> 256 at full size
FULL = [(4, 64, 256, 256, 512), (1, 128, 128, 128, 256), (1, 256, 256, 256, 512)]


@pytest.mark.parametrize("shape", FULL, ids=["heads 64->256 x4", "128->128 @128x256", "hourglass 256->256"])
def test_full_size_forward_properties(shape):
    B, ci, co, H, W = shape
    x, w = _t("x%s" % (shape,), (B, ci, H, W)), _t("w%s" % (shape,), (co, ci, 3, 3), 0.05)
    bias = _t("b%s" % (shape,), (co,))
    out = _conv(x, w, bias=bias)
    assert torch.isfinite(out).all()                                   # every element written
    _band_check(x, w, out, bias, seed=1)
    _band_check(x, w, out, bias, seed=2)
    assert torch.equal(_conv(x, w, bias=bias), out)                    # fixed summation order: bit-identical reruns
    # linearity in x (no bias)
    x2 = _t("x2%s" % (shape,), (1, ci, H, W))
    a, b_, c = _conv(x[:1], w), _conv(x2, w), _conv(x[:1] + x2, w)
    assert (a + b_ - c).abs().max().item() <= TOL * c.abs().max().item()
    assert torch.equal(_conv(2.0 * x[:1], w), 2.0 * a)                 # power-of-two scaling commutes with the split
    # images of a batch are independent
    if B > 1:
        one = _conv(x[B - 1:].contiguous(), w, bias=bias)
        assert torch.equal(one[0], out[B - 1])
    # impulses at the corners and in the middle return the flipped kernel; far from them the output is exactly zero
    z = torch.zeros((1, ci, H, W), device=DEV)
    for (c, y, xx) in ((0, 0, 0), (ci - 1, H - 1, W - 1), (ci // 2, H // 2, W // 2 + 1)):
        z[0, c, y, xx] = 1.0
    oz = _conv(z, w)
    assert (oz[0, :, 8:H // 2 - 8, :] == 0).all() and (oz[0, :, :, 8:W // 2 - 8][:, 8:H // 2 - 8] == 0).all()
    np.testing.assert_allclose(oz[0, :, 0, 0].cpu().numpy(), w[:, 0, 1, 1].cpu().numpy(), rtol=0, atol=TOL * float(w.abs().max()))
    np.testing.assert_allclose(oz[0, :, H // 2 + 1, W // 2].cpu().numpy(), w[:, ci // 2, 0, 2].cpu().numpy(), rtol=0,
                               atol=TOL * float(w.abs().max()))
    np.testing.assert_allclose(oz[0, :, H - 2, W - 2].cpu().numpy(), w[:, ci - 1, 2, 2].cpu().numpy(), rtol=0,
                               atol=TOL * float(w.abs().max()))


def test_full_size_weight_gradient_adjoint():
    """<conv(x, w), go> == <w, wgrad(x, go)> at the heads' training shape (64 -> 256 @256x512 x4)."""
    L = _C.lib()
    B, ci, co, H, W = 4, 64, 256, 256, 512
    x, w, go = _t("ax", (B, ci, H, W)), _t("aw", (co, ci, 3, 3), 0.05), _t("ago", (B, co, H, W))
    y = _conv(x, w)
    assert L.cp_conv3x3_mfma_wgrad_supported(ci, co, H, W)
    gw = torch.zeros((co, ci, 3, 3), device=DEV)
    _C.check(L.cp_conv3x3_mfma_wgrad(P(x), P(go), P(gw), B, ci, H, W, co, _C.stream()), "wgrad")
    lhs = (y.double() * go.double()).sum().item()
    rhs = (w.double() * gw.double()).sum().item()
    norm = (y.double().abs() * go.double().abs()).sum().item()
    assert abs(lhs - rhs) <= 1e-5 * norm
    # and the gradient against the library on a channel slice (float atomics: tolerance, not equality)
    ref = torch.nn.grad.conv2d_weight(x, (8, ci, 3, 3), go[:, :8].contiguous(), padding=1)
    assert (gw[:8] - ref).abs().max().item() <= 1e-4 * ref.abs().max().item()


def test_full_size_fused_heads_kernel():
    """cp_heads_fused_forward at the bench's launch (64 -> 4 x 256 -> 8 / 32 / 1 / 2 @256x512): a 32-row band against
    the library's fp32 convolutions, bit-identical reruns, linearity of the pre-activation path is not testable
    through the ReLU -- instead: the heads are independent (each head alone gives the same map)."""
    L = _C.lib()
    couts, hc, cin, H, W = (8, 32, 1, 2), 256, 64, 256, 512
    nh = len(couts)
    x = _t("hf/x", (1, cin, H, W))
    w1, b1 = _t("hf/w1", (nh * hc, cin, 3, 3), 0.05), _t("hf/b1", (nh * hc,))
    w2 = [_t("hf/w2_%d" % i, (co, hc), 0.1) for i, co in enumerate(couts)]
    b2 = [_t("hf/b2_%d" % i, (co,)) for i, co in enumerate(couts)]
    vp = ctypes.c_void_p

    def run(sel):
        ws1 = torch.cat([w1[i * hc:(i + 1) * hc] for i in sel]).contiguous()
        bs1 = torch.cat([b1[i * hc:(i + 1) * hc] for i in sel]).contiguous()
        wp1 = torch.empty(L.cp_conv_mfma_weight_bytes(cin, len(sel) * hc, 9), dtype=torch.uint8, device=DEV)
        _C.check(L.cp_conv_mfma_prepare(P(ws1), cin, len(sel) * hc, 9, 0, P(wp1), _C.stream()), "prepare")
        w2p = []
        for i in sel:
            buf = torch.empty(L.cp_heads_fused_w2_bytes(hc), dtype=torch.uint8, device=DEV)
            _C.check(L.cp_heads_fused_prepare_w2(P(w2[i]), couts[i], hc, P(buf), _C.stream()), "prepare_w2")
            w2p.append(buf)
        outs = [torch.full((1, couts[i], H, W), float("nan"), device=DEV) for i in sel]
        rc = L.cp_heads_fused_forward(P(x), P(wp1), P(bs1), (vp * len(sel))(*[t.data_ptr() for t in w2p]),
                                      (vp * len(sel))(*[b2[i].data_ptr() for i in sel]),
                                      (vp * len(sel))(*[t.data_ptr() for t in outs]),
                                      (ctypes.c_int32 * len(sel))(*[couts[i] for i in sel]), len(sel), 1, cin, H, W, hc,
                                      _C.stream())
        assert rc == 0
        return outs

    outs = run(range(nh))
    again = run(range(nh))
    y0 = 101
    hid = F.relu(F.conv2d(x[:, :, y0 - 1:y0 + 33], w1, b1, padding=(0, 1)))
    for i, co in enumerate(couts):
        assert torch.isfinite(outs[i]).all() and torch.equal(outs[i], again[i])
        ref = F.conv2d(hid[:, i * hc:(i + 1) * hc], w2[i].view(co, hc, 1, 1), b2[i])
        got = outs[i][:, :, y0:y0 + 32]
        assert (got - ref).abs().max().item() <= TOL * ref.abs().max().item(), i
    alone = run([1])
    assert torch.equal(alone[0], outs[1])


def test_hourglass_full_size_properties():
    """BASELINE config 4 at its full size (Hourglass-104, 2 stacks, 24-vertex polar head, 1 x 3 x 1024 x 2048): the
    inference path (prepare_inference: folded BatchNorm, fused epilogues, fused heads, fused up-sample + add) equals
    the plain eval path, the outputs are finite and the decoded indices are inside the map."""
    from centerpoly_amd.models.decode import polydet_decode
    from centerpoly_amd.models.model import create_model
    heads = {"hm": 8, "poly": 48, "pseudo_depth": 1, "reg": 2}
    torch.manual_seed(3)
    m = create_model("hourglass", heads, 256).to(DEV).eval()
    with torch.no_grad():
        for mod in m.modules():                                        # non-trivial BatchNorm statistics
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.running_mean.normal_(0, 0.05)
                mod.running_var.uniform_(0.7, 1.3)
        x = torch.from_numpy(synth.normal("convfull/hg/x", (1, 3, 1024, 2048))).to(DEV)
        plain = m(x)
        m.prepare_inference()
        fast = m(x)
    assert len(plain) == len(fast) == 2
    for s in range(2):
        for h in heads:
            a, b = fast[s][h], plain[s][h]
            assert tuple(a.shape) == (1, heads[h], 256, 512) and torch.isfinite(a).all()
            assert (a - b).abs().max().item() <= 2e-3 * max(b.abs().max().item(), 1e-6), (s, h)
    out = fast[-1]
    dets, inds, clses = polydet_decode(out["hm"].sigmoid(), out["poly"], out["pseudo_depth"], reg=out["reg"], K=128,
                                       rep="polar", return_inds=True)
    assert tuple(dets.shape) == (1, 128, 2 * 24 + 7) and torch.isfinite(dets).all()
    assert int(inds.min()) >= 0 and int(inds.max()) < 256 * 512 and int(clses.min()) >= 0 and int(clses.max()) < 8


@pytest.mark.parametrize("shape", [(4, 32, 64, 512, 1024), (4, 16, 32, 1024, 2048)], ids=["level2 32<-64", "level1 16<-32"])
def test_full_size_stride_2_input_gradient(shape):
    """cp_conv3x3_s2_input_grad at the training launches: every element written, linear in grad_out, bit-identical rerun,
    the adjoint identity <grad_in, x> = <grad_out, conv_s2(x)> against the kernel's own stride-2 forward, and equality
    with the library's fp32 conv_transpose on a band of rows."""
    B, ci, co, H, W = shape
    L = _C.lib()
    Ho, Wo = H // 2, W // 2
    w, go = _t("s2w%s" % (shape,), (co, ci, 3, 3), 0.05), _t("s2go%s" % (shape,), (B, co, Ho, Wo))
    wp = torch.empty(L.cp_conv_mfma_weight_bytes(co, ci, 9), dtype=torch.uint8, device=DEV)
    _C.check(L.cp_conv_mfma_prepare(P(w), co, ci, 9, 6, P(wp), _C.stream()), "prepare")

    def igrad(g, res=None):
        out = torch.full((B, ci, H, W), float("nan"), device=DEV) if res is None else res
        _C.check(L.cp_conv3x3_s2_input_grad(P(g), P(wp), P(res), P(out), B, ci, H, W, co, _C.stream()), "igrad")
        return out
    g1 = igrad(go)
    assert torch.isfinite(g1).all() and torch.equal(g1, igrad(go))
    g2 = igrad(go * 0.5)
    assert (g2 - 0.5 * g1).abs().max().item() <= 1e-6 * g1.abs().max().item()
    acc = igrad(go, res=g1.clone())                              # accumulate onto itself in place: 2 x
    assert (acc - 2 * g1).abs().max().item() <= 1e-6 * g1.abs().max().item()
    # band: rows [2 y0, 2 y0 + 2 r) of grad_in depend on grad_out rows y0 - 1 .. y0 + r
    y0, r = 37, 16
    ref = F.conv_transpose2d(go[:, :, y0 - 1:y0 + r + 1], w, stride=2, padding=1, output_padding=1)
    got = g1[:, :, 2 * y0:2 * y0 + 2 * r]
    refb = ref[:, :, 2:2 + 2 * r]
    assert (got - refb).abs().max().item() <= TOL * refb.abs().max().item()
    if ci >= 24:                                                 # adjoint against the kernel's own stride-2 forward
        x = _t("s2x%s" % (shape,), (B, ci, H, W))
        wpf = torch.empty(L.cp_conv_mfma_weight_bytes(ci, co, 9), dtype=torch.uint8, device=DEV)
        _C.check(L.cp_conv_mfma_prepare(P(w), ci, co, 9, 0, P(wpf), _C.stream()), "prepare")
        y = torch.empty((B, co, Ho, Wo), device=DEV)
        ptrs, chans = (ctypes.c_void_p * 1)(x.data_ptr()), (ctypes.c_int32 * 1)(ci)
        _C.check(L.cp_conv_mfma_forward_strided(ptrs, chans, 1, P(wpf), None, None, P(y), B, H, W, co, 9, 2, 0, _C.stream()), "fwd")
        lhs, rhs = (g1.double() * x.double()).sum().item(), (go.double() * y.double()).sum().item()
        assert abs(lhs - rhs) <= 1e-4 * abs(rhs)


def test_full_size_masked_head_gradient():
    """cp_conv_mfma_input_grad_relu at the heads' training launch (4 x 256 channels @256x512 from an 8-channel head):
    masked exactly where y <= 0, equal to the library's gradient elsewhere on a band, bias gradient = the channel sums
    of what it wrote, bit-identical rerun of the map."""
    B, c1, c2, H, W = 4, 256, 8, 256, 512
    L = _C.lib()
    w2, go = _t("mhw", (c2, c1, 1, 1), 0.1), _t("mhgo", (B, c2, H, W))
    y = torch.relu(_t("mhy", (B, c1, H, W)))
    wp = torch.empty(L.cp_conv_mfma_weight_bytes(c2, c1, 1), dtype=torch.uint8, device=DEV)
    _C.check(L.cp_conv_mfma_prepare(P(w2), c2, c1, 1, 1, P(wp), _C.stream()), "prepare")
    ws = torch.empty(L.cp_conv_mfma_input_grad_relu_workspace_bytes(B, c1, H, W), dtype=torch.uint8, device=DEV)

    def run():
        g = torch.full((B, c1, H, W), float("nan"), device=DEV)
        gb = torch.zeros(c1, device=DEV)
        _C.check(L.cp_conv_mfma_input_grad_relu(P(go), P(wp), P(y), P(g), P(gb), B, c1, H, W, c2, 1, P(ws), ws.numel(),
                                                _C.stream()), "igrad_relu")
        return g, gb
    g, gb = run()
    assert torch.isfinite(g).all() and torch.equal(g, run()[0])
    assert (g[y <= 0] == 0).all()
    y0, r = 101, 32
    ref = F.conv_transpose2d(go[:, :, y0:y0 + r], w2) * (y[:, :, y0:y0 + r] > 0)
    assert (g[:, :, y0:y0 + r] - ref).abs().max().item() <= TOL * ref.abs().max().item()
    sums = g.double().sum(dim=(0, 2, 3))
    assert (gb.double() - sums).abs().max().item() <= 1e-5 * sums.abs().max().item()

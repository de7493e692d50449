"""DCNv2 FORWARD on the kernels bench.py actually times, at the sizes it times them (round-3 verdict, weak #1): the
LDS-region split-bf16 kernel (`contraction="auto"` picks it at 64->64 @256x512 and 128->128 @128x256) and the fused
module launch (`cp_dcn_v2_forward_fused`: conv_offset_mask inside the DCN kernel), both through the C ABI.

  properties   zero offsets = 0.5 * conv2d + b, linearity in x and in the weight, independent batch entries (bit-equal),
               bit-identical reruns, an integer offset = a shifted tap -- on the full map
  whole map    every output element against oracle/dcn.py (the oracle takes 2 s per full map), for a 0.5-px white-noise
               field (every sample inside the staged window), a 1.5-px one, a smooth field of ~3 px (about half the
               steps through the cold gathers) and a field whose offsets throw samples off the image
  module       the fused launch: its copied-out 27 channels against conv2d in float64, its output against the oracle's
               DCN on those offsets, on the whole map
  network      DLA-34 at 1 x 3 x 1024 x 2048 (BASELINE config 2, the headline workload): prepare_inference("auto")
               against the exact-f32 arithmetic <= 1e-3 of each head's max-norm; the oracle's decode of the DEVICE heads
               equals the device decode bit for bit

Reference semantics: the DCN constructed at src/lib/models/networks/pose_dla_dcn.py:354 (upstream DCNv2 as restated in
oracle/dcn.py), DLASeg.forward pose_dla_dcn.py:470-482, polydet_decode models/decode.py:512-670."""
import numpy as np
import pytest
import torch

from centerpoly_amd import _C, synth
from oracle import dcn as odcn

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 2e-5          # of the output's max-norm: split-bf16 x3 measures 3e-6, exact f32 3e-7

# (Cin, Cout, H, W): config 2's launches that run the region kernel -- the last four (deep, small maps) with their input
# channels split over grid z and the slices' partial sums reduced by a second launch (round 4)
SHAPES = [(64, 64, 256, 512), (128, 128, 128, 256), (128, 64, 128, 256), (256, 256, 64, 128), (256, 128, 64, 128),
          (512, 256, 32, 64), (256, 64, 64, 128)]
IDS = ["%d-%d@%dx%d" % s for s in SHAPES]


def g(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _fwd(x, om, w, b, contraction="auto", **kw):
    from centerpoly_amd.models.networks.DCNv2.dcn_v2 import dcn_v2_forward_raw
    return dcn_v2_forward_raw(x, om, w, b, contraction=contraction, **kw)


def _kernel_id(B, Cin, Cout, H, W):
    from centerpoly_amd.models.networks.DCNv2.dcn_v2 import auto_contraction
    s = _C.DcnShape(B, Cin, H, W, Cout, 3, 3, 1, 1, 1, 1)
    return _C.lib().cp_dcn_v2_forward_kernel(s, _C.DCN_CONTRACTION[auto_contraction(s)])


def _field(tag, B, H, W, kind):
    """27-channel offset / mask-logit tensor: 'w0.5' / 'w1.5' white noise of that many px, 'smooth3' a 9x9 box-filtered
    field of ~3 px (bench.py's `roofline_by_offsets` fields), 'far' 1.5-px noise with 2 % of the offsets thrown +-300 px."""
    om = synth.normal("fwdfull/%s/om" % tag, (B, 27, H, W))
    if kind.startswith("w"):
        om[:, :18] *= float(kind[1:])
    elif kind == "smooth3":
        k = torch.full((1, 1, 9, 9), 1.0 / 81.0)
        sm = torch.nn.functional.conv2d(torch.from_numpy(om[:, :18]).reshape(B * 18, 1, H, W), k, padding=4)
        om[:, :18] = (sm.reshape(B, 18, H, W) * 27.0).numpy()
    elif kind == "far":
        om[:, :18] *= 1.5
        far = synth.uniform("fwdfull/%s/far" % tag, (B, 18, H, W)) < 0.02
        om[:, :18] = np.where(far, np.sign(om[:, :18]) * 300.0, om[:, :18])
    else:
        raise ValueError(kind)
    return om


@pytest.mark.parametrize("shape", SHAPES[:2], ids=IDS[:2])
def test_region_kernel_full_size_properties(shape):
    Cin, Cout, H, W = shape
    assert _kernel_id(2, Cin, Cout, H, W) == 2, "bench shape no longer on the LDS-region kernel: update this test"
    x = g(synth.normal("fwdfull/p/x", (2, Cin, H, W)))
    w = g(synth.normal("fwdfull/p/w", (Cout, Cin, 3, 3), 0.0, 0.05))
    b = g(synth.normal("fwdfull/p/b", (Cout,)))
    zero = torch.zeros(2, 27, H, W, device=DEV)
    out0 = _fwd(x, zero, w, b)
    ref0 = 0.5 * torch.nn.functional.conv2d(x.double(), w.double(), None, padding=1) + b.double().view(1, -1, 1, 1)
    assert (out0.double() - ref0).abs().max().item() <= TOL * ref0.abs().max().item()
    om = g(_field("p", 2, H, W, "w1.5"))
    base = _fwd(x, om, w, b)
    assert torch.isfinite(base).all() and torch.equal(base, _fwd(x, om, w, b)), "rerun differs"
    nb = torch.zeros_like(b)
    scale = float(base.abs().max())
    # power-of-two scaling commutes with the bf16 split: bit-equal
    assert torch.equal(_fwd(2.0 * x, om, w, nb), 2.0 * _fwd(x, om, w, nb))
    x2 = g(synth.normal("fwdfull/p/x2", (2, Cin, H, W)))
    lin = _fwd(x + x2, om, w, nb) - _fwd(x, om, w, nb) - _fwd(x2, om, w, nb)
    assert float(lin.abs().max()) <= 3 * TOL * scale
    w2 = g(synth.normal("fwdfull/p/w2", (Cout, Cin, 3, 3), 0.0, 0.05))
    add = _fwd(x, om, w + w2, nb) - _fwd(x, om, w, nb) - _fwd(x, om, w2, nb)
    assert float(add.abs().max()) <= 3 * TOL * scale
    one = _fwd(x[1:2].contiguous(), om[1:2].contiguous(), w, b)
    assert torch.equal(one[0], base[1]), "batch entries are not independent"
    # an integer offset (+1 row, +2 columns on every tap) = the zero-offset result of the shifted image
    sh = torch.zeros(1, 27, H, W, device=DEV)
    sh[:, 0:18:2] = 1.0
    sh[:, 1:18:2] = 2.0
    xs = torch.zeros_like(x[:1])
    xs[:, :, :H - 1, :W - 2] = x[:1, :, 1:, 2:]
    a1 = _fwd(x[:1].contiguous(), sh, w, b)
    a2 = _fwd(xs, zero[:1].contiguous(), w, b)
    assert torch.equal(a1[:, :, 2:H - 2, 2:W - 4], a2[:, :, 2:H - 2, 2:W - 4])
    # against the exact-f32 gather kernel on the whole map
    exact = _fwd(x, om, w, b, contraction="f32")
    assert float((base - exact).abs().max()) <= TOL * float(exact.abs().max())


def _oracle(x, om, w, b):
    """oracle/dcn.py on the WHOLE map (2 s for 64 -> 64 @256x512 on 8 cores), image by image."""
    outs = []
    for i in range(x.shape[0]):
        o1, o2, m = torch.chunk(om[i:i + 1].cpu(), 3, dim=1)
        outs.append(odcn.dcn_v2_forward(x[i:i + 1].cpu(), torch.cat((o1, o2), 1), torch.sigmoid(m), w.cpu(),
                                        b.cpu() if b is not None else None))
    return torch.cat(outs, 0)


@pytest.mark.parametrize("shape", SHAPES, ids=IDS)
@pytest.mark.parametrize("kind", ["w0.5", "w1.5", "smooth3", "far"])
def test_region_kernel_whole_map_vs_oracle(shape, kind):
    """The full-size launch against the oracle on every output element: window path (0.5 px), mixed (1.5 px), the smooth
    ~3-px field whose taps diverge beyond the shared window (cold gathers), and samples thrown off the image (DCNv2's
    zero-outside rule at all four borders)."""
    Cin, Cout, H, W = shape
    assert _kernel_id(1, Cin, Cout, H, W) == 2
    x = g(synth.normal("fwdfull/b/x", (1, Cin, H, W)))
    w = g(synth.normal("fwdfull/b/w", (Cout, Cin, 3, 3), 0.0, 1.0 / np.sqrt(9 * Cin)))
    b = g(synth.normal("fwdfull/b/b", (Cout,)))
    om = g(_field("b" + kind, 1, H, W, kind))
    out = _fwd(x, om, w, b).cpu()
    ref = _oracle(x, om, w, b)
    err = (out - ref).abs().max().item() / ref.abs().max().item()
    assert err <= TOL, (kind, err)
    if kind == "w1.5":
        # the K-split launches reduce their slices inside the kernel (the slice that arrives last adds them in slice
        # order): three more runs give the same bits, with and without the folded-BN epilogue
        for _ in range(3):
            assert torch.equal(_fwd(x, om, w, b).cpu(), out), "rerun differs"
        sc, sh = g(synth.uniform("fwdfull/b/sc", (Cout,)) + 0.5), g(synth.normal("fwdfull/b/sh", (Cout,)))
        ep = _fwd(x, om, w, None, ep_scale=sc, ep_shift=sh, relu=True).cpu()
        want = torch.relu((ref - b.cpu().view(1, -1, 1, 1)) * sc.cpu().view(1, -1, 1, 1) + sh.cpu().view(1, -1, 1, 1))
        assert (ep - want).abs().max().item() <= TOL * max(want.abs().max().item(), 1.0)


@pytest.mark.parametrize("kind", ["model", "large"])
def test_fused_module_launch_full_size(kind):
    """cp_dcn_v2_forward_fused at the bench's dominant launch (64 -> 64 @256x512, the launch `roofline` reports): the 27
    channels it copies out against conv2d in float64, its output against the oracle's DCN on those very offsets (so the
    comparison does not amplify the ~1e-5 px the split-bf16 offset convolution may differ by) -- both on the whole map --,
    a folded-BN + ReLU epilogue, a bit-identical rerun, two images in one launch.  'model': offset weights at their
    fan-in scale (what the bench model draws); 'large': 4x that (|offsets| of several px: cold path)."""
    from centerpoly_amd.models.networks.DCNv2.dcn_v2 import dcn_v2_module_forward
    B, Cin, Cout, H, W = 2, 64, 64, 256, 512
    s = _C.DcnShape(B, Cin, H, W, Cout, 3, 3, 1, 1, 1, 1)
    assert _C.lib().cp_dcn_v2_forward_fused_supported(s)
    x = g(synth.normal("fwdfull/m/x", (B, Cin, H, W)))
    w = g(synth.normal("fwdfull/m/w", (Cout, Cin, 3, 3), 0.0, 1.0 / np.sqrt(9 * Cin)))
    gain = 1.0 if kind == "model" else 4.0
    wom = g(synth.normal("fwdfull/m/wom", (27, Cin, 3, 3), 0.0, gain / np.sqrt(9 * Cin)))
    bom = g(synth.normal("fwdfull/m/bom", (27,), 0.0, 0.3))
    sc, sh = g(synth.uniform("fwdfull/m/sc", (Cout,), 0.5, 1.5)), g(synth.normal("fwdfull/m/sh", (Cout,)))
    r = dcn_v2_module_forward(x, wom, bom, w, None, ep_scale=sc, ep_shift=sh, relu=True, want_om=True)
    assert r is not None
    out, om = r
    r2 = dcn_v2_module_forward(x, wom, bom, w, None, ep_scale=sc, ep_shift=sh, relu=True, want_om=True)
    assert torch.equal(out, r2[0]) and torch.equal(om, r2[1]) and torch.isfinite(out).all()
    r3 = dcn_v2_module_forward(x, wom, bom, w, None, ep_scale=sc, ep_shift=sh, relu=True, want_om=False)
    assert r3[1] is None and torch.equal(out, r3[0]), "the launch without the copy-out differs"
    om_ref = torch.nn.functional.conv2d(x.double(), wom.double(), bom.double(), padding=1)
    assert (om.double() - om_ref).abs().max().item() <= TOL * om_ref.abs().max().item()
    ref = _oracle(x, om, w, None)
    ref = torch.relu(ref * sc.cpu().view(1, -1, 1, 1) + sh.cpu().view(1, -1, 1, 1))
    assert (out.cpu() - ref).abs().max().item() <= TOL * ref.abs().max().item(), kind


@pytest.mark.parametrize("size", [(1, 1024, 2048), (2, 352, 1216)], ids=["config 2: 1x1024x2048", "2x352x1216 (odd multiples of 32)"])
def test_dla34_full_size_inference_path_vs_exact_f32(size):
    """BASELINE config 2 at its full size (DLA-34 + DCNv2, 1 x 3 x 1024 x 2048): the inference path bench.py times
    (prepare_inference("auto"): folded BatchNorm, split-bf16 convolutions, region / fused DCN kernels, fused heads)
    against the same weights under the exact-f32 arithmetic on the plain eval path, <= 1e-3 of each head's max-norm
    (the north star's bar; measured ~1e-5).  The oracle's decode of the DEVICE heads equals the device decode bit for bit
    (indices, classes, records).  The second size (a KITTI-like map, two images) walks the edges of the same path: level
    maps of 88 x 304 down to 11 x 38 -- heights off the tile grids, a width that is not a multiple of 4 (the dword form of
    the convolution epilogue), the K-split forms of the small maps, the base pair on a ragged last tile row."""
    from centerpoly_amd import arithmetic
    from centerpoly_amd.models.decode import polydet_decode
    from centerpoly_amd.models.model import create_model
    from oracle import decode as odec
    heads = {"hm": 8, "poly": 32, "pseudo_depth": 1, "reg": 2}
    model = create_model("dla_34", heads, 256)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = {k: torch.from_numpy(v) for k, v in synth.fill_by_name(shapes).items()}
    B, H, W = size
    x = g(synth.normal("bench/input%s" % (size if size[0] != 1 else "",), (B, 3, H, W)))
    try:
        arithmetic.configure("exact_f32")
        plain = create_model("dla_34", heads, 256)
        plain.load_state_dict(sd)
        plain = plain.to(DEV).eval()
        with torch.no_grad():
            ref = {k: v.clone() for k, v in plain(x)[-1].items()}
        del plain
        arithmetic.configure("split_bf16")
        model.load_state_dict(sd)
        model = model.to(DEV).eval()
        model.prepare_inference(dcn_contraction="auto")
        with torch.no_grad():
            out = model(x)[-1]
            out = {k: v.clone() for k, v in out.items()}
            hm = out["hm"].sigmoid()
            dets, inds, clses = polydet_decode(hm, out["poly"], out["pseudo_depth"], reg=out["reg"], K=128,
                                               return_inds=True)
    finally:
        arithmetic.configure("split_bf16")
    for h in heads:
        assert tuple(out[h].shape) == (B, heads[h], H // 4, W // 4) and torch.isfinite(out[h]).all()
        err = (out[h] - ref[h]).abs().max().item() / ref[h].abs().max().item()
        assert err <= 1e-3, (h, err)
    err_hm = (hm - ref["hm"].sigmoid()).abs().max().item()
    assert err_hm <= 1e-3
    dref, iref, cref = odec.polydet_decode(hm.cpu(), out["poly"].cpu(), out["pseudo_depth"].cpu(), out["reg"].cpu(), K=128)
    assert torch.equal(inds.cpu(), iref) and torch.equal(clses.cpu(), cref)
    assert torch.equal(dets.cpu(), dref)

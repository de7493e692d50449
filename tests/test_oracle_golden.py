"""Pin the CPU oracle against golden vectors produced by the reference's own Python
(tests/golden/gen_golden.py).  CPU only."""
import json
import warnings

import numpy as np
import pytest
import torch

import cases
from oracle import decode as odec
from oracle import losses as olos
from oracle import nets as onet
from oracle import post as opost

T = torch.from_numpy


@pytest.mark.parametrize("case", cases.DECODE_CASES, ids=lambda c: c[0])
def test_decode_matches_reference(case, golden):
    name, B, C, h, w, N, K, rep = case
    g = golden("decode_" + name)
    heat, polys, depth, reg = (T(a) for a in cases.decode_inputs_np(*case))
    nm = odec.nms(heat)
    assert int((nm != 0).sum()) == int(g["nms_nnz"])
    assert float(nm.double().sum()) == float(g["nms_sum"])
    scores, inds, clses, ys, xs = odec.topk(nm, K)
    assert np.array_equal(inds.numpy(), g["inds"])            # bit-exact indices
    assert np.array_equal(clses.numpy(), g["clses"])
    assert np.array_equal(scores.numpy(), g["scores"])
    assert np.array_equal(ys.numpy(), g["ys"]) and np.array_equal(xs.numpy(), g["xs"])
    dets, inds2, _ = odec.polydet_decode(heat, polys, depth, reg, K=K, rep=rep)
    assert np.array_equal(inds2.numpy(), g["inds"])
    np.testing.assert_allclose(dets.numpy(), g["dets"], rtol=1e-6, atol=1e-5)
    dets_nr, _, _ = odec.polydet_decode(heat, polys, depth, None, K=K, rep=rep)
    np.testing.assert_allclose(dets_nr.numpy(), g["dets_noreg"], rtol=1e-6, atol=1e-5)
    if rep == "cartesian":                                      # no trig: bit-exact
        assert np.array_equal(dets.numpy(), g["dets"])


def test_topk_tie_rule_is_lowest_index_first():
    heat = torch.zeros(1, 2, 8, 8)
    heat[0, 0, 2, 3] = 0.5
    heat[0, 1, 5, 5] = 0.5
    heat[0, 1, 1, 1] = 0.7
    scores, inds, clses, ys, xs = odec.topk(odec.nms(heat), 6)
    assert clses[0, :3].tolist() == [1, 0, 1] and inds[0, :3].tolist() == [9, 19, 45]
    # zeros: class 0 first, lowest index first, skipping nothing
    assert clses[0, 3:].tolist() == [0, 0, 0] and inds[0, 3:].tolist() == [0, 1, 2]


def test_focal_matches_reference(golden):
    batch, out = cases.loss_batch("base", 2, 32, 48, 16, "cartesian")
    g = golden("loss_focal")
    x = T(out["hm"]).requires_grad_(True)
    act = olos.sigmoid_clamp(x)
    loss = olos.neg_loss(act, T(batch["hm"]))
    loss.backward()
    assert np.array_equal(act.detach().numpy(), g["act"])
    np.testing.assert_allclose(loss.item(), g["loss"], rtol=1e-6)
    np.testing.assert_allclose(x.grad.numpy(), g["grad"], rtol=1e-5, atol=1e-9)
    g0 = golden("loss_focal_nopos")
    x0 = T(out["hm"]).requires_grad_(True)
    l0 = olos.neg_loss(olos.sigmoid_clamp(x0), torch.zeros_like(x0))
    l0.backward()
    np.testing.assert_allclose(l0.item(), g0["loss"], rtol=1e-6)
    np.testing.assert_allclose(x0.grad.numpy(), g0["grad"], rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("key", ["reg", "pseudo_depth"])
def test_regl1_matches_reference(key, golden):
    batch, out = cases.loss_batch("base", 2, 32, 48, 16, "cartesian")
    g = golden("loss_regl1_" + key)
    o = T(out[key]).requires_grad_(True)
    l = olos.reg_l1_loss(o, T(batch["reg_mask"]), T(batch["ind"]), T(batch[key]))
    l.backward()
    np.testing.assert_allclose(l.item(), g["loss"], rtol=1e-6)
    np.testing.assert_allclose(o.grad.numpy(), g["grad"], rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize("key", ["reg", "pseudo_depth"])
def test_regsl1_and_mse_match_reference(key, golden):
    """`--reg_loss sl1` (RegLoss, losses.py:801-815) and `--mse_loss` (torch MSELoss on the raw head,
    trains/polydet.py:23) against values and gradients recorded from the reference's own classes."""
    batch, out = cases.loss_batch("base", 2, 32, 48, 16, "cartesian")
    g = golden("loss_regsl1_" + key)
    o = T(out[key] * 3.0).requires_grad_(True)
    l = olos.reg_sl1_loss(o, T(batch["reg_mask"]), T(batch["ind"]), T(batch[key]))
    l.backward()
    np.testing.assert_allclose(l.item(), g["loss"], rtol=1e-6)
    np.testing.assert_allclose(o.grad.numpy(), g["grad"], rtol=1e-6, atol=1e-9)
    gm = golden("loss_mse")
    x = T(out["hm"]).requires_grad_(True)
    stats = olos.polydet_loss([{k: (x if k == "hm" else T(v)) for k, v in out.items()}],
                              {k: T(v) for k, v in batch.items()}, mse_loss=True)[1]
    np.testing.assert_allclose(stats["hm_l"].item(), gm["loss"], rtol=1e-6)


@pytest.mark.parametrize("case", cases.POLY_CASES, ids=lambda c: c[0])
def test_polyloss_matches_reference(case, golden):
    name, B, h, w, N, rep, pl, order = case
    g = golden("loss_poly_" + name)
    batch, out = cases.loss_batch(name, B, h, w, N, rep)
    assert int(batch["reg_mask"].sum()) == int(g["nobj"])
    o = T(out["poly"]).requires_grad_(True)
    r = olos.poly_loss(o, T(batch["reg_mask"]), T(batch["ind"]), T(batch["poly"]), pl, rep, order)
    if order:
        np.testing.assert_allclose(r[1].item(), g["order"], rtol=1e-5, atol=1e-7)
        total, main = r[0] + r[1], r[0]
    else:
        total = main = r
    np.testing.assert_allclose(main.item(), g["loss"], rtol=1e-5, atol=1e-6)
    total.backward()
    idx = T(batch["ind"])
    rows = torch.gather(o.grad.view(B, 2 * N, -1), 2, idx.unsqueeze(1).expand(B, 2 * N, idx.shape[1]))
    scale = np.abs(g["grad_rows"]).max()
    np.testing.assert_allclose(rows.numpy(), g["grad_rows"], rtol=1e-3, atol=1e-5 * scale)
    np.testing.assert_allclose(o.grad.abs().double().sum().item(), g["grad_abs_sum"], rtol=1e-4)


def test_weiler_atherton_known_answers(golden):
    g = golden("wa_kats")
    names = sorted({k.rsplit("_subject", 1)[0] for k in g.files if k.endswith("_subject")
                    and not k.endswith("_area_subject")})
    assert len(names) == 6
    for n in names:
        s, c = T(g[n + "_subject"]), T(g[n + "_clip"])
        poly = olos.wa_clip(s, c)
        assert poly.shape[0] == g[n + "_poly"].shape[0], n
        if poly.shape[0]:
            np.testing.assert_allclose(poly.numpy(), g[n + "_poly"], rtol=1e-6, atol=1e-6)
        a = olos.area(poly)
        np.testing.assert_allclose(a.item(), g[n + "_area"], rtol=1e-6)
        np.testing.assert_allclose(olos.area(s).item(), g[n + "_area_subject"], rtol=1e-6)
        inter = float(a.item() == 0.0) * torch.min(olos.area(s), olos.area(c)) + a
        iou = inter / (olos.area(c) + olos.area(s) - inter + 1e-6)
        np.testing.assert_allclose(iou.item(), g[n + "_iou"], rtol=1e-6)
    # the survey's probed constants
    np.testing.assert_allclose(g["same16_area"], 325.2809, rtol=1e-6)
    np.testing.assert_allclose(g["rot015_area"], 312.8208, rtol=1e-6)
    np.testing.assert_allclose(g["inside_iou"], 0.25, rtol=1e-5)


def test_area_double_counts_first_term():
    # 8x8 square read as polar is not the probe; use the literal formula on a known polygon
    sq = torch.tensor([[1.0, 0.0], [1.0, np.pi / 2], [1.0, np.pi], [1.0, 3 * np.pi / 2]])
    # true area 2, literal formula adds the k=0 cross term (x0*y1 - y0*x1)/2 = 0.5 once more
    np.testing.assert_allclose(olos.area(sq).item(), 2.5, rtol=1e-6)


def test_traversal_defined_where_reference_crashes():
    assert olos.wa_traverse(4, 4, [[0, 0, 0]], []) == []          # IndexError case -> empty
    assert olos.wa_traverse(4, 4, [], [[0, 0, 0]]) == []


def _sd(g):
    shapes = {k: tuple(v) for k, v in json.loads(str(g["shapes"])).items()}
    return {k: T(v) for k, v in cases.fill_weights(shapes).items()}, shapes


@pytest.mark.parametrize("ns", [1, 2])
def test_hourglass_matches_reference(ns, golden):
    g = golden("net_hourglass%d" % ns)
    sd, shapes = _sd(g)
    with torch.no_grad():
        outs = onet.hourglass_forward(sd, T(cases.net_input("hourglass")), dict(cases.HEADS), ns)
    for s in range(ns):
        for h in dict(cases.HEADS):
            ref = g["s%d_%s" % (s, h)]
            np.testing.assert_allclose(outs[s][h].numpy(), ref, rtol=1e-3, atol=1e-4 * np.abs(ref).max())


def test_dla34_wiring_matches_reference(golden):
    g = golden("net_dla34")
    sd, shapes = _sd(g)
    assert len(shapes) == 402
    assert shapes["ida_up.proj_1.conv.conv_offset_mask.weight"] == (27, 128, 3, 3)
    with torch.no_grad():
        out = onet.dla_seg_forward(sd, T(cases.net_input("dla")), dict(cases.HEADS))[0]
    for h in dict(cases.HEADS):
        ref = g["s0_" + h]
        np.testing.assert_allclose(out[h].numpy(), ref, rtol=1e-3, atol=1e-4 * np.abs(ref).max())


def test_post_process_matches_reference(golden):
    g = golden("post_cart16")
    case = cases.DECODE_CASES[0]
    name, B, C, h, w, N, K, rep = case
    heat, polys, depth, reg = (T(a) for a in cases.decode_inputs_np(*case))
    dets, _, _ = odec.polydet_decode(heat, polys, depth, reg, K=K, rep=rep)
    ret = opost.polydet_post_process(dets.numpy()[:1].copy(), [cases.POST_META["c"]],
                                     [cases.POST_META["s"]], h, w, C)
    for j in range(1, C + 1):
        a = np.array(ret[0][j], dtype=np.float32).reshape(-1, 2 * N + 6)
        np.testing.assert_allclose(a, g["cls%d" % j], rtol=1e-6, atol=1e-4)


# ---- DCNv2: two independently structured derivations held against each other -------------------
@pytest.mark.parametrize("case", [(1, 3, 4, 6, 7, 0.7, 1), (2, 2, 3, 5, 5, 3.0, 1), (1, 2, 2, 4, 9, 8.0, 1),
                                  (1, 3, 2, 7, 6, 1.5, 2)],
                         ids=["small-offsets", "3px-offsets", "samples-leave-the-image", "dilation-2"])
def test_dcn_definition_vs_upstream_kernel_rules(case):
    """oracle/dcn.py (definition + torch autograd, float64 here) against oracle/dcn_im2col.py (the
    published im2col / col2im / col2im_coord kernels' own formulas, no autograd): forward and all
    five gradients.  Neither is pinned by the reference (DCNv2 is absent from it); this pins them to
    each other."""
    import torch
    from centerpoly_amd import synth
    from oracle import dcn as odcn
    from oracle import dcn_im2col as oim
    B, Cin, Cout, H, W, off_std, dil = case
    tag = "dcn2/%d%d%d%d%d" % (B, Cin, Cout, H, W)
    x = synth.normal(tag + "/x", (B, Cin, H, W)).astype(np.float64)
    off = synth.normal(tag + "/off", (B, 18, H, W)).astype(np.float64) * off_std + 0.013
    msk = 1.0 / (1.0 + np.exp(-synth.normal(tag + "/m", (B, 9, H, W)).astype(np.float64)))
    w = synth.normal(tag + "/w", (Cout, Cin, 3, 3)).astype(np.float64)
    b = synth.normal(tag + "/b", (Cout,)).astype(np.float64)
    go = synth.normal(tag + "/go", (B, Cout, H, W)).astype(np.float64)
    tx, toff, tm, tw, tb = (torch.from_numpy(v).requires_grad_(True) for v in (x, off, msk, w, b))
    y = odcn.dcn_v2_forward(tx, toff, tm, tw, tb, 1, dil, dil)
    y.backward(torch.from_numpy(go))
    y2 = oim.forward(x, off, msk, w, b, 1, dil, dil)
    np.testing.assert_allclose(y2, y.detach().numpy(), rtol=1e-10, atol=1e-10)
    gi, goff, gm, gw, gb = oim.backward(x, off, msk, w, go, 1, dil, dil)
    for name, got, want in (("input", gi, tx.grad), ("offset", goff, toff.grad), ("mask", gm, tm.grad),
                            ("weight", gw, tw.grad), ("bias", gb, tb.grad)):
        np.testing.assert_allclose(got, want.numpy(), rtol=1e-9, atol=1e-9, err_msg="grad_" + name)


def test_dcn_zero_offset_known_answer_im2col():
    """Known answer for the kernel-rule restatement: zero offsets and mask 0.5 give 0.5*conv2d + b."""
    import torch
    from centerpoly_amd import synth
    from oracle import dcn_im2col as oim
    x = synth.normal("dcn2/kat/x", (1, 3, 6, 5)).astype(np.float64)
    w = synth.normal("dcn2/kat/w", (4, 3, 3, 3)).astype(np.float64)
    b = synth.normal("dcn2/kat/b", (4,)).astype(np.float64)
    y = oim.forward(x, np.zeros((1, 18, 6, 5)), np.full((1, 9, 6, 5), 0.5), w, b)
    ref = 0.5 * torch.nn.functional.conv2d(torch.from_numpy(x), torch.from_numpy(w), None, padding=1) \
        + torch.from_numpy(b).view(1, -1, 1, 1)
    np.testing.assert_allclose(y, ref.numpy(), rtol=1e-12, atol=1e-12)

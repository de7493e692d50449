"""`--cat_spec_poly` / `--dense_poly` (reference: models/decode.py:534-537, trains/polydet.py:103-110,
datasets/sample/polydet.py:245-248,401-403,424-441 -- all default off, but inside the function bodies of the hot path).

Fixtures come from the reference's own Python (tests/golden/gen_catspec_dense_golden.py, gen_sampler_golden.py):
  * decode: per-class polygon gather on a map whose width is 2N (the only shape the reference's view accepts) and the
    RuntimeError it raises otherwise;
  * PolydetLoss with cat_spec_poly: the reference raises RuntimeError at its first object -- so does the mirror;
  * dense_poly: masked L1 over whole maps, value and gradient; the targets are in tests/test_targets.py."""
import types

import numpy as np
import pytest
import torch

import cases
from oracle import decode as odec
from oracle import losses as olos

T = torch.from_numpy
DEV = "cuda"


def test_oracle_catspec_decode_matches_reference(golden):
    name, B, C, h, w, N, K = cases.CATSPEC_DECODE
    gold = golden("decode_" + name)
    heat, polys, depth, reg = (T(a) for a in cases.catspec_decode_inputs_np(*cases.CATSPEC_DECODE))
    dets, inds, clses = odec.polydet_decode(heat, polys, depth, reg, K=K, rep="cartesian", cat_spec_poly=True)
    assert np.array_equal(dets.numpy(), gold["dets"])                       # cartesian: bit-exact
    # the class really selects the block: detection k's polygon = channels cls * 2N .. of its pixel, + centre
    b, k = 1, 5
    c, sp = int(clses[b, k]), int(inds[b, k])
    raw = polys[b, c * 2 * N:(c + 1) * 2 * N].reshape(2 * N, -1)[:, sp]
    xs = float(sp % w) + float(reg[b, 0].reshape(-1)[sp])
    assert abs(float(dets[b, k, 6]) - (float(raw[0]) + xs)) < 1e-5
    with pytest.raises(RuntimeError) as e:                                   # any width but 2N
        odec.polydet_decode(heat[..., :w - 8].contiguous(), polys[..., :w - 8].contiguous(), depth[..., :w - 8].contiguous(),
                            reg[..., :w - 8].contiguous(), K=K, cat_spec_poly=True)
    assert str(gold["error_type"]) == "RuntimeError" and str(e.value) == str(gold["error_text"])


def test_oracle_dense_and_catspec_losses_match_reference(golden):
    name, B, N, h, w = cases.DENSE_LOSS
    gold = golden("loss_" + name)
    pred, tgt, mask = cases.dense_loss_inputs_np(*cases.DENSE_LOSS)
    p = T(pred).requires_grad_(True)
    loss = olos.dense_poly_l1(p, T(tgt), T(mask))
    (loss * float(gold["grad_scale"])).backward()
    np.testing.assert_allclose(loss.item(), float(gold["loss"]), rtol=1e-6)
    assert np.array_equal(p.grad.numpy(), gold["grad"])
    # PolydetLoss with cat_spec_poly: the oracle raises what the reference raises
    gerr = golden("loss_catspec_error")
    batch, heads = cases.loss_batch("l1_cart16", 2, 32, 48, 16, "cartesian")
    tb = {k: T(v) for k, v in batch.items()}
    tb["cat_spec_mask"] = torch.zeros((2, tb["reg_mask"].shape[1], 8 * 32), dtype=torch.uint8)
    with pytest.raises(RuntimeError) as e:
        olos.polydet_loss([{k: T(v) for k, v in heads.items()}], tb, cat_spec_poly=True)
    assert str(e.value) == str(gerr["error_text"]) and str(gerr["error_type"]) == "RuntimeError"


@pytest.mark.gpu
def test_catspec_decode_kernel_vs_reference_golden(golden):
    from centerpoly_amd.models.decode import polydet_decode
    name, B, C, h, w, N, K = cases.CATSPEC_DECODE
    gold = golden("decode_" + name)
    heat, polys, depth, reg = (T(a).to(DEV) for a in cases.catspec_decode_inputs_np(*cases.CATSPEC_DECODE))
    dets, inds, clses = polydet_decode(heat, polys, depth, reg=reg, cat_spec_poly=True, K=K, return_inds=True)
    ref, rinds, rcls = odec.polydet_decode(heat.cpu(), polys.cpu(), depth.cpu(), reg.cpu(), K=K, cat_spec_poly=True)
    assert torch.equal(inds.cpu(), rinds) and torch.equal(clses.cpu(), rcls)
    assert np.array_equal(dets.cpu().numpy(), gold["dets"])                 # the reference's own records, bit for bit
    with pytest.raises(RuntimeError) as e:                                   # the reference's precondition: width == 2N
        polydet_decode(heat[..., :w - 8].contiguous(), polys[..., :w - 8].contiguous(), depth[..., :w - 8].contiguous(),
                       reg=reg[..., :w - 8].contiguous(), cat_spec_poly=True, K=K)
    assert str(e.value) == str(gold["error_text"])
    # through the C ABI the kernel itself has no such limit: any map, class picks the block (checked against a gather)
    from centerpoly_amd.models.decode import _decode_native
    h2, w2 = 20, 28
    heat2 = torch.sigmoid(T(np.ascontiguousarray(cases.synth.heat_logits("dec/cs2/hm", 1, 5, h2, w2)))).to(DEV)
    polys2 = T(cases.synth.normal("dec/cs2/poly", (1, 5 * 12, h2, w2))).to(DEV)
    z = torch.zeros((1, 2, h2, w2), device=DEV)
    d2, i2, c2 = _decode_native(heat2, polys2, z[:, :1].contiguous(), z, 16, "cartesian", cat_spec_poly=True)
    for k in range(16):
        c, sp = int(c2[0, k]), int(i2[0, k])
        raw = polys2[0, c * 12:(c + 1) * 12].reshape(12, -1)[:, sp].cpu()
        want = raw.clone()
        want[0::2] += float(sp % w2)
        want[1::2] += float(sp // w2)
        assert torch.equal(d2[0, k, 6:18].cpu(), want), k


@pytest.mark.gpu
def test_dense_poly_l1_kernel_vs_reference_golden(golden):
    from centerpoly_amd.models.losses import dense_poly_l1_loss
    name, B, N, h, w = cases.DENSE_LOSS
    gold = golden("loss_" + name)
    pred, tgt, mask = cases.dense_loss_inputs_np(*cases.DENSE_LOSS)
    p = T(pred).to(DEV).requires_grad_(True)
    loss = dense_poly_l1_loss(p, T(tgt).to(DEV), T(mask).to(DEV))
    (loss * float(gold["grad_scale"])).backward()
    np.testing.assert_allclose(loss.item(), float(gold["loss"]), rtol=1e-6)
    got, want = p.grad.cpu().numpy(), gold["grad"]
    assert np.array_equal(got != 0, want != 0)                               # masked-out and exact-hit elements: 0
    np.testing.assert_allclose(got, want, rtol=1e-6, atol=0)
    # empty mask: 0 / 1e-4 = 0, zero gradient
    p2 = T(pred).to(DEV).requires_grad_(True)
    l2 = dense_poly_l1_loss(p2, T(tgt).to(DEV), torch.zeros_like(p2))
    l2.backward()
    assert float(l2.detach()) == 0.0 and float(p2.grad.abs().max()) == 0.0


def _opt(**kw):
    base = dict(num_stacks=1, poly_loss="l1", rep="cartesian", poly_order=False, hm_weight=1.0, off_weight=1.0,
                poly_weight=1.0, depth_weight=0.1, reg_offset=True, reg_loss="l1", task="polydet", mse_loss=False,
                cat_spec_poly=False, dense_poly=False)
    base.update(kw)
    return types.SimpleNamespace(**base)


@pytest.mark.gpu
def test_polydet_loss_dense_and_catspec_branches(golden):
    """PolydetLoss.forward (trains/polydet.py:103-110): --dense_poly = the dense masked L1 in place of PolyLoss, every
    other term unchanged (equal to the oracle's polydet_loss); --cat_spec_poly raises the reference's RuntimeError."""
    from centerpoly_amd.trains.polydet import PolydetLoss
    batch, heads = cases.loss_batch("l1_cart16", 2, 32, 48, 16, "cartesian")
    _, tgt, mask = cases.dense_loss_inputs_np("dense16b", 2, 16, 32, 48)
    tb = {k: T(v) for k, v in batch.items()}
    tb["dense_poly"], tb["dense_poly_mask"] = T(tgt), T(mask)
    del tb["poly"]                                                           # the dense dict has no 'poly' (:441)
    ref, rstats = olos.polydet_loss([{k: T(v) for k, v in heads.items()}], tb, dense_poly=True)
    out = {k: T(v).to(DEV).requires_grad_(True) for k, v in heads.items()}
    heads_d = {k: v * 1.0 for k, v in out.items()}                          # non-leaf, like a conv output
    loss, stats = PolydetLoss(_opt(dense_poly=True))([heads_d], {k: v.to(DEV) for k, v in tb.items()})
    loss.backward()
    for k in rstats:
        np.testing.assert_allclose(float(stats[k]), float(rstats[k]), rtol=1e-5, err_msg=k)
    assert float(out["poly"].grad.abs().sum()) > 0
    gerr = golden("loss_catspec_error")
    tb2 = {k: T(v).to(DEV) for k, v in batch.items()}
    tb2["cat_spec_mask"] = torch.zeros((2, tb2["reg_mask"].shape[1], 8 * 32), dtype=torch.uint8, device=DEV)
    tb2["cat_spec_poly"] = torch.zeros((2, tb2["reg_mask"].shape[1], 8 * 32), device=DEV)
    with pytest.raises(RuntimeError) as e:
        PolydetLoss(_opt(cat_spec_poly=True))([{k: T(v).to(DEV) for k, v in heads.items()}], tb2)
    assert str(e.value) == str(gerr["error_text"])

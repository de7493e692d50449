#!/usr/bin/env python3
"""Fixtures of the `--cat_spec_poly` / `--dense_poly` branches, from the REFERENCE's own Python (build container only):

  decode_catspec16.npz   models/decode.py::polydet_decode(cat_spec_poly=True) on a map whose width equals 2N (the only
                         shape its `polys.view(batch, K, cat, nbr_points)` accepts: `nbr_points` is read off the map,
                         decode.py:514), plus the exception it raises on any other width
  loss_catspec_error.npz the exception models/losses.py::PolyLoss.forward raises when trains/polydet.py:103-106 hands it the
                         [B, M, C*2N] cat_spec_mask (`if mask[batch][i]:`, losses.py:870)
  loss_dense16.npz       trains/polydet.py:107-110: torch.nn.L1Loss(reduction='sum')(pred * mask, target * mask) /
                         (mask.sum() + 1e-4) -- value and gradient (trains/polydet.py itself needs numba / progress /
                         wandb, absent here: the three-line expression is evaluated with the reference's own operands)

Usage:  python tests/golden/gen_catspec_dense_golden.py"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference/src/lib")
sys.modules["cv2"] = types.ModuleType("cv2")
sys.modules["seaborn"] = types.ModuleType("seaborn")

from models.decode import polydet_decode            # noqa: E402  (the reference's own)
from models.losses import PolyLoss                  # noqa: E402

from cases import CATSPEC_DECODE, DENSE_LOSS, catspec_decode_inputs_np, dense_loss_inputs_np, loss_batch  # noqa: E402

T = torch.from_numpy


def main():
    name, B, C, h, w, N, K = CATSPEC_DECODE
    heat, polys, depth, reg = catspec_decode_inputs_np(name, B, C, h, w, N, K)
    dets = polydet_decode(T(heat), T(polys), T(depth), reg=T(reg), cat_spec_poly=True, K=K, rep="cartesian")
    out = {"dets": dets.numpy()}
    try:                                             # any other width: the view fails
        polydet_decode(T(heat[..., :w - 8].copy()), T(polys[..., :w - 8].copy()), T(depth[..., :w - 8].copy()),
                       reg=T(reg[..., :w - 8].copy()), cat_spec_poly=True, K=K, rep="cartesian")
        raise SystemExit("expected the reference to raise")
    except RuntimeError as e:
        out["error_type"] = np.array(type(e).__name__)
        out["error_text"] = np.array(str(e))
    np.savez_compressed(os.path.join(HERE, "decode_%s.npz" % name), **out)
    print("decode_%s: dets %s; other widths -> %s: %s" % (name, dets.shape, out["error_type"], out["error_text"]))

    # PolyLoss with the cat-spec mask (what PolydetLoss.forward passes when opt.cat_spec_poly)
    batch, heads = loss_batch("l1_cart16", 2, 32, 48, 16, "cartesian")
    M = batch["reg_mask"].shape[1]
    cs_mask = np.zeros((2, M, 8 * 32), np.uint8)
    cs_mask[:, :, :32] = batch["reg_mask"][:, :, None]
    cs_poly = np.zeros((2, M, 8 * 32), np.float32)
    opt = types.SimpleNamespace(poly_loss="l1", rep="cartesian", poly_order=False)
    try:
        PolyLoss(opt)(T(np.tile(heads["poly"], (1, 8, 1, 1))), T(cs_mask), T(batch["ind"]), T(cs_poly), hm=T(heads["hm"]))
        raise SystemExit("expected the reference to raise")
    except RuntimeError as e:
        np.savez_compressed(os.path.join(HERE, "loss_catspec_error.npz"), error_type=np.array(type(e).__name__),
                            error_text=np.array(str(e)))
        print("loss_catspec_error: %s: %s" % (type(e).__name__, e))

    name, B, N, h, w = DENSE_LOSS
    pred, tgt, mask = dense_loss_inputs_np(name, B, N, h, w)
    p = T(pred).requires_grad_(True)
    crit_dense_poly = torch.nn.L1Loss(reduction="sum")          # trains/polydet.py:31
    mask_weight = T(mask).sum() + 1e-4                          # :108
    loss = crit_dense_poly(p * T(mask), T(tgt) * T(mask)) / mask_weight
    (loss * 0.7).backward()
    np.savez_compressed(os.path.join(HERE, "loss_%s.npz" % name), loss=loss.detach().numpy(), grad=p.grad.numpy(),
                        grad_scale=np.float32(0.7), mask_sum=T(mask).sum().numpy())
    print("loss_%s: %.6f (mask sum %d)" % (name, float(loss), int(mask.sum())))


if __name__ == "__main__":
    main()

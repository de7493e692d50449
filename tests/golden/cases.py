"""Input builders shared by tests/golden/gen_golden.py (which runs the reference)
and the tests (which run the oracle and the HIP path on the same inputs).
Everything is regenerated from centerpoly_amd.synth by stream name."""
import numpy as np
import torch

from centerpoly_amd import synth

HEADS = (("hm", 8), ("poly", 32), ("pseudo_depth", 1), ("reg", 2))


def _sigmoid_f32(x):
    # torch's CPU fp32 sigmoid, so generator and tests feed bit-identical heat
    return torch.sigmoid(torch.from_numpy(x)).numpy()


DECODE_CASES = [
    # name, B, C, h, w, N, K, rep
    ("cart16", 2, 8, 64, 96, 16, 40, "cartesian"),
    ("cart32", 1, 8, 96, 320, 32, 128, "cartesian"),
    ("polar24", 2, 8, 64, 96, 24, 40, "polar"),
    ("polarfixed12", 1, 9, 48, 64, 12, 32, "polar_fixed"),
]


def decode_inputs_np(name, B, C, h, w, N, K, rep):
    heat = _sigmoid_f32(synth.heat_logits("dec/%s/hm" % name, B, C, h, w))
    if rep == "cartesian":
        polys = (synth.normal("dec/%s/poly" % name, (B, 2 * N, h, w), 0.0, 8.0))
    else:
        p = synth.uniform("dec/%s/poly" % name, (B, 2 * N, h, w), 1.0, 30.0)
        p[:, 1::2] = synth.uniform("dec/%s/ang" % name, (B, N, h, w), -1.0, 7.0)
        polys = p
    depth = (synth.uniform("dec/%s/depth" % name, (B, 1, h, w)))
    reg = (synth.uniform("dec/%s/reg" % name, (B, 2, h, w)))
    return heat, polys, depth, reg



def loss_batch(name, B, h, w, N, rep, mean_objs=7):
    """Shared by the generator and the tests: batch + raw head outputs where the
    poly head holds target + perturbation at the object centres."""
    batch = synth.train_batch(B, h, w, nbr_points=N, rep=rep, mean_objs=mean_objs,
                              stream="loss/" + name, with_input=False)
    out = {
        "hm": synth.heat_logits("loss/%s/hm" % name, B, 8, h, w),
        "reg": synth.uniform("loss/%s/reg" % name, (B, 2, h, w)),
        "pseudo_depth": synth.uniform("loss/%s/depth" % name, (B, 1, h, w)),
        "poly": synth.normal("loss/%s/poly" % name, (B, 2 * N, h, w), 0.0, 0.5),
    }
    noise = synth.normal("loss/%s/pert" % name, (B, batch["poly"].shape[1], 2 * N))
    for b in range(B):
        for k in range(batch["reg_mask"].shape[1]):
            if not batch["reg_mask"][b, k]:
                continue
            cy, cx = divmod(int(batch["ind"][b, k]), w)
            t = batch["poly"][b, k].copy()
            if rep == "cartesian":
                t += 2.0 * noise[b, k]
            else:
                t[0::2] += 1.5 * noise[b, k, 0::2]
                t[1::2] += 0.12 * noise[b, k, 1::2]
                if k % 3 == 0:          # order term: negative angles AFTER a positive one
                    t[1 + N::2] -= 6.0
                elif k % 3 == 1:        # order term: a swapped pair -> non-zero hinge
                    t[3], t[7] = t[7], t[3]
            out["poly"][b, :, cy, cx] = t
    return batch, out


POLY_CASES = [
    # name, B, h, w, N, rep, poly_loss, poly_order
    ("l1_cart16", 2, 32, 48, 16, "cartesian", "l1", False),
    ("l1iou_cart16", 2, 32, 48, 16, "cartesian", "l1+iou", False),
    ("iou_polar16", 2, 32, 48, 16, "polar", "iou", False),
    ("l1iou_polar24", 1, 32, 48, 24, "polar", "l1+iou", False),
    ("l1_polar16_order", 2, 32, 48, 16, "polar", "l1", True),
    ("l1_cart32_order", 1, 24, 80, 32, "cartesian", "l1", True),
    ("l1iou_polar16_order", 1, 32, 48, 16, "polar", "l1+iou", True),
    ("relu_cart16", 1, 32, 48, 16, "cartesian", "relu", False),
    ("l1_polarfixed16", 1, 32, 48, 16, "polar_fixed", "l1", False),
]




def fill_weights(shapes):
    """name -> float32 array for a state_dict described by name -> shape.
    conv_offset_mask weights are halved so learned offsets stay O(1) pixel."""
    w = synth.fill_by_name(shapes)
    for k in w:
        if "conv_offset_mask" in k:
            w[k] = (w[k] * 0.5).astype(w[k].dtype)
    return w


def net_input(kind):
    if kind == "hourglass":
        return synth.normal("net/input", (1, 3, 128, 128))
    return synth.normal("net/input_dla", (1, 3, 64, 96))


POST_META = dict(c=np.array([1024.0, 512.0], dtype=np.float32), s=2048.0)


# ---- round 4: --cat_spec_poly / --dense_poly (decode.py:534-537, trains/polydet.py:103-110) -------------------------
CATSPEC_DECODE = ("catspec16", 2, 8, 24, 32, 16, 32)        # name, B, C, h, w (= 2N: the reference's view needs it), N, K


def catspec_decode_inputs_np(name, B, C, h, w, N, K):
    heat = _sigmoid_f32(synth.heat_logits("dec/%s/hm" % name, B, C, h, w))
    polys = synth.normal("dec/%s/poly" % name, (B, C * 2 * N, h, w), 0.0, 8.0)
    depth = synth.uniform("dec/%s/depth" % name, (B, 1, h, w))
    reg = synth.uniform("dec/%s/reg" % name, (B, 2, h, w))
    return heat, polys, depth, reg


DENSE_LOSS = ("dense16", 2, 16, 24, 40)                       # name, B, N, h, w


def dense_loss_inputs_np(name, B, N, h, w):
    """pred / dense_poly / dense_poly_mask [B, 2N, h, w]: blobs of constant polygon rows like draw_dense_reg leaves,
    a mask that is 1 exactly where the target is non-zero, and a few pred == target elements (sign(0) = 0)."""
    pred = synth.normal("dloss/%s/pred" % name, (B, 2 * N, h, w), 0.0, 3.0)
    tgt = np.zeros((B, 2 * N, h, w), np.float32)
    rows = synth.normal("dloss/%s/rows" % name, (B, 6, 2 * N), 0.0, 6.0)
    cen = synth.integers("dloss/%s/cen" % name, (B, 6, 3), 2, 20)
    for b in range(B):
        for k in range(6):
            cy, cx, r = int(cen[b, k, 0]), int(cen[b, k, 1]) * 2 % w, 1 + int(cen[b, k, 2]) % 4
            tgt[b, :, max(cy - r, 0):cy + r + 1, max(cx - r, 0):cx + r + 1] = rows[b, k][:, None, None]
    tgt[:, 3] = 0.0                                       # a channel whose target is 0 everywhere: masked out
    mask = (tgt != 0).astype(np.float32)
    pred[0, 0, :4, :4] = tgt[0, 0, :4, :4]                # exact hits
    return pred, tgt, mask

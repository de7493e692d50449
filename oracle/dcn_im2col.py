"""Oracle, second derivation: DCNv2 forward AND backward restated from the published im2col /
col2im kernels of CharlesShang/DCNv2 (`modulated_deformable_im2col_cuda`,
`modulated_deformable_col2im_cuda`, `modulated_deformable_col2im_coord_cuda` and their helpers
`dmcn_im2col_bilinear`, `dmcn_get_gradient_weight`, `dmcn_get_coordinate_weight`).  TEST
INFRASTRUCTURE -- numpy float64, loops over (tap, channel), vectorised over pixels only.

Why it exists: the algorithm is a third-party dependency that is absent from /root/reference (the
reference only does `from .DCNv2.dcn_v2 import DCN`, src/lib/models/networks/pose_dla_dcn.py:16,354)
and nothing the reference holds pins it.  oracle/dcn.py states the DEFINITION (zero-padded bilinear
sampling + torch autograd for the gradients); this file follows the upstream KERNELS' own rules and
hand-written gradient formulas -- no autograd anywhere -- so that tests/test_oracle_golden.py can hold
two independently structured derivations against each other (forward and all five gradients).

Upstream rules restated here:
  im2col          h_im = h_in + i*dil + offset_h; the sample counts only if h_im > -1, w_im > -1,
                  h_im < H, w_im < W; column = bilinear(h_im, w_im) * mask
  bilinear        corners (h_low, w_low) .. (h_high, w_high) = floor, floor + 1; a corner contributes
                  only if h_low >= 0 / w_low >= 0 / h_high <= H-1 / w_high <= W-1
  col2im          grad_im[y][x] += weight(h_im, w_im, y, x) * grad_col * mask for the integer (y, x)
                  with |h_im - y| < 1 and |w_im - x| < 1, weight = (1 - |h_im - y|) * (1 - |w_im - x|)
                  by cases on floor / floor + 1 (dmcn_get_gradient_weight)
  col2im_coord    grad_offset = sum_c grad_col * mask * d(bilinear)/d(h or w)
                  (dmcn_get_coordinate_weight: -+ the opposite axis' weights times the corner values);
                  grad_mask = sum_c grad_col * bilinear
Offset channel layout: 2k = dy of tap k, 2k+1 = dx of tap k (k = i*kw + j), one deformable group.
Parity status: unpinned by the reference, as oracle/dcn.py.
"""
import numpy as np


def _bilinear(im, h, w):
    """dmcn_im2col_bilinear on one channel plane im[H,W] at float coordinates h, w (arrays)."""
    H, W = im.shape
    h_low = np.floor(h).astype(np.int64)
    w_low = np.floor(w).astype(np.int64)
    h_high, w_high = h_low + 1, w_low + 1
    lh, lw = h - h_low, w - w_low
    hh, hw = 1 - lh, 1 - lw

    def at(yy, xx, ok):
        return np.where(ok, im[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)], 0.0)

    v1 = at(h_low, w_low, (h_low >= 0) & (w_low >= 0))
    v2 = at(h_low, w_high, (h_low >= 0) & (w_high <= W - 1))
    v3 = at(h_high, w_low, (h_high <= H - 1) & (w_low >= 0))
    v4 = at(h_high, w_high, (h_high <= H - 1) & (w_high <= W - 1))
    return hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4


def _coordinate_weight(im, h, w, bp_dir):
    """dmcn_get_coordinate_weight: derivative of the bilinear sample w.r.t. h (bp_dir 0) or w (1)."""
    H, W = im.shape
    outside = (h <= -1) | (h >= H) | (w <= -1) | (w >= W)
    h_low = np.floor(h).astype(np.int64)
    w_low = np.floor(w).astype(np.int64)
    h_high, w_high = h_low + 1, w_low + 1

    def at(yy, xx, ok):
        return np.where(ok, im[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)], 0.0)

    v_ll = at(h_low, w_low, (h_low >= 0) & (w_low >= 0))
    v_lh = at(h_low, w_high, (h_low >= 0) & (w_high <= W - 1))
    v_hl = at(h_high, w_low, (h_high <= H - 1) & (w_low >= 0))
    v_hh = at(h_high, w_high, (h_high <= H - 1) & (w_high <= W - 1))
    if bp_dir == 0:
        wt = -(w_low + 1 - w) * v_ll - (w - w_low) * v_lh + (w_low + 1 - w) * v_hl + (w - w_low) * v_hh
    else:
        wt = -(h_low + 1 - h) * v_ll + (h_low + 1 - h) * v_lh - (h - h_low) * v_hl + (h - h_low) * v_hh
    return np.where(outside, 0.0, wt)


def _coords(B, H, W, Ho, Wo, offset, k, kw, stride, pad, dil):
    i, j = k // kw, k % kw
    h_in = (np.arange(Ho) * stride - pad)[None, :, None]
    w_in = (np.arange(Wo) * stride - pad)[None, None, :]
    h_im = h_in + i * dil + offset[:, 2 * k]
    w_im = w_in + j * dil + offset[:, 2 * k + 1]
    return h_im, w_im


def im2col(x, offset, mask, kh=3, kw=3, stride=1, pad=1, dil=1):
    """modulated_deformable_im2col: columns [B, Cin*kh*kw, Ho, Wo] (row = c*kh*kw + i*kw + j)."""
    x = np.asarray(x, np.float64)
    offset = np.asarray(offset, np.float64)
    mask = np.asarray(mask, np.float64)
    B, C, H, W = x.shape
    Ho = (H + 2 * pad - (dil * (kh - 1) + 1)) // stride + 1
    Wo = (W + 2 * pad - (dil * (kw - 1) + 1)) // stride + 1
    K = kh * kw
    col = np.zeros((B, C * K, Ho, Wo))
    for k in range(K):
        h_im, w_im = _coords(B, H, W, Ho, Wo, offset, k, kw, stride, pad, dil)
        inside = (h_im > -1) & (w_im > -1) & (h_im < H) & (w_im < W)
        for b in range(B):
            for c in range(C):
                val = np.where(inside[b], _bilinear(x[b, c], h_im[b], w_im[b]), 0.0)
                col[b, c * K + k] = val * mask[b, k]
    return col


def forward(x, offset, mask, weight, bias, stride=1, pad=1, dil=1):
    """dcn_v2_forward: out = weight[Cout, Cin*K] @ columns + bias."""
    Cout, Cin, kh, kw = weight.shape
    col = im2col(x, offset, mask, kh, kw, stride, pad, dil)
    B, _, Ho, Wo = col.shape
    out = np.einsum("ok,bkp->bop", np.asarray(weight, np.float64).reshape(Cout, -1), col.reshape(B, -1, Ho * Wo))
    if bias is not None:
        out = out + np.asarray(bias, np.float64)[None, :, None]
    return out.reshape(B, Cout, Ho, Wo)


def backward(x, offset, mask, weight, grad_out, stride=1, pad=1, dil=1):
    """dcn_v2_backward: (grad_input, grad_offset, grad_mask, grad_weight, grad_bias), every formula
    hand-written as upstream has it (grad columns = weight^T @ grad_out, then col2im_coord and col2im;
    grad_weight = grad_out @ columns^T)."""
    x = np.asarray(x, np.float64)
    offset = np.asarray(offset, np.float64)
    mask = np.asarray(mask, np.float64)
    weight = np.asarray(weight, np.float64)
    grad_out = np.asarray(grad_out, np.float64)
    B, C, H, W = x.shape
    Cout, _, kh, kw = weight.shape
    K = kh * kw
    _, _, Ho, Wo = grad_out.shape
    wm = weight.reshape(Cout, C * K)
    go = grad_out.reshape(B, Cout, Ho * Wo)
    gcol = np.einsum("ok,bop->bkp", wm, go).reshape(B, C * K, Ho, Wo)
    col = im2col(x, offset, mask, kh, kw, stride, pad, dil)
    grad_weight = np.einsum("bop,bkp->ok", go, col.reshape(B, C * K, Ho * Wo)).reshape(weight.shape)
    grad_bias = go.sum(axis=(0, 2))
    grad_input = np.zeros_like(x)
    grad_offset = np.zeros_like(offset)
    grad_mask = np.zeros_like(mask)
    for k in range(K):
        h_im, w_im = _coords(B, H, W, Ho, Wo, offset, k, kw, stride, pad, dil)
        outside = (h_im <= -1) | (w_im <= -1) | (h_im >= H) | (w_im >= W)
        for b in range(B):
            hb, wb = h_im[b], w_im[b]
            for c in range(C):
                g = gcol[b, c * K + k]
                # col2im_coord: mask gradient from the sampled value, offset gradients from the
                # coordinate weights (upstream moves an outside sample to (-2, -2): weight 0)
                grad_mask[b, k] += np.where(outside[b], 0.0, g * _bilinear(x[b, c], hb, wb))
                hq = np.where(outside[b], -2.0, hb)
                wq = np.where(outside[b], -2.0, wb)
                grad_offset[b, 2 * k] += _coordinate_weight(x[b, c], hq, wq, 0) * g * mask[b, k]
                grad_offset[b, 2 * k + 1] += _coordinate_weight(x[b, c], hq, wq, 1) * g * mask[b, k]
                # col2im: scatter to the integer neighbours within distance < 1 of the sample
                top = g * mask[b, k]
                cur_h = hb.astype(np.int64)           # C cast: truncation toward zero
                cur_w = wb.astype(np.int64)
                for dy in range(-2, 3):
                    for dx in range(-2, 3):
                        yy, xx = cur_h + dy, cur_w + dx
                        near = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W) & (np.abs(hb - yy) < 1) & \
                               (np.abs(wb - xx) < 1)
                        wt = _gradient_weight(hb, wb, yy, xx, H, W)
                        contrib = np.where(near, wt * top, 0.0)
                        np.add.at(grad_input[b, c], (np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)), contrib)
    return grad_input, grad_offset, grad_mask, grad_weight, grad_bias


def _gradient_weight(ah, aw, h, w, H, W):
    """dmcn_get_gradient_weight(argmax_h, argmax_w, h, w): bilinear weight of integer (h, w)."""
    outside = (ah <= -1) | (ah >= H) | (aw <= -1) | (aw >= W)
    h_low = np.floor(ah).astype(np.int64)
    w_low = np.floor(aw).astype(np.int64)
    h_high, w_high = h_low + 1, w_low + 1
    wt = np.zeros(ah.shape)
    wt = np.where((h == h_low) & (w == w_low), (h + 1 - ah) * (w + 1 - aw), wt)
    wt = np.where((h == h_low) & (w == w_high), (h + 1 - ah) * (aw + 1 - w), wt)
    wt = np.where((h == h_high) & (w == w_low), (ah + 1 - h) * (w + 1 - aw), wt)
    wt = np.where((h == h_high) & (w == w_high), (ah + 1 - h) * (aw + 1 - w), wt)
    return np.where(outside, 0.0, wt)

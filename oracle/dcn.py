"""Oracle: modulated deformable convolution (DCNv2), torch CPU fp32.  TEST INFRASTRUCTURE.

The algorithm lives in a third-party dependency that is NOT in /root/reference:
CharlesShang/DCNv2 (un-vendored, no pinned version; the reference only does
`from .DCNv2.dcn_v2 import DCN`, src/lib/models/networks/pose_dla_dcn.py:16, and
constructs `DCN(chi, cho, kernel_size=(3,3), stride=1, padding=1, dilation=1,
deformable_groups=1)` at pose_dla_dcn.py:354).  This file restates the published
DCNv2 definition (Zhu et al., "Deformable ConvNets v2"):

    y(p) = b + sum_k  W_k . ( m_k(p) * x(p + p_k + dp_k(p)) )

with x(.) sampled bilinearly and ZERO outside the image, m_k = sigmoid of the
mask channels, and the 27-channel `conv_offset_mask` output laid out as
[o1 (9) | o2 (9) | mask (9)] where offset = cat(o1, o2) is read as interleaved
(dy, dx) per tap: channel 2k = dy_k, 2k+1 = dx_k, tap k = ky*kw + kx.
Sampling rule at the image border (public DCNv2 semantics): a sample position
(h, w) contributes only if h > -1, w > -1, h < H, w < W; each of the four
neighbours contributes only if it lies inside the image.

Parity status: unpinned by the reference (no test / fixture there touches DCN).
"""
import torch
import torch.nn.functional as F


def _bilinear_columns(x, offset, mask, kh, kw, stride, pad, dil, dg):
    """Sampled+modulated columns: [B, Cin, kh*kw, Ho, Wo]."""
    B, C, H, W = x.shape
    Ho = (H + 2 * pad - (dil * (kh - 1) + 1)) // stride + 1
    Wo = (W + 2 * pad - (dil * (kw - 1) + 1)) // stride + 1
    K = kh * kw
    cpg = C // dg
    dev, dt = x.device, x.dtype
    ho = torch.arange(Ho, device=dev, dtype=dt).view(1, 1, Ho, 1) * stride - pad
    wo = torch.arange(Wo, device=dev, dtype=dt).view(1, 1, 1, Wo) * stride - pad
    cols = []
    xf = x.reshape(B, C, H * W)
    for g in range(dg):
        xg = xf[:, g * cpg:(g + 1) * cpg]
        per_tap = []
        for k in range(K):
            ky, kx = k // kw, k % kw
            oy = offset[:, g * 2 * K + 2 * k].unsqueeze(1)       # [B,1,Ho,Wo]
            ox = offset[:, g * 2 * K + 2 * k + 1].unsqueeze(1)
            m = mask[:, g * K + k].unsqueeze(1)
            py = ho + ky * dil + oy
            px = wo + kx * dil + ox
            inside = (py > -1) & (px > -1) & (py < H) & (px < W)
            y0 = torch.floor(py)
            x0 = torch.floor(px)
            ly, lx = py - y0, px - x0
            hy, hx = 1 - ly, 1 - lx
            y0i, x0i = y0.long(), x0.long()
            y1i, x1i = y0i + 1, x0i + 1

            def corner(yi, xi):
                ok = (yi >= 0) & (yi <= H - 1) & (xi >= 0) & (xi <= W - 1) & inside
                idx = (yi.clamp(0, H - 1) * W + xi.clamp(0, W - 1)).view(B, 1, Ho * Wo)
                v = torch.gather(xg, 2, idx.expand(B, cpg, Ho * Wo)).view(B, cpg, Ho, Wo)
                return v * ok.to(dt)

            val = (hy * hx) * corner(y0i, x0i) + (hy * lx) * corner(y0i, x1i) \
                + (ly * hx) * corner(y1i, x0i) + (ly * lx) * corner(y1i, x1i)
            per_tap.append(val * m)
        cols.append(torch.stack(per_tap, dim=2))                 # [B,cpg,K,Ho,Wo]
    return torch.cat(cols, dim=1), Ho, Wo


def dcn_v2_forward(x, offset, mask, weight, bias, stride=1, pad=1, dil=1, dg=1):
    """x[B,Cin,H,W], offset[B,2*dg*K,Ho,Wo] (dy,dx interleaved), mask[B,dg*K,Ho,Wo]
    (already sigmoid-ed), weight[Cout,Cin,kh,kw], bias[Cout] -> [B,Cout,Ho,Wo]."""
    Cout, Cin, kh, kw = weight.shape
    col, Ho, Wo = _bilinear_columns(x, offset, mask, kh, kw, stride, pad, dil, dg)
    B = x.shape[0]
    col = col.reshape(B, Cin * kh * kw, Ho * Wo)
    out = torch.matmul(weight.reshape(Cout, Cin * kh * kw), col)
    if bias is not None:
        out = out + bias.view(1, Cout, 1)
    return out.view(B, Cout, Ho, Wo)


def dcn_module_forward(x, weight, bias, om_weight, om_bias, stride=1, pad=1, dil=1, dg=1):
    """The `DCN` nn.Module's forward: offset/mask conv -> chunk -> sigmoid -> dcn_v2.
    Parameter names weight, bias, conv_offset_mask.{weight,bias} are fixed by the
    reference checkpoints."""
    kh, kw = weight.shape[2:]
    om = F.conv2d(x, om_weight, om_bias, stride=stride, padding=pad)
    o1, o2, m = torch.chunk(om, 3, dim=1)
    offset = torch.cat((o1, o2), dim=1)
    return dcn_v2_forward(x, offset, torch.sigmoid(m), weight, bias, stride, pad, dil, dg)

"""Oracle: functional forward of DLA-34(+DCN up-sampling) and Hourglass from a
state_dict, torch CPU fp32, inference (eval-mode BN).  TEST INFRASTRUCTURE.

Independent of the product's nn.Module classes: it walks the reference's
checkpoint key grammar with torch.nn.functional only.

Follows:
  * DLA base      src/lib/models/networks/pose_dla_dcn.py:225-293 (BasicBlock 32-60,
                  Root 148-166, Tree 169-222; dla34 levels/channels :310-313)
  * DLAUp/IDAUp   pose_dla_dcn.py:362-413, DeformConv :347-359
  * DLASeg        pose_dla_dcn.py:427-482
  * Hourglass     src/lib/models/networks/large_hourglass.py:24-81, 283-342, 345-484
"""
import torch
import torch.nn.functional as F

from .dcn import dcn_module_forward


def _bn(sd, p, x, eps=1e-5):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"],
                        sd[p + ".weight"], sd[p + ".bias"], False, 0.0, eps)


def _conv(sd, p, x, stride=1, pad=0):
    return F.conv2d(x, sd[p + ".weight"], sd.get(p + ".bias"), stride=stride, padding=pad)


# ------------------------------- DLA-34 --------------------------------------

def _basic_block(sd, p, x, stride, residual=None):
    if residual is None:
        residual = x
    out = F.relu(_bn(sd, p + ".bn1", _conv(sd, p + ".conv1", x, stride, 1)))
    out = _bn(sd, p + ".bn2", _conv(sd, p + ".conv2", out, 1, 1))
    return F.relu(out + residual)


def _root(sd, p, xs):
    x = _conv(sd, p + ".conv", torch.cat(xs, 1))
    return F.relu(_bn(sd, p + ".bn", x))          # residual_root=False for dla34


def _tree(sd, p, x, levels, stride, level_root, residual=None, children=None):
    children = [] if children is None else children
    bottom = F.max_pool2d(x, stride, stride) if stride > 1 else x
    if (p + ".project.0.weight") in sd:
        residual = _bn(sd, p + ".project.1", _conv(sd, p + ".project.0", bottom))
    else:
        residual = bottom
    if level_root:
        children.append(bottom)
    if levels == 1:
        x1 = _basic_block(sd, p + ".tree1", x, stride, residual)
        x2 = _basic_block(sd, p + ".tree2", x1, 1)
        return _root(sd, p + ".root", [x2, x1] + children)
    x1 = _tree(sd, p + ".tree1", x, levels - 1, stride, False, residual)
    children.append(x1)
    return _tree(sd, p + ".tree2", x1, levels - 1, 1, False, children=children)


def dla34_base(sd, x, p="base"):
    levels = [1, 1, 1, 2, 2, 1]
    x = F.relu(_bn(sd, p + ".base_layer.1", _conv(sd, p + ".base_layer.0", x, 1, 3)))
    ys = []
    x = F.relu(_bn(sd, p + ".level0.1", _conv(sd, p + ".level0.0", x, 1, 1)))
    ys.append(x)
    x = F.relu(_bn(sd, p + ".level1.1", _conv(sd, p + ".level1.0", x, 2, 1)))
    ys.append(x)
    for lv in range(2, 6):
        x = _tree(sd, "%s.level%d" % (p, lv), x, levels[lv], 2, lv >= 3)
        ys.append(x)
    return ys


def _deform_conv(sd, p, x):
    y = dcn_module_forward(x, sd[p + ".conv.weight"], sd[p + ".conv.bias"],
                           sd[p + ".conv.conv_offset_mask.weight"],
                           sd[p + ".conv.conv_offset_mask.bias"])
    return F.relu(_bn(sd, p + ".actf.0", y))


def _ida_up(sd, p, layers, startp, endp):
    for i in range(startp + 1, endp):
        k = i - startp
        w = sd["%s.up_%d.weight" % (p, k)]
        f = w.shape[2] // 2
        y = _deform_conv(sd, "%s.proj_%d" % (p, k), layers[i])
        y = F.conv_transpose2d(y, w, None, stride=f, padding=f // 2, groups=w.shape[0])
        layers[i] = _deform_conv(sd, "%s.node_%d" % (p, k), y + layers[i - 1])


def dla_seg_forward(sd, x, heads, down_ratio=4, last_level=5):
    """pose_dla_dcn.py:470-482 -> [dict]."""
    first = {2: 1, 4: 2, 8: 3, 16: 4}[down_ratio]
    layers = dla34_base(sd, x)
    out = [layers[-1]]
    for i in range(len(layers) - first - 1):
        _ida_up(sd, "dla_up.ida_%d" % i, layers, len(layers) - i - 2, len(layers))
        out.insert(0, layers[-1])
    y = [out[i].clone() for i in range(last_level - first)]
    _ida_up(sd, "ida_up", y, 0, len(y))
    z = {}
    for h in heads:
        t = F.relu(_conv(sd, h + ".0", y[-1], 1, 1))
        z[h] = _conv(sd, h + ".2", t)
    return [z]


# ------------------------------ Hourglass ------------------------------------

def _convolution(sd, p, x, k, stride=1, with_bn=True):
    y = _conv(sd, p + ".conv", x, stride, (k - 1) // 2)
    if with_bn:
        y = _bn(sd, p + ".bn", y)
    return F.relu(y)


def _residual(sd, p, x, stride=1):
    y = F.relu(_bn(sd, p + ".bn1", _conv(sd, p + ".conv1", x, stride, 1)))
    y = _bn(sd, p + ".bn2", _conv(sd, p + ".conv2", y, 1, 1))
    if (p + ".skip.0.weight") in sd:
        s = _bn(sd, p + ".skip.1", _conv(sd, p + ".skip.0", x, stride))
    else:
        s = x
    return F.relu(y + s)


def _seq_residual(sd, p, x, count, first_stride=1):
    for i in range(count):
        x = _residual(sd, "%s.%d" % (p, i), x, first_stride if i == 0 else 1)
    return x


def _kp_module(sd, p, x, n, modules):
    up1 = _seq_residual(sd, p + ".up1", x, modules[0])
    low1 = _seq_residual(sd, p + ".low1", x, modules[0], 2)      # stride-2 conv replaces pooling
    if n > 1:
        low2 = _kp_module(sd, p + ".low2", low1, n - 1, modules[1:])
    else:
        low2 = _seq_residual(sd, p + ".low2", low1, modules[1])
    low3 = _seq_residual(sd, p + ".low3", low2, modules[0])
    up2 = F.interpolate(low3, scale_factor=2)                       # nearest
    return up1 + up2


def hourglass_forward(sd, x, heads, nstack):
    """large_hourglass.py:438-462 -> list of dicts, one per stack."""
    modules = [2, 2, 2, 2, 2, 4]
    inter = _convolution(sd, "pre.0", x, 7, 2)
    inter = _residual(sd, "pre.1", inter, 2)
    outs = []
    for s in range(nstack):
        kp = _kp_module(sd, "kps.%d" % s, inter, 5, modules)
        cnv = _convolution(sd, "cnvs.%d" % s, kp, 3)
        out = {}
        for h in heads:
            t = _convolution(sd, "%s.%d.0" % (h, s), cnv, 3, with_bn=False)
            out[h] = _conv(sd, "%s.%d.1" % (h, s), t)
        outs.append(out)
        if s < nstack - 1:
            a = _bn(sd, "inters_.%d.1" % s, _conv(sd, "inters_.%d.0" % s, inter))
            b = _bn(sd, "cnvs_.%d.1" % s, _conv(sd, "cnvs_.%d.0" % s, cnv))
            inter = _residual(sd, "inters.%d" % s, F.relu(a + b))
    return outs

"""Mirror of src/lib/models/utils.py (the parts polydet uses)."""
import torch

from .. import _C


def _sigmoid(x):
    """utils.py:8-10: in-place sigmoid, then clamp to [1e-4, 1-1e-4].
    (The training path fuses this into the focal kernel; this stand-alone form is
    kept for API parity and is plain elementwise torch on the device.)"""
    return torch.clamp(x.sigmoid_(), min=1e-4, max=1 - 1e-4)


def _gather_feat(feat, ind, mask=None):
    """utils.py:12-20: feat[B,HW,D], ind[B,M] -> [B,M,D]."""
    dim = feat.size(2)
    ind = ind.unsqueeze(2).expand(ind.size(0), ind.size(1), dim)
    feat = feat.gather(1, ind)
    if mask is not None:
        mask = mask.unsqueeze(2).expand_as(feat)
        feat = feat[mask].view(-1, dim)
    return feat


def _transpose_and_gather_feat(feat, ind):
    """utils.py:22-26 WITHOUT the full NHWC permute copy: gathers the M pixels
    straight out of the NCHW map (strided reads), same result [B,M,D]."""
    B, D = feat.shape[:2]
    idx = ind.unsqueeze(1).expand(B, D, ind.shape[1])
    return torch.gather(feat.reshape(B, D, -1), 2, idx).permute(0, 2, 1).contiguous()


def flip_tensor(x):
    return torch.flip(x, [3])

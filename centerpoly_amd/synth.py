"""Counter-based synthetic inputs for the polydet hot path.

Every value is a pure function of (seed, stream name, element index), so the CPU
container, the GPU box and the golden-vector generator regenerate identical
tensors without shipping data files and without depending on torch's RNG.

Shapes follow the batch schema of the reference's sampler
(src/lib/datasets/sample/polydet.py:425-449) and SURVEY.md section 8(d).
"""
import zlib

import numpy as np

SEED = 317  # src/lib/opts.py:43

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix(z):
    """splitmix64 finaliser, vectorised over a uint64 array."""
    z = (z + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def _stream_key(seed, stream):
    h = zlib.crc32(stream.encode("utf-8")) & 0xFFFFFFFF
    return np.uint64(((int(seed) & 0xFFFFFFFF) << 32) | h)


def bits(stream, n, seed=SEED, offset=0):
    """n uint64 words of stream `stream`."""
    with np.errstate(over="ignore"):
        ctr = np.arange(offset, offset + n, dtype=np.uint64)
        return _mix(_mix(ctr ^ _stream_key(seed, stream)) + ctr)


def uniform(stream, shape, lo=0.0, hi=1.0, seed=SEED, dtype=np.float32):
    n = int(np.prod(shape)) if len(shape) else 1
    u = (bits(stream, n, seed) >> np.uint64(40)).astype(np.float64) * (1.0 / (1 << 24))
    return (lo + (hi - lo) * u).astype(dtype).reshape(shape)


def normal(stream, shape, mean=0.0, std=1.0, seed=SEED, dtype=np.float32):
    n = int(np.prod(shape)) if len(shape) else 1
    u1 = ((bits(stream + "/a", n, seed) >> np.uint64(11)).astype(np.float64) + 1.0) * (1.0 / (1 << 53))
    u2 = (bits(stream + "/b", n, seed) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))
    z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
    return (mean + std * z).astype(dtype).reshape(shape)


def integers(stream, shape, lo, hi, seed=SEED):
    """Integers in [lo, hi)."""
    n = int(np.prod(shape)) if len(shape) else 1
    r = bits(stream, n, seed) % np.uint64(hi - lo)
    return (r.astype(np.int64) + lo).reshape(shape)


def fill_by_name(state_dict_shapes, seed=SEED, scale=None):
    """Deterministic weights keyed by parameter NAME (no weight files shipped).

    state_dict_shapes: mapping name -> shape.  Conv/linear weights get a
    fan-in-scaled normal, BN weights ~1, BN running_var in [0.5, 1.5], biases small.
    Returns name -> float32/int64 numpy array.
    """
    out = {}
    for name, shape in state_dict_shapes.items():
        shape = tuple(shape)
        if name.endswith("num_batches_tracked"):
            out[name] = np.zeros(shape, dtype=np.int64)
        elif name.endswith("running_var"):
            out[name] = uniform(name, shape, 0.5, 1.5, seed)
        elif name.endswith("running_mean"):
            out[name] = normal(name, shape, 0.0, 0.1, seed)
        elif len(shape) == 1 and name.endswith("weight"):  # BN gamma
            out[name] = uniform(name, shape, 0.8, 1.2, seed)
        elif len(shape) == 1:  # biases
            out[name] = normal(name, shape, 0.0, 0.05, seed)
        else:
            fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
            s = (1.0 / np.sqrt(fan_in)) if scale is None else scale
            out[name] = normal(name, shape, 0.0, s, seed)
    return out


def smooth_field(stream, shape, seed=SEED):
    """White noise + one 3x3 box blur: many distinct local maxima, fp32-distinct values."""
    x = normal(stream, shape, seed=seed, dtype=np.float64)
    p = np.pad(x, [(0, 0)] * (x.ndim - 2) + [(1, 1), (1, 1)], mode="edge")
    h, w = shape[-2:]
    acc = np.zeros_like(x)
    for dy in range(3):
        for dx in range(3):
            acc += p[..., dy:dy + h, dx:dx + w]
    return (acc / 3.0).astype(np.float32)


def heat_logits(stream, B, C, h, w, seed=SEED):
    """Head logits whose sigmoid has >= K distinct positive peaks per class."""
    return (smooth_field(stream, (B, C, h, w), seed) * 1.5 - 2.19).astype(np.float32)


def _gaussian_splat(hm, cx, cy, radius):
    """max-composited Gaussian with exact 1.0 at the integer centre
    (same construction as src/lib/utils/image.py:126-141, restated)."""
    d = 2 * radius + 1
    sigma = d / 6.0
    ax = np.arange(-radius, radius + 1, dtype=np.float64)
    g = np.exp(-(ax[None, :] ** 2 + ax[:, None] ** 2) / (2 * sigma * sigma))
    g[g < np.finfo(np.float64).eps * g.max()] = 0
    g = g.astype(np.float32)
    H, W = hm.shape
    l, r = min(cx, radius), min(W - cx, radius + 1)
    t, b = min(cy, radius), min(H - cy, radius + 1)
    view = hm[cy - t:cy + b, cx - l:cx + r]
    np.maximum(view, g[radius - t:radius + b, radius - l:radius + r], out=view)


def train_batch(B, h, w, nbr_points=16, num_classes=8, max_objs=128, rep="cartesian",
                mean_objs=20, seed=SEED, stream="batch", in_h=None, in_w=None,
                with_input=True):
    """Training batch dict (numpy) with the reference sampler's schema.

    Keys: input, hm, reg_mask, ind, reg, pseudo_depth, poly, freq_mask, peak.
    """
    N = nbr_points
    in_h = in_h or 4 * h
    in_w = in_w or 4 * w
    out = {}
    if with_input:
        out["input"] = normal(stream + "/input", (B, 3, in_h, in_w), seed=seed)
    hm = np.zeros((B, num_classes, h, w), dtype=np.float32)
    reg_mask = np.zeros((B, max_objs), dtype=np.uint8)
    ind = np.zeros((B, max_objs), dtype=np.int64)
    reg = np.zeros((B, max_objs, 2), dtype=np.float32)
    depth = np.zeros((B, max_objs, 1), dtype=np.float32)
    poly = np.zeros((B, max_objs, 2 * N), dtype=np.float32)
    peak = np.zeros((B, max_objs, 2), dtype=np.float32)
    # object counts ~ clip(Poisson(mean), 1, 66) via inverse-CDF on one uniform
    u = uniform(stream + "/nobj", (B,), seed=seed, dtype=np.float64)
    ks = np.arange(0, 200)
    logp = -mean_objs + ks * np.log(mean_objs) - np.cumsum(np.log(np.maximum(ks, 1)))
    cdf = np.cumsum(np.exp(logp))
    nobj = np.clip(np.searchsorted(cdf, u), 1, min(66, max_objs))
    for b in range(B):
        n = int(nobj[b])
        s = "%s/%d" % (stream, b)
        # distinct centres
        perm = np.argsort(bits(s + "/ctr", h * w, seed), kind="stable")[:n]
        cls = integers(s + "/cls", (n,), 0, num_classes, seed)
        rad = uniform(s + "/rad", (n, N), 2.0, 40.0, seed)
        phase = uniform(s + "/phase", (n,), 0.0, 2 * np.pi / N, seed)
        reg_mask[b, :n] = 1
        ind[b, :n] = perm
        reg[b, :n] = uniform(s + "/reg", (n, 2), seed=seed)
        depth[b, :n, 0] = uniform(s + "/depth", (n,), 0.0, 1.0, seed)
        # star-shaped polygon around the centre, angles increasing in [0, 2pi)
        theta = phase[:, None] + np.arange(N)[None, :] * (2 * np.pi / N)
        if rep == "cartesian":
            poly[b, :n, 0::2] = (rad * np.cos(theta)).astype(np.float32)
            poly[b, :n, 1::2] = (rad * np.sin(theta)).astype(np.float32)
        else:
            poly[b, :n, 0::2] = rad
            poly[b, :n, 1::2] = theta.astype(np.float32)
        for k in range(n):
            cy, cx = int(perm[k] // w), int(perm[k] % w)
            peak[b, k] = (cx, cy)
            _gaussian_splat(hm[b, cls[k]], cx, cy, int(2 + rad[k].mean() / 6))
    out.update(hm=hm, reg_mask=reg_mask, ind=ind, reg=reg, pseudo_depth=depth, poly=poly,
               freq_mask=np.ones((B,), dtype=np.float32), peak=peak)
    return out


def raw_annotations(stream, in_h, in_w, nbr_points=16, num_classes=8, n_objs=None, seed=SEED):
    """One image's RAW polydet annotations in image coordinates -- what the reference's sampler
    reads from the COCO json (src/lib/datasets/sample/polydet.py:160-170) before it builds the
    targets: a list of {bbox: [x, y, w, h], poly: [2N], cls_id, pseudo_depth, freq}.  Polygons
    are star shaped around a centre, vertices ordered by increasing angle starting near the
    top-left like the GT files; some objects hang over the image border and a few are tiny."""
    N = nbr_points
    if n_objs is None:
        n_objs = int(integers(stream + "/n", (1,), 1, 41, seed)[0])
    ctr = uniform(stream + "/ctr", (n_objs, 2), 0.0, 1.0, seed, dtype=np.float64)
    rad = uniform(stream + "/rad", (n_objs, N), 6.0, 180.0, seed, dtype=np.float64)
    size = uniform(stream + "/size", (n_objs,), 0.02, 1.0, seed, dtype=np.float64)
    cls = integers(stream + "/cls", (n_objs,), 0, num_classes, seed)
    depth = uniform(stream + "/depth", (n_objs,), 0.0, 1.0, seed)
    freq = uniform(stream + "/freq", (num_classes,), 0.05, 1.0, seed)
    anns = []
    for k in range(n_objs):
        cx, cy = ctr[k, 0] * in_w, ctr[k, 1] * in_h
        theta = -0.75 * np.pi + np.arange(N) * (2 * np.pi / N)
        xs = np.round(cx + size[k] * rad[k] * np.cos(theta), 2)
        ys = np.round(cy + size[k] * rad[k] * np.sin(theta), 2)
        pts = np.stack([xs, ys], axis=1).reshape(-1)
        x0, y0, x1, y1 = xs.min(), ys.min(), xs.max(), ys.max()
        anns.append({"bbox": [float(x0), float(y0), float(x1 - x0), float(y1 - y0)],
                     "poly": [float(v) for v in pts], "cls_id": int(cls[k]),
                     "pseudo_depth": float(depth[k]), "freq": float(freq[cls[k]])})
    return anns

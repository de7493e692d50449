"""Oracle: host tail of the detector (numpy).  TEST INFRASTRUCTURE.

Follows src/lib/utils/post_process.py:105-122 (polydet_post_process),
src/lib/utils/image.py:19-66 (transform_preds / get_affine_transform /
affine_transform) and src/lib/detectors/polydet.py:45-76.
`cv2.getAffineTransform` (image.py:56,58) is the exact affine through three
point pairs; restated as a 6x6 linear solve in float64.

soft_nms follows src/lib/external/nms.pyx:77-170 (Cython).  PINNED since round 4: oracle/build_ref_nms.py compiles
the reference's file into oracle/_ref/ (two tokens of the unrelated `nms()` patched in memory for numpy 2, `soft_nms`
itself unmodified), tests/golden/gen_softnms_golden.py records its outputs, tests/test_detector_io.py holds this
restatement and the C ABI to them bit for bit.  (The pin corrected the round-1 restatement: Cython turns the int
literal in `x2 - x1 + 1` into the double constant 1.0, so those sums run in double -- 1-ulp differences in the scores.)
"""
import numpy as np


def _third_point(a, b):
    d = a - b
    return b + np.array([-d[1], d[0]], dtype=np.float32)


def affine_from_3pts(src, dst):
    A = np.zeros((6, 6), dtype=np.float64)
    rhs = np.zeros(6, dtype=np.float64)
    for k in range(3):
        A[2 * k, 0:3] = (src[k, 0], src[k, 1], 1.0)
        A[2 * k + 1, 3:6] = (src[k, 0], src[k, 1], 1.0)
        rhs[2 * k], rhs[2 * k + 1] = dst[k, 0], dst[k, 1]
    return np.linalg.solve(A, rhs).reshape(2, 3)


def get_affine_transform(center, scale, rot, output_size, inv=0):
    """image.py:27-60 with rot in degrees, shift = 0."""
    if not isinstance(scale, (np.ndarray, list)):
        scale = np.array([scale, scale], dtype=np.float32)
    src_w = scale[0]
    dst_w, dst_h = output_size[0], output_size[1]
    rot_rad = np.pi * rot / 180
    sn, cs = np.sin(rot_rad), np.cos(rot_rad)
    sp = [0, src_w * -0.5]
    src_dir = np.array([sp[0] * cs - sp[1] * sn, sp[0] * sn + sp[1] * cs])
    dst_dir = np.array([0, dst_w * -0.5], np.float32)
    src = np.zeros((3, 2), dtype=np.float32)
    dst = np.zeros((3, 2), dtype=np.float32)
    src[0, :] = center
    src[1, :] = center + src_dir
    dst[0, :] = [dst_w * 0.5, dst_h * 0.5]
    dst[1, :] = np.array([dst_w * 0.5, dst_h * 0.5], np.float32) + dst_dir
    src[2, :] = _third_point(src[0, :], src[1, :])
    dst[2, :] = _third_point(dst[0, :], dst[1, :])
    if inv:
        return affine_from_3pts(np.float32(dst), np.float32(src))
    return affine_from_3pts(np.float32(src), np.float32(dst))


def transform_preds(coords, center, scale, output_size):
    """image.py:19-24: per-row fp32 [x,y,1] times the float64 2x3 matrix."""
    t = get_affine_transform(center, scale, 0, output_size, inv=1)
    pts = np.concatenate([coords[:, 0:2].astype(np.float32),
                          np.ones((coords.shape[0], 1), np.float32)], axis=1)
    out = np.zeros(coords.shape)
    out[:, 0:2] = pts @ t.T
    return out


def polydet_post_process(dets, c, s, h, w, num_classes):
    """post_process.py:105-122."""
    ret = []
    for i in range(dets.shape[0]):
        top = {}
        dets[i, :, :2] = transform_preds(dets[i, :, 0:2], c[i], s[i], (w, h))
        dets[i, :, 2:4] = transform_preds(dets[i, :, 2:4], c[i], s[i], (w, h))
        for j in range(6, dets.shape[-1] - 1, 2):
            dets[i, :, j:j + 2] = transform_preds(dets[i, :, j:j + 2], c[i], s[i], (w, h))
        classes = dets[i, :, 5]
        for j in range(num_classes):
            inds = classes == j
            top[j + 1] = np.concatenate([dets[i, inds, :4].astype(np.float32),
                                         dets[i, inds, 4:5].astype(np.float32),
                                         dets[i, inds, 6:].astype(np.float32)], axis=1).tolist()
        ret.append(top)
    return ret


def detector_post_process(dets, meta, scale, num_classes):
    """detectors/polydet.py:45-60."""
    dets = dets.reshape(1, -1, dets.shape[2])
    out = polydet_post_process(dets.copy(), [meta["c"]], [meta["s"]],
                               meta["out_height"], meta["out_width"], num_classes)
    width = dets.shape[2] - 1
    for j in range(1, num_classes + 1):
        a = np.array(out[0][j], dtype=np.float32).reshape(-1, width)
        a[:, :4] /= scale
        a[:, 5:-1] /= scale
        out[0][j] = a
    return out[0]


def soft_nms(boxes, sigma=0.5, Nt=0.3, threshold=0.001, method=0):
    """external/nms.pyx:77-170, IN PLACE on float32 [n, >= 5] rows (x1,y1,x2,y2,score,...).
    Literal behaviour: only columns 0-4 are swapped / overwritten (the polygon columns of a
    row stay where they were), rows are never removed from the array (N shrinks internally),
    and the caller in detectors/polydet.py:66-67 ignores the returned keep list.

    Arithmetic = what Cython makes of the `cdef float` text (pinned bit for bit by tests/golden/softnms_ref.npz, the
    reference's own build): an int literal next to a C float becomes the DOUBLE constant `1.0`, so `x2 - x1 + 1` is a
    float difference plus 1.0 in double; `area`, `iw`, `ih` round to float on assignment; `ua = float(...)` is a
    double expression rounded to float; `ov`, `-(ov*ov)/sigma` and `weight*score` are float operations; `1 - ov` is
    double rounded to float; `np.exp` runs in double on the float quotient."""
    f, d = np.float32, np.float64
    sigma, Nt, threshold = f(sigma), f(Nt), f(threshold)
    N0 = N = boxes.shape[0]
    for i in range(N0):                      # range(N) is evaluated once (i is a Python object)
        maxscore, maxpos = boxes[i, 4], i
        t = boxes[i, 0:5].copy()
        pos = i + 1
        while pos < N:
            if maxscore < boxes[pos, 4]:
                maxscore, maxpos = boxes[pos, 4], pos
            pos += 1
        boxes[i, 0:5] = boxes[maxpos, 0:5]
        boxes[maxpos, 0:5] = t
        tx1, ty1, tx2, ty2 = boxes[i, 0], boxes[i, 1], boxes[i, 2], boxes[i, 3]
        pos = i + 1
        while pos < N:
            x1, y1, x2, y2 = boxes[pos, 0], boxes[pos, 1], boxes[pos, 2], boxes[pos, 3]
            area = f((d(x2 - x1) + 1.0) * (d(y2 - y1) + 1.0))
            iw = f(d((tx2 if tx2 <= x2 else x2) - (tx1 if tx1 >= x1 else x1)) + 1.0)
            if iw > 0:
                ih = f(d((ty2 if ty2 <= y2 else y2) - (ty1 if ty1 >= y1 else y1)) + 1.0)
                if ih > 0:
                    ua = f((d(tx2 - tx1) + 1.0) * (d(ty2 - ty1) + 1.0) + d(area) - d(iw * ih))
                    ov = (iw * ih) / ua
                    if method == 1:
                        weight = f(1.0 - d(ov)) if ov > Nt else f(1)
                    elif method == 2:
                        weight = f(np.exp(d(-(ov * ov) / sigma)))
                    else:
                        weight = f(0) if ov > Nt else f(1)
                    boxes[pos, 4] = weight * boxes[pos, 4]
                    if boxes[pos, 4] < threshold:
                        boxes[pos, 0:5] = boxes[N - 1, 0:5]
                        N -= 1
                        pos -= 1
            pos += 1
    return list(range(N))


def merge_outputs(detections, num_classes, max_per_image, nms=False):
    """detectors/polydet.py:62-76; soft-nms (Nt=0.5, gaussian) when several scales were run
    or --nms is set (`nms=True`)."""
    results = {}
    for j in range(1, num_classes + 1):
        results[j] = np.concatenate([d[j] for d in detections], axis=0).astype(np.float32)
        if nms:
            soft_nms(results[j], Nt=0.5, method=2)
    scores = np.hstack([results[j][:, 4] for j in range(1, num_classes + 1)])
    if len(scores) > max_per_image:
        kth = len(scores) - max_per_image
        thresh = np.partition(scores, kth)[kth]
        for j in range(1, num_classes + 1):
            results[j] = results[j][results[j][:, 4] >= thresh]
    return results

"""Oracle: the detector's image pre-processing (numpy).  TEST INFRASTRUCTURE.

Follows BaseDetector.pre_process (src/lib/detectors/base_detector.py:40-87):
    cv2.resize -> cv2.warpAffine(..., flags=cv2.INTER_LINEAR) -> (x / 255. - mean) / std
    -> CHW float32 (+ the horizontally flipped copy under --flip_test).

`cv2.warpAffine` lives in a dependency that is NOT under /root/reference and is not installed
in this image (opencv-python, unpinned in the reference's requirements.txt), so PARITY OF THIS
FUNCTION IS UNPINNED: it restates the published OpenCV algorithm for 8-bit images
(modules/imgproc/src/imgwarp.cpp, cv::warpAffine + remapBilinear, unchanged across 3.x/4.x):
  * the forward matrix is inverted in float64 (dst -> src),
  * source coordinates are fixed point with 10 fractional bits,
        X = (cvRound(M0*x*1024) + cvRound((M1*y + M2)*1024) + 16) >> 5,
    i.e. 5 fractional bits (1/32 px) after rounding; cvRound = round-half-to-even,
  * bilinear weights are 15-bit integers (32-fx)(32-fy)*32 ... (they sum to 2^15 exactly, so
    OpenCV's weight fix-up never fires for INTER_LINEAR),
  * each channel = (sum w_i * p_i + 2^14) >> 15, taps outside the source read the constant
    border 0, and the result is uint8.
When the affine map is an integer translation (the reference's default `keep_res` test mode:
s = (inp_w, inp_h), c = the integer centre) every weight is 0 or 2^15 and the warp is an exact
copy, so the unpinned part only matters under --fix_res / non-unit scales.
"""
import numpy as np

from .post import get_affine_transform

AB_BITS, INTER_BITS = 10, 5
AB_SCALE = 1 << AB_BITS
ROUND_DELTA = AB_SCALE // (1 << INTER_BITS) // 2          # 16


def invert_affine(M):
    """cv::warpAffine's in-place inversion of the 2x3 forward map (float64)."""
    M = np.array(M, dtype=np.float64).reshape(2, 3).copy()
    D = M[0, 0] * M[1, 1] - M[0, 1] * M[1, 0]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = M[1, 1] * D, M[0, 0] * D
    M[0, 0] = A11
    M[0, 1] *= -D
    M[1, 0] *= -D
    M[1, 1] = A22
    b1 = -M[0, 0] * M[0, 2] - M[0, 1] * M[1, 2]
    b2 = -M[1, 0] * M[0, 2] - M[1, 1] * M[1, 2]
    M[0, 2], M[1, 2] = b1, b2
    return M


def warp_affine_u8(img, M, dsize):
    """img uint8 [H,W,C], M forward 2x3 (src -> dst), dsize = (width, height) -> uint8 [h,w,C]."""
    assert img.dtype == np.uint8 and img.ndim == 3
    H, W, C = img.shape
    dw, dh = int(dsize[0]), int(dsize[1])
    Mi = invert_affine(M)
    xs = np.arange(dw, dtype=np.float64)
    ys = np.arange(dh, dtype=np.float64)
    adelta = np.rint(Mi[0, 0] * xs * AB_SCALE).astype(np.int64)
    bdelta = np.rint(Mi[1, 0] * xs * AB_SCALE).astype(np.int64)
    X0 = np.rint((Mi[0, 1] * ys + Mi[0, 2]) * AB_SCALE).astype(np.int64) + ROUND_DELTA
    Y0 = np.rint((Mi[1, 1] * ys + Mi[1, 2]) * AB_SCALE).astype(np.int64) + ROUND_DELTA
    X = (X0[:, None] + adelta[None, :]) >> (AB_BITS - INTER_BITS)
    Y = (Y0[:, None] + bdelta[None, :]) >> (AB_BITS - INTER_BITS)
    # arithmetic shift = floor; saturate_cast<short> of the integer part
    sx = np.clip(X >> INTER_BITS, -32768, 32767)
    sy = np.clip(Y >> INTER_BITS, -32768, 32767)
    fx, fy = X & 31, Y & 31
    w00 = (32 - fx) * (32 - fy) * 32
    w01 = fx * (32 - fy) * 32
    w10 = (32 - fx) * fy * 32
    w11 = fx * fy * 32
    src = img.astype(np.int64)

    def tap(yy, xx):
        ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
        v = src[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)]
        return v * ok[..., None]

    acc = (w00[..., None] * tap(sy, sx) + w01[..., None] * tap(sy, sx + 1)
           + w10[..., None] * tap(sy + 1, sx) + w11[..., None] * tap(sy + 1, sx + 1))
    return ((acc + (1 << 14)) >> 15).astype(np.uint8)


def normalize_chw(inp_u8, mean, std):
    """((inp / 255. - mean) / std).astype(float32).transpose(2,0,1): numpy promotes to float64."""
    mean = np.asarray(mean, dtype=np.float32).reshape(1, 1, 3)
    std = np.asarray(std, dtype=np.float32).reshape(1, 1, 3)
    return ((inp_u8 / 255. - mean) / std).astype(np.float32).transpose(2, 0, 1)


def pre_process(image, scale, mean, std, fix_res=False, input_h=None, input_w=None, pad=31,
                down_ratio=4, flip_test=False):
    """base_detector.py:50-87 for scale == 1 (cv2.resize to the same size is the identity)."""
    assert scale == 1, "cv2.resize is not restated"
    height, width = image.shape[0:2]
    new_height, new_width = int(height * scale), int(width * scale)
    if fix_res:
        inp_height, inp_width = input_h, input_w
        c = np.array([new_width / 2., new_height / 2.], dtype=np.float32)
        s = max(height, width) * 1.0
    else:
        inp_height = (new_height | pad) + 1
        inp_width = (new_width | pad) + 1
        c = np.array([new_width // 2, new_height // 2], dtype=np.float32)
        s = np.array([inp_width, inp_height], dtype=np.float32)
    trans_input = get_affine_transform(c, s, 0, [inp_width, inp_height])
    inp = warp_affine_u8(image, trans_input, (inp_width, inp_height))
    images = normalize_chw(inp, mean, std).reshape(1, 3, inp_height, inp_width)
    if flip_test:
        images = np.concatenate((images, images[:, :, :, ::-1]), axis=0)
    meta = {"c": c, "s": s, "out_height": inp_height // down_ratio,
            "out_width": inp_width // down_ratio}
    return images, meta, trans_input


def color_aug_normalize(inp01, order, alphas, light_delta, mean, std, color_on=True):
    """Colour augmentation + normalisation of the training sampler on a float32 BGR HWC image holding
    x / 255 (src/lib/datasets/sample/polydet.py:128-136 -> src/lib/utils/image.py:231-264), with the random
    draws passed in: `order` = the three ops in application order (0 brightness, 1 contrast, 2 saturation),
    `alphas` their factors, `light_delta` = eig_vec . (eig_val * alpha_pca) (float64[3]).  numpy float32
    arithmetic in the reference's order; cv2.cvtColor(BGR2GRAY) is restated as 0.114 B + 0.587 G + 0.299 R in
    float32 (OpenCV itself is absent: unpinned, as the warp above).  Returns float32 CHW."""
    img = np.array(inp01, dtype=np.float32, copy=True)
    if color_on:
        gs = (img[..., 0] * np.float32(0.114) + img[..., 1] * np.float32(0.587)) + img[..., 2] * np.float32(0.299)
        gs = gs.astype(np.float32)
        gs_mean = np.float32(np.float64(gs.astype(np.float64).sum()) / gs.size)
        for op, alpha in zip(order, alphas):
            a = np.float32(alpha)
            img *= a
            if op == 1:
                img += gs_mean * (np.float32(1.0) - a)
            elif op == 2:
                img += (gs * (np.float32(1.0) - a))[:, :, None]
        img = (img.astype(np.float64) + np.asarray(light_delta, np.float64).reshape(1, 1, 3)).astype(np.float32)
    out = (img - np.asarray(mean, np.float32).reshape(1, 1, 3)) / np.asarray(std, np.float32).reshape(1, 1, 3)
    return np.ascontiguousarray(out.transpose(2, 0, 1))

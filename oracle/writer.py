"""Oracle: the Cityscapes result writer of the polydet evaluation,
`CITYSCAPES.format_and_write_to_cityscapes` (src/lib/datasets/dataset/cityscapes.py:196-283).
TEST INFRASTRUCTURE.

Per image: detections above `thresh`, vertices truncated to integers, processed in ascending depth:
PIL polygon fill + outline, every pixel of the closed Bresenham contour dilated with a PIL ellipse of
radius 2, the part hidden by nearer instances removed (`polygon_mask * (1 - to_remove_mask)`); instances
with score >= 0.5 are added to the removal mask; masks with more than 100 pixels are written as PNG files
and listed as `masks/<image>_<count>.png <label id> <min(1, 1.2 score)>`.

Restated with the SAME PIL calls the reference makes (PIL 12.2 is installed in this image: its own
rasteriser produces the masks), and with the `bresenham` package (absent here; pinned version unknown)
restated from its published algorithm (`bresenham(x0, y0, x1, y1)`: integer line walk that yields both end
points).  The reference file itself cannot be imported (pycocotools, cv2, bresenham, shapely, wandb): the
fixtures of tests/golden/gen_writer_golden.py are this restatement's output under the installed PIL."""
import numpy as np

NO_MASK_LABELS = ("pole", "traffic sign", "traffic light")


def bresenham(x0, y0, x1, y1):
    """The PyPI `bresenham` package's generator, restated."""
    dx, dy = x1 - x0, y1 - y0
    xsign = 1 if dx > 0 else -1
    ysign = 1 if dy > 0 else -1
    dx, dy = abs(dx), abs(dy)
    if dx > dy:
        xx, xy, yx, yy = xsign, 0, 0, ysign
    else:
        dx, dy = dy, dx
        xx, xy, yx, yy = 0, ysign, xsign, 0
    D = 2 * dy - dx
    y = 0
    for x in range(dx + 1):
        yield x0 + x * xx + y * yx, y0 + x * xy + y * yy
        if D >= 0:
            y += 1
            D -= 2 * dx
        D += 2 * dy


def _to_float(x):
    return float("{:.2f}".format(x))


def image_instances(per_class, class_name, thresh):
    """cityscapes.py:225-238: [(points, score, label, depth)] of one image, sorted by depth."""
    params = []
    for cls_ind in per_class:
        if cls_ind == "fg":
            continue
        for bbox in per_class[cls_ind]:
            if bbox[4] > thresh:
                polygon = list(map(_to_float, bbox[5:-1]))
                polygon = [(int(x), int(y)) for x, y in zip(polygon[0::2], polygon[1::2])]
                params.append((polygon, bbox[4], class_name[cls_ind], bbox[-1]))
    return sorted(params, key=lambda a: a[-1])


def instance_masks(params, width=2048, height=1024):
    """cityscapes.py:240-272: the occlusion-ordered masks [(mask uint8 HxW, keep)] of sorted instances."""
    from PIL import Image, ImageDraw
    ones = np.ones((height, width))
    to_remove = np.zeros((height, width))
    out = []
    for points, score, label, depth in params:
        pm = Image.new("L", (width, height), 0)
        if label not in NO_MASK_LABELS:
            ImageDraw.Draw(pm).polygon(points, outline=255, fill=255)
            contour = list(bresenham(points[-1][0], points[-1][1], points[0][0], points[0][1]))
            for i in range(len(points) - 1):
                contour += bresenham(points[i][0], points[i][1], points[i + 1][0], points[i + 1][1])
            radius = 2
            for p in set(contour):
                ImageDraw.Draw(pm).ellipse([(p[0] - radius, p[1] - radius), (p[0] + radius, p[1] + radius)],
                                           outline=255, fill=255)
            pm = Image.fromarray(np.array(pm) * (ones - to_remove).astype(np.uint8))
        arr = np.array(pm)
        if score >= 0.5:
            to_remove += arr
            to_remove[to_remove > 0] = 1
        keep = label not in NO_MASK_LABELS and np.count_nonzero(arr) > 100
        out.append((arr, keep))
    return out


def format_image(per_class, image_name, class_name, label_to_id, thresh, width=2048, height=1024):
    """One image of format_and_write_to_cityscapes: (text lines, {mask file name: mask array})."""
    import os
    params = image_instances(per_class, class_name, thresh)
    lines, files, count = [], {}, 0
    base = os.path.basename(image_name)
    for (points, score, label, depth), (arr, keep) in zip(params, instance_masks(params, width, height)):
        if keep:
            name = base.replace(".png", "_" + str(count) + ".png")
            lines.append("masks/" + name + " " + str(label_to_id[label]) + " " + str(min(1, score * 1.2)) + "\n")
            files[name] = arr
            count += 1
    return lines, files

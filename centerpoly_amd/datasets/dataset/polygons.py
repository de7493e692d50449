"""The reference's polygon datasets (src/lib/datasets/dataset/{cityscapes,kitti_poly,IDD}.py) without
pycocotools / cv2: the annotation files are plain COCO-style JSON (`images`, `annotations` with `bbox`
[x, y, w, h], `poly` [2N numbers], `category_id`, `pseudo_depth`; `categories`), indexed here with the
`json` module; images are read with PIL and handed on as 8-bit BGR arrays (what `cv2.imread` returns).

Constants (class lists, class frequencies, mean / std, default resolution, colour-augmentation PCA)
are the datasets' own.  Paths: the reference hard-codes `../cityscapesStuff/BBoxes` etc. relative to
`src/`; here `--annot_dir` / `--data_dir` say where the JSON files and the images are, and a missing
file is an error that names the path -- no silent substitution by synthetic data."""
import json
import os

import numpy as np
import torch.utils.data as data


class CocoIndex(object):
    """The four pycocotools.coco.COCO calls the sampler uses (sample/polydet.py:69-77)."""

    def __init__(self, annot_path):
        with open(annot_path) as f:
            d = json.load(f)
        self.imgs = {im["id"]: im for im in d["images"]}
        self.anns = {a["id"]: a for a in d["annotations"]}
        self.cats = {c["id"]: c for c in d.get("categories", [])}
        self._by_img = {}
        for a in d["annotations"]:
            self._by_img.setdefault(a["image_id"], []).append(a["id"])

    def getImgIds(self):
        return list(self.imgs.keys())

    def loadImgs(self, ids):
        return [self.imgs[i] for i in ids]

    def getAnnIds(self, imgIds):
        return [a for i in imgIds for a in self._by_img.get(i, [])]

    def loadAnns(self, ids):
        return [self.anns[i] for i in ids]


class PolygonDataset(data.Dataset):
    num_classes = 8
    default_resolution = [512, 1024]
    max_objs = 128
    # PCA of the colour augmentation (identical in the three dataset files)
    _eig_val = np.array([0.2141788, 0.01817699, 0.00341571], dtype=np.float32)
    _eig_vec = np.array([[-0.58752847, -0.69563484, 0.41340352], [-0.5832747, 0.00994535, -0.81221408],
                         [-0.56089297, 0.71832671, 0.41158938]], dtype=np.float32)
    class_name = ["__background__", "person", "rider", "car", "truck", "bus", "train", "motorcycle", "bicycle"]
    _valid_ids = [1, 2, 3, 4, 5, 6, 7, 8]
    annot_subdir = ""
    name = ""

    def annot_file(self, split):
        raise NotImplementedError

    def __init__(self, opt, split):
        super(PolygonDataset, self).__init__()
        self.opt = opt
        self.split = split
        annot_dir = getattr(opt, "annot_dir", "") or os.path.join(opt.data_dir, self.annot_subdir)
        self.annot_path = os.path.join(annot_dir, self.annot_file(split))
        self.img_dir = getattr(opt, "img_dir", "") or os.path.join(opt.data_dir, self.name, "images")
        if not os.path.isfile(self.annot_path):
            raise FileNotFoundError(
                "dataset %r (%s split): annotation file %s not found -- pass --annot_dir / --data_dir, or use "
                "--dataset synthetic for the offline synthetic set" % (self.name, split, self.annot_path))
        self.cat_ids = {v: i for i, v in enumerate(self._valid_ids)}
        self._data_rng = np.random.RandomState(123)
        print("==> initializing %s %s data." % (self.name, split))
        self.coco = CocoIndex(self.annot_path)
        self.images = self.coco.getImgIds()
        self.num_samples = len(self.images)
        print("Loaded {} {} samples".format(split, self.num_samples))

    def __len__(self):
        return self.num_samples

    def read_image(self, file_name):
        """8-bit BGR [H, W, 3] (cv2.imread's layout).  Absolute paths of the annotation files are
        re-rooted under img_dir by their base name when they do not exist as given."""
        from PIL import Image
        path = file_name if os.path.isabs(file_name) and os.path.exists(file_name) else \
            os.path.join(self.img_dir, os.path.basename(file_name))
        if not os.path.exists(path):
            raise FileNotFoundError("image %s not found (img_dir %s)" % (file_name, self.img_dir))
        return np.ascontiguousarray(np.asarray(Image.open(path).convert("RGB"))[:, :, ::-1])

    def run_eval(self, results, save_dir):
        """The vendored evaluators (cityscapesscripts, pycocotools) are outside the accelerated path: the
        detections are written in the reference's json layout (convert_polygon_eval_format) for them."""
        dets = []
        for image_id, per_cls in results.items():
            for cls_ind, rows in per_cls.items():
                for row in rows:
                    dets.append({"image_id": int(image_id), "category_id": int(self._valid_ids[cls_ind - 1]),
                                 "bbox": [float("%.2f" % v) for v in row[0:4]], "score": float("%.2f" % row[4]),
                                 "polygon": [float("%.2f" % v) for v in row[5:-1]],
                                 "depth": float(row[-1])})
        os.makedirs(save_dir, exist_ok=True)
        with open(os.path.join(save_dir, "results.json"), "w") as f:
            json.dump(dets, f)
        print("%s: wrote %d detections to %s/results.json" % (self.name, len(dets), save_dir))
        return 0.0


def _to_float(x):
    return float("{:.2f}".format(x))


class CityscapesWriterMixin(object):
    """format_and_write_to_cityscapes (src/lib/datasets/dataset/cityscapes.py:196-283): per image a text
    file `<image>.txt` listing `masks/<image>_<k>.png <label id> <confidence>` and the instance masks as
    8-bit PNG files, instances processed in ascending depth, nearer confident ones hiding farther ones.
    The masks are rasterised on the GPU (cp_instance_masks); selection, ordering, naming and the file
    output (PIL, like the reference) stay on the host."""
    canvas = (2048, 1024)                                    # (width, height), hard-coded by the reference
    no_mask_labels = ("pole", "traffic sign", "traffic light")

    def image_instances(self, per_class):
        params = []
        for cls_ind in per_class:
            if cls_ind == "fg":
                continue
            for row in per_class[cls_ind]:
                if row[4] > self.opt.thresh:
                    poly = [_to_float(v) for v in row[5:-1]]
                    pts = [(int(x), int(y)) for x, y in zip(poly[0::2], poly[1::2])]
                    params.append((pts, row[4], self.class_name[cls_ind], row[-1]))
        return sorted(params, key=lambda a: a[-1])

    def instance_masks(self, params, device=None):
        """Occlusion-ordered masks of depth-sorted instances: (uint8 [n, H, W] host array, counts [n])."""
        import torch

        from ... import _C
        W, H = self.canvas
        n = len(params)
        if n == 0:
            return np.zeros((0, H, W), np.uint8), np.zeros((0,), np.int32)
        if n > 128:
            raise ValueError("more than 128 instances in one image (max_per_image = K <= 128)")
        N = len(params[0][0])
        dev = device or torch.device("cuda")
        poly = torch.tensor([p[0] for p in params], dtype=torch.int32).reshape(n, N, 2).to(dev)
        flags = torch.tensor([(0 if p[2] in self.no_mask_labels else 1) | (2 if p[1] >= 0.5 else 0) for p in params],
                             dtype=torch.uint8).to(dev)
        masks = torch.empty((n, H, W), dtype=torch.uint8, device=dev)
        counts = torch.empty((n,), dtype=torch.int32, device=dev)
        _C.check(_C.lib().cp_instance_masks(_C.ptr(poly), _C.ptr(flags), n, N, H, W, _C.ptr(masks), _C.ptr(counts),
                                            _C.stream()), "cp_instance_masks")
        return masks.cpu().numpy(), counts.cpu().numpy()

    def format_and_write_to_cityscapes(self, all_bboxes, save_dir):
        from PIL import Image
        id_to_file = {im["id"]: im["file_name"] for im in self.coco.imgs.values()}
        masks_dir = os.path.join(save_dir, "masks")
        os.makedirs(masks_dir, exist_ok=True)
        for image_id in all_bboxes:
            base = os.path.basename(id_to_file[int(image_id)])
            params = self.image_instances(all_bboxes[image_id])
            masks, counts = self.instance_masks(params)
            count = 0
            with open(os.path.join(save_dir, base.replace(".png", ".txt")), "w") as text_file:
                for (pts, score, label, depth), mask, nz in zip(params, masks, counts):
                    if label not in self.no_mask_labels and nz > 100:
                        name = base.replace(".png", "_" + str(count) + ".png")
                        text_file.write("masks/" + name + " " + str(self.label_to_id[label]) + " "
                                        + str(min(1, score * 1.2)) + "\n")
                        count += 1
                        Image.fromarray(mask).save(os.path.join(masks_dir, name))


class CITYSCAPES(CityscapesWriterMixin, PolygonDataset):
    """src/lib/datasets/dataset/cityscapes.py:39-110."""
    name = "cityscapes"
    annot_subdir = os.path.join("cityscapesStuff", "BBoxes")
    mean = np.array([0.28404999637454165, 0.32266921542410754, 0.2816898182839038], dtype=np.float32).reshape(1, 1, 3)
    std = np.array([0.04230349568017417, 0.04088212241688149, 0.04269893084955519], dtype=np.float32).reshape(1, 1, 3)
    class_name = PolygonDataset.class_name + ["pole", "traffic sign", "traffic light"]
    class_frequencies = {"person": 0.14062428170827013, "rider": 0.015518384984665498, "car": 0.20898266905714155,
                         "truck": 0.003822132907776267, "bus": 0.0031719762791339126,
                         "train": 0.0012740443025920892, "motorcycle": 0.005831707941761728,
                         "bicycle": 0.0322057384531526, "pole": 0.34640870553158515,
                         "traffic sign": 0.16402335310072175, "traffic light": 0.07813700573319936}

    label_to_id = {"person": 24, "rider": 25, "car": 26, "truck": 27, "bus": 28, "train": 31, "motorcycle": 32,
                   "bicycle": 33, "pole": -1, "traffic sign": -1, "traffic light": -1}

    def annot_file(self, split):
        if split == "test":
            return "test.json"
        return "%s%d_regular_interval.json" % ("val" if split == "val" else "train", self.opt.nbr_points)

    def run_eval(self, results, save_dir):
        """cityscapes.py:400-432 up to the vendored evaluator: results.json + the per-image mask files the
        Cityscapes instance-level evaluation reads (evalInstanceLevelSemanticLabeling itself is outside the
        accelerated path)."""
        super(CITYSCAPES, self).run_eval(results, save_dir)
        res_dir = os.path.join(save_dir, "results")
        os.makedirs(res_dir, exist_ok=True)
        self.format_and_write_to_cityscapes(results, res_dir)
        return 0.0


class KITTIPOLY(PolygonDataset):
    """src/lib/datasets/dataset/kitti_poly.py:15-60."""
    name = "kitti_poly"
    annot_subdir = os.path.join("KITTIPolyStuff", "BBoxes")
    mean = np.array([0.485, 0.456, 0.406], np.float32).reshape(1, 1, 3)
    std = np.array([0.229, 0.224, 0.225], np.float32).reshape(1, 1, 3)
    class_frequencies = {"person": 0.15, "rider": 0.03, "car": 0.20, "truck": 0.03, "bus": 0.03, "train": 0.03,
                         "motorcycle": 0.03, "bicycle": 0.03}

    def annot_file(self, split):
        if split == "test":
            return "test.json"
        return "%s%d.json" % ("val" if split == "val" else "train", self.opt.nbr_points)


class IDD(PolygonDataset):
    """src/lib/datasets/dataset/IDD.py:15-60 (9 classes, the Cityscapes statistics)."""
    name = "IDD"
    num_classes = 9
    annot_subdir = os.path.join("IDDStuff", "BBoxes")
    mean = CITYSCAPES.mean
    std = CITYSCAPES.std
    class_name = ["__background__", "person", "rider", "motorcycle", "bicycle", "autorickshaw", "car", "truck",
                  "bus", "vehicle fallback"]
    _valid_ids = [1, 2, 3, 4, 5, 6, 7, 8, 9]
    class_frequencies = {"person": 0.15, "rider": 0.03, "car": 0.20, "truck": 0.03, "bus": 0.03, "motorcycle": 0.03,
                         "bicycle": 0.03, "autorickshaw": 0.33, "vehicle fallback": 0.18}

    def annot_file(self, split):
        if split == "test":
            return "test.json"
        return "%s%d_regular_interval.json" % ("val" if split == "val" else "train", self.opt.nbr_points)

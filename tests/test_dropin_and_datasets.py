"""B2 boundary and the callers either side of the path, on the CPU:
  * `install_as_reference_lib()` makes the reference drivers' own import lines resolve (src/main.py:12-19,
    src/test.py:16-21, the sampler's `from utils.image import ...`), with the signatures SURVEY 8(b) lists
  * dataset factory: reference names map to real dataset classes that refuse missing data; the reference's
    own KITTI annotation JSON is read and sampled (host half of the sampler)
  * host colour augmentation == the oracle's restatement."""
import inspect
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_ANN = "/root/reference/KITTIPolyStuff/BBoxes"


def test_reference_import_lines_resolve_after_install():
    code = textwrap.dedent('''
        import sys
        sys.path.insert(0, %r)
        import centerpoly_amd
        centerpoly_amd.install_as_reference_lib()
        # src/main.py:12-19
        from opts import opts
        from models.model import create_model, load_model, save_model
        from models.data_parallel import DataParallel
        from logger import Logger
        from datasets.dataset_factory import get_dataset
        from trains.train_factory import train_factory
        # src/test.py:16-21
        from external.nms import soft_nms
        from utils.utils import AverageMeter
        from datasets.dataset_factory import dataset_factory
        from detectors.detector_factory import detector_factory
        # the plugin slot and the sampler's helpers (pose_dla_dcn.py:16, sample/polydet.py:11-14)
        from models.networks.DCNv2.dcn_v2 import DCN
        from utils.image import flip, color_aug, get_affine_transform, affine_transform
        from utils.image import gaussian_radius, draw_umich_gaussian
        from models.losses import FocalLoss, RegL1Loss, PolyLoss
        from models.decode import polydet_decode
        from utils.post_process import polydet_post_process
        import inspect
        assert list(inspect.signature(create_model).parameters)[:3] == ["arch", "heads", "head_conv"]
        assert list(inspect.signature(DCN.__init__).parameters)[1:6] == \\
            ["in_channels", "out_channels", "kernel_size", "stride", "padding"]
        assert "polydet" in train_factory and "polydet" in detector_factory
        for name in ("set_device", "train", "val", "_get_losses", "save_result"):
            assert hasattr(train_factory["polydet"], name), name
        for name in ("run", "pre_process", "process", "post_process", "merge_outputs"):
            assert hasattr(detector_factory["polydet"], name), name
        o = opts().parse(["polydet"])
        assert o.arch == "dla_34" and o.head_conv == 256 and o.pad == 31          # reference defaults
        assert {"cityscapes", "kitti_poly", "IDD"} <= set(dataset_factory)
        print("ok")
    ''') % ROOT
    p = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0 and p.stdout.strip().endswith(b"ok"), p.stderr.decode()[-800:]


def test_data_parallel_refuses_several_devices_in_one_process():
    import torch
    from centerpoly_amd.models.data_parallel import DataParallel
    m = torch.nn.Linear(2, 2)
    assert DataParallel(m, device_ids=[0]) is m
    with pytest.raises(RuntimeError, match="one process per GPU"):
        DataParallel(m, device_ids=[0, 1], chunk_sizes=[3, 5])


def test_real_dataset_names_refuse_missing_data(tmp_path):
    from centerpoly_amd.datasets.dataset_factory import get_dataset
    from centerpoly_amd.opts import opts
    opt = opts().parse(["polydet", "--dataset", "cityscapes", "--root_dir", str(tmp_path)])
    Dataset = get_dataset(opt.dataset, opt.task)
    opt = opts().update_dataset_info_and_set_heads(opt, Dataset)
    assert float(opt.mean.ravel()[0]) == pytest.approx(0.28404999637454165)        # the dataset's own statistics
    with pytest.raises(FileNotFoundError, match="annotation file"):
        Dataset(opt, "train")
    with pytest.raises(KeyError):
        get_dataset("coco", "polydet")


@pytest.mark.skipif(not os.path.isdir(REF_ANN), reason="the reference's annotation files are not on this box")
def test_kitti_annotations_are_read_and_sampled(tmp_path):
    """The reference's own val16.json through the JSON index and the host half of the sampler; images are
    generated PNG files (the dataset's images are not available offline)."""
    import json
    from PIL import Image
    from centerpoly_amd.datasets.dataset_factory import get_dataset
    from centerpoly_amd.opts import opts
    ann = json.load(open(os.path.join(REF_ANN, "val16.json")))
    rng = np.random.RandomState(0)
    for im in ann["images"]:
        Image.fromarray(rng.randint(0, 255, (376, 1242, 3), dtype=np.uint8)).save(
            str(tmp_path / os.path.basename(im["file_name"])))
    opt = opts().parse(["polydet", "--dataset", "kitti_poly", "--annot_dir", REF_ANN, "--img_dir", str(tmp_path),
                        "--input_h", "384", "--input_w", "1280", "--nbr_points", "16"])
    Dataset = get_dataset(opt.dataset, opt.task)
    opt = opts().update_dataset_info_and_set_heads(opt, Dataset)
    for split in ("val", "train"):
        ds = Dataset(opt, "val")
        ds.split = split
        assert len(ds) == len(ann["images"])
        np.random.seed(3)
        item = ds[0]
        n_gt = sum(1 for a in ann["annotations"] if a["image_id"] == ann["images"][0]["id"])
        assert int(item["num_objs"]) == min(n_gt, 128)
        assert item["image_u8"].shape == (376, 1242, 3) and item["image_u8"].dtype == np.uint8
        assert item["poly"].shape == (128, 32) and item["trans_input"].shape == (6,)
        first = next(a for a in ann["annotations"] if a["image_id"] == ann["images"][0]["id"])
        np.testing.assert_allclose(item["poly"][0], first["poly"])
        np.testing.assert_allclose(item["bbox"][0], [first["bbox"][0], first["bbox"][1],
                                                     first["bbox"][0] + first["bbox"][2],
                                                     first["bbox"][1] + first["bbox"][3]])
        if split == "train":
            assert item["color"][0] == 1.0 and sorted(item["color"][1:4]) == [0.0, 1.0, 2.0]
            assert np.all(np.abs(item["color"][4:7] - 1.0) <= 0.4)
        else:
            assert item["color"][0] == 0.0 and "meta" in item


def test_host_color_aug_matches_the_oracle_restatement():
    from centerpoly_amd.utils import image as I
    from oracle import pre as opre
    rng = np.random.RandomState(5)
    img = rng.rand(20, 31, 3).astype(np.float32)
    eig_val = np.array([0.2141788, 0.01817699, 0.00341571], dtype=np.float32)
    eig_vec = np.array([[-0.58752847, -0.69563484, 0.41340352], [-0.5832747, 0.00994535, -0.81221408],
                        [-0.56089297, 0.71832671, 0.41158938]], dtype=np.float32)
    import random
    random.seed(11)
    a = img.copy()
    I.color_aug(np.random.RandomState(7), a, eig_val, eig_vec)
    random.seed(11)
    order, alphas, light = I.color_aug_params(np.random.RandomState(7), random)
    delta = np.dot(eig_vec.astype(np.float64), eig_val.astype(np.float64) * light)
    ref = opre.color_aug_normalize(img, order, alphas, delta, (0, 0, 0), (1, 1, 1))
    np.testing.assert_allclose(a.transpose(2, 0, 1), ref, rtol=2e-6, atol=2e-6)


@pytest.mark.gpu
def test_device_color_aug_and_training_input_pipeline():
    """cp_color_aug_normalize against the oracle on the device-warped image (bit-exact expected: both apply
    the same float32 operations in the same order), with and without colour augmentation."""
    import torch
    from centerpoly_amd.datasets.sample.polydet import build_inputs
    from centerpoly_amd.utils.image import get_affine_transform, warp_affine_normalize
    from oracle import pre as opre
    rng = np.random.RandomState(9)
    img = rng.randint(0, 255, (2, 96, 160, 3), dtype=np.uint8)
    trans = np.stack([get_affine_transform(np.array([80., 48.], np.float32), 160.0 * s, 0, [128, 64]).reshape(6)
                      for s in (0.8, 1.2)])
    color = np.array([[1, 2, 0, 1, 0.7, 1.3, 0.9, 0.01, -0.02, 0.005], [0, 0, 0, 0, 1, 1, 1, 0, 0, 0]], np.float64)
    mean, std = (0.284, 0.323, 0.282), (0.0423, 0.0409, 0.0427)
    dev = torch.device("cuda")
    out = build_inputs(torch.from_numpy(img).to(dev), trans, color, mean, std, 64, 128).cpu().numpy()
    for b in range(2):
        w01 = warp_affine_normalize(torch.from_numpy(img[b]).to(dev), trans[b], (0, 0, 0), (1, 1, 1), 64, 128)[0]
        hwc = w01.cpu().numpy().transpose(1, 2, 0)
        ref = opre.color_aug_normalize(hwc, color[b, 1:4].astype(int), color[b, 4:7], color[b, 7:10], mean, std,
                                       color_on=bool(color[b, 0]))
        np.testing.assert_allclose(out[b], ref, rtol=1e-6, atol=1e-5)

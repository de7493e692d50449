"""BaseDetector (reference: src/lib/detectors/base_detector.py:18-191).

Same surface: BaseDetector(opt); .pre_process(image, scale, meta) -> (images, meta);
.run(image_or_path_or_tensor) -> {'results', 'tot','load','pre','net','dec','post','merge'}.
The accelerated path is GPU-only: opt.gpus = [-1] raises instead of silently
running on the host.  cv2 is not required: the 8-bit image is uploaded as is and
the affine warp + normalisation of pre_process run in one HIP kernel with OpenCV's
fixed-point arithmetic (cp_preprocess_warp_normalize), a .npy path or ndarray is
accepted as the image source, and `opt.load_model == ''` keeps the random
initialisation (the reference calls torch.load unconditionally, :27).
"""
import time

import numpy as np
import torch

from ..models.model import create_model, load_model
from ..utils.image import get_affine_transform, warp_affine_normalize


def _resize_bilinear_u8(img_dev, new_w, new_h):
    """cv2.resize stand-in for --test_scales != 1 (device, bilinear, result rounded back to 8 bits;
    OpenCV's fixed-point resize can differ from it by one grey level)."""
    t = img_dev.permute(2, 0, 1)[None].float()
    t = torch.nn.functional.interpolate(t, size=(new_h, new_w), mode="bilinear", align_corners=False)
    return t[0].permute(1, 2, 0).round_().clamp_(0, 255).to(torch.uint8).contiguous()


class BaseDetector(object):
    def __init__(self, opt):
        if opt.gpus[0] < 0:
            raise RuntimeError("centerpoly_amd detectors run on a HIP device only (--gpus -1 "
                               "has no CPU fallback)")
        opt.device = torch.device("cuda")
        print("Creating model...")
        self.model = create_model(opt.arch, opt.heads, opt.head_conv)
        if getattr(opt, "load_model", ""):
            self.model = load_model(self.model, opt.load_model)
        self.model = self.model.to(opt.device)
        self.model.eval()
        if hasattr(self.model, "prepare_inference"):
            self.model.prepare_inference()      # BN folded into convs / the DCN epilogue
        self.mean = np.array(opt.mean, dtype=np.float32).reshape(1, 1, 3)
        self.std = np.array(opt.std, dtype=np.float32).reshape(1, 1, 3)
        self.max_per_image = opt.K
        self.num_classes = opt.num_classes
        self.scales = opt.test_scales
        self.opt = opt
        self.pause = True

    def pre_process(self, image, scale, meta=None):
        height, width = image.shape[0:2]
        new_height, new_width = int(height * scale), int(width * scale)
        if self.opt.fix_res:
            inp_height, inp_width = self.opt.input_h, self.opt.input_w
            c = np.array([new_width / 2.0, new_height / 2.0], dtype=np.float32)
            s = max(height, width) * 1.0
        else:
            inp_height = (new_height | self.opt.pad) + 1
            inp_width = (new_width | self.opt.pad) + 1
            c = np.array([new_width // 2, new_height // 2], dtype=np.float32)
            s = np.array([inp_width, inp_height], dtype=np.float32)
        trans_input = get_affine_transform(c, s, 0, [inp_width, inp_height])
        src = self._upload(image)
        if (new_height, new_width) != (height, width):
            src = _resize_bilinear_u8(src, new_width, new_height)
        images = warp_affine_normalize(src, trans_input, self.mean, self.std, inp_height, inp_width,
                                       flip_copy=bool(self.opt.flip_test))
        meta = {"c": c, "s": s, "out_height": inp_height // self.opt.down_ratio,
                "out_width": inp_width // self.opt.down_ratio}
        return images, meta

    def _upload(self, image):
        """8-bit HWC image -> device.  A plain (pageable) copy: measured on MI355X it costs
        0.17 ms for a 2048x1024 image and is steady, while staging through a pinned buffer with
        a non-blocking copy showed periodic ~90 ms stalls (tools/probe_stalls.py)."""
        if image.dtype != np.uint8 or image.ndim != 3 or image.shape[2] != 3:
            raise TypeError("pre_process needs an 8-bit [H,W,3] image (got %s %s)"
                            % (image.dtype, image.shape))
        return torch.from_numpy(np.ascontiguousarray(image)).to(self.opt.device)

    def process(self, images, return_time=False):
        raise NotImplementedError

    def post_process(self, dets, meta, scale=1, fg=None):
        raise NotImplementedError

    def merge_outputs(self, detections):
        raise NotImplementedError

    def run(self, image_or_path_or_tensor, id=1, meta=None):
        load_time = pre_time = net_time = dec_time = post_time = merge_time = 0
        start_time = time.time()
        pre_processed = False
        if isinstance(image_or_path_or_tensor, np.ndarray):
            image = image_or_path_or_tensor
        elif isinstance(image_or_path_or_tensor, str):
            image = np.load(image_or_path_or_tensor)
        else:
            image = image_or_path_or_tensor["image"][0].numpy()
            pre_processed_images = image_or_path_or_tensor
            pre_processed = True
        loaded_time = time.time()
        load_time += loaded_time - start_time

        detections = []
        for scale in self.scales:
            scale_start_time = time.time()
            if not pre_processed:
                images, meta = self.pre_process(image, scale, meta)
            else:
                images = pre_processed_images["images"][scale][0]
                meta = pre_processed_images["meta"][scale]
                meta = {k: v.numpy()[0] for k, v in meta.items()}
            images = images.to(self.opt.device)
            torch.cuda.synchronize()
            pre_process_time = time.time()
            pre_time += pre_process_time - scale_start_time
            output, dets, forward_time = self.process(images, return_time=True)
            torch.cuda.synchronize()
            net_time += forward_time - pre_process_time
            decode_time = time.time()
            dec_time += decode_time - forward_time
            dets = self.post_process(dets, meta, scale)
            post_process_time = time.time()
            post_time += post_process_time - decode_time
            detections.append(dets)

        results = self.merge_outputs(detections)
        end_time = time.time()
        merge_time += end_time - post_process_time
        return {"results": results, "tot": end_time - start_time, "load": load_time,
                "pre": pre_time, "net": net_time, "dec": dec_time, "post": post_time,
                "merge": merge_time}

"""Options for the polydet hot path (reference: src/lib/opts.py).

Same flag names, defaults and derived fields as the reference for everything the
path consumes (SURVEY.md section 5 "Config / flags"); flags of other tasks are
not restated.  Fixed here: the undefined `r_variation` (reference :391-396) is a
real option, `--clip_value` has type float (reference :104), the duplicate
`reg` head update is dropped.
"""
import argparse
import os


def chunk_sizes_for(batch_size, master_batch_size, n):
    """Per-replica share of a global batch (reference: src/lib/opts.py:301-310): replica 0 takes
    `master_batch_size` (default batch_size // n), the rest is spread over the others, the first ones
    taking the remainder."""
    master = batch_size // n if master_batch_size == -1 else master_batch_size
    rest = batch_size - master
    sizes = [master]
    for i in range(n - 1):
        chunk = rest // (n - 1)
        if i < rest % (n - 1):
            chunk += 1
        sizes.append(chunk)
    return sizes


class opts(object):
    def __init__(self):
        p = argparse.ArgumentParser()
        p.add_argument("task", default="polydet", nargs="?", help="polydet")
        p.add_argument("--dataset", default="cityscapes",
                       help="cityscapes | kitti_poly | IDD (annotation JSON + images) | synthetic (offline)")
        p.add_argument("--exp_id", default="default")
        p.add_argument("--test", action="store_true")
        p.add_argument("--debug", type=int, default=0)
        p.add_argument("--load_model", default="")
        p.add_argument("--resume", action="store_true")
        p.add_argument("--gpus", default="0", help="-1 is rejected: no CPU path")
        p.add_argument("--num_workers", type=int, default=4)
        p.add_argument("--not_cuda_benchmark", action="store_true")
        p.add_argument("--seed", type=int, default=317)
        p.add_argument("--print_iter", type=int, default=0)
        p.add_argument("--hide_data_time", action="store_true")
        p.add_argument("--save_all", action="store_true")
        p.add_argument("--metric", default="loss")
        # model
        p.add_argument("--arch", default="dla_34",
                       help="dla_34 | hourglass | smallhourglass")
        p.add_argument("--head_conv", type=int, default=-1)
        p.add_argument("--down_ratio", type=int, default=4)
        p.add_argument("--nbr_points", type=int, default=16)
        p.add_argument("--rep", default="cartesian", choices=["cartesian", "polar", "polar_fixed"])
        p.add_argument("--r_variation", default="none", choices=["none", "one", "two", "four"])
        # input
        p.add_argument("--input_res", type=int, default=-1)
        p.add_argument("--input_h", type=int, default=-1)
        p.add_argument("--input_w", type=int, default=-1)
        # train
        p.add_argument("--lr", type=float, default=4e-6)
        p.add_argument("--lr_step", type=str, default="90,120")
        p.add_argument("--num_epochs", type=int, default=240)
        p.add_argument("--batch_size", type=int, default=32)
        p.add_argument("--master_batch_size", type=int, default=-1)
        p.add_argument("--num_iters", type=int, default=-1)
        p.add_argument("--val_intervals", type=int, default=5)
        p.add_argument("--trainval", action="store_true")
        p.add_argument("--clip", action="store_true")
        p.add_argument("--clip_value", type=float, default=1.0)
        p.add_argument("--root_dir", default="../",
                       help="where exp/<dataset>/<task>/<exp_id> is created (reference: hard-coded ../)")
        p.add_argument("--synthetic_samples", type=int, default=64,
                       help="items per epoch of the synthetic dataset")
        p.add_argument("--device_targets", action="store_true",
                       help="loader workers only pack the raw annotations; heat maps and regression "
                            "targets are built on the GPU (cp_polydet_targets) after the batch upload")
        p.add_argument("--arithmetic", default="split_bf16", choices=["split_bf16", "exact_f32"],
                       help="contraction arithmetic of the convolutions / heads / DCNv2 (centerpoly_amd/arithmetic.py): "
                            "split-bf16 x3 on the bf16 matrix cores (default, ~2^-16 per product) or exact fp32 chains")
        p.add_argument("--no_reorder_flip", action="store_true")
        # sampler augmentation (reference: opts.py "train" group)
        p.add_argument("--not_rand_crop", action="store_true")
        p.add_argument("--shift", type=float, default=0.1)
        p.add_argument("--scale", type=float, default=0.4)
        p.add_argument("--flip", type=float, default=0.5)
        p.add_argument("--no_color_aug", action="store_true")
        p.add_argument("--annot_dir", default="",
                       help="directory of the annotation JSON files (reference: ../<dataset>Stuff/BBoxes)")
        p.add_argument("--img_dir", default="", help="directory of the image files")
        p.add_argument("--bucket_cap_mb", type=int, default=32,
                       help="gradient all-reduce bucket size (RCCL over xGMI)")
        # test
        p.add_argument("--flip_test", action="store_true")
        p.add_argument("--test_scales", type=str, default="1")
        p.add_argument("--nms", action="store_true")
        p.add_argument("--K", type=int, default=128)
        p.add_argument("--thresh", type=float, default=0.05,
                       help="threshold for the outputs kept for evaluation (result writer)")
        p.add_argument("--not_prefetch_test", action="store_true")
        p.add_argument("--fix_res", action="store_true")
        p.add_argument("--keep_res", action="store_true")
        # loss
        p.add_argument("--mse_loss", action="store_true")
        p.add_argument("--reg_loss", default="l1")
        p.add_argument("--poly_loss", default="l1", choices=["l1", "iou", "l1+iou", "relu"])
        p.add_argument("--poly_order", action="store_true")
        p.add_argument("--hm_weight", type=float, default=1)
        p.add_argument("--off_weight", type=float, default=1)
        p.add_argument("--poly_weight", type=float, default=1)
        p.add_argument("--depth_weight", type=float, default=0.1)
        p.add_argument("--wh_weight", type=float, default=0.1)
        # task
        p.add_argument("--not_reg_offset", action="store_true")
        p.add_argument("--cat_spec_poly", action="store_true")
        p.add_argument("--dense_poly", action="store_true")
        for flag in ("eval_oracle_hm", "eval_oracle_border_hm", "eval_oracle_offset",
                     "eval_oracle_poly", "eval_oracle_pseudo_depth"):
            p.add_argument("--" + flag, action="store_true")
        self.parser = p

    def parse(self, args=""):
        opt = self.parser.parse_args() if args == "" else self.parser.parse_args(args)
        opt.gpus_str = opt.gpus
        from . import arithmetic
        arithmetic.configure(opt.arithmetic)
        opt.gpus = [int(g) for g in opt.gpus.split(",")]
        opt.gpus = [i for i in range(len(opt.gpus))] if opt.gpus[0] >= 0 else [-1]
        opt.lr_step = [int(i) for i in opt.lr_step.split(",")]
        opt.test_scales = [float(i) for i in opt.test_scales.split(",")]
        opt.fix_res = not opt.keep_res
        opt.reg_offset = not opt.not_reg_offset
        if opt.head_conv == -1:
            opt.head_conv = 256 if "dla" in opt.arch else 64
        opt.pad = 127 if "hourglass" in opt.arch else 31
        opt.num_stacks = 2 if opt.arch == "hourglass" else 1
        if opt.trainval:
            opt.val_intervals = 100000000
        opt.master_batch_size_arg = opt.master_batch_size
        opt.chunk_sizes = chunk_sizes_for(opt.batch_size, opt.master_batch_size, len(opt.gpus))
        opt.master_batch_size = opt.chunk_sizes[0]
        opt.root_dir = os.path.join(opt.root_dir)
        opt.data_dir = opt.root_dir
        opt.exp_dir = os.path.join(opt.root_dir, "exp", opt.dataset, opt.task)
        opt.save_dir = os.path.join(opt.exp_dir, opt.exp_id)
        opt.debug_dir = os.path.join(opt.save_dir, "debug")
        if opt.resume and opt.load_model == "":
            path = opt.save_dir[:-4] if opt.save_dir.endswith("TEST") else opt.save_dir
            opt.load_model = os.path.join(path, "model_last.pth")
        return opt

    def update_dataset_info_and_set_heads(self, opt, dataset):
        input_h, input_w = dataset.default_resolution
        opt.mean, opt.std = dataset.mean, dataset.std
        opt.num_classes = dataset.num_classes
        input_h = opt.input_res if opt.input_res > 0 else input_h
        input_w = opt.input_res if opt.input_res > 0 else input_w
        opt.input_h = opt.input_h if opt.input_h > 0 else input_h
        opt.input_w = opt.input_w if opt.input_w > 0 else input_w
        opt.output_h = opt.input_h            # literal: the reference does not divide
        opt.output_w = opt.input_w
        opt.input_res = max(opt.input_h, opt.input_w)
        opt.output_res = max(opt.output_h, opt.output_w)
        if opt.task != "polydet":
            raise AssertionError("task not defined!")
        n_poly = opt.nbr_points * 2
        opt.heads = {"hm": opt.num_classes,
                     "poly": n_poly if not opt.cat_spec_poly else n_poly * opt.num_classes,
                     "pseudo_depth": 1}
        extra = {"one": 1, "two": 2, "four": 4}.get(opt.r_variation)
        if opt.reg_offset:
            opt.heads.update({"reg": 2})
        if extra:
            opt.heads.update({"radius": extra})
        print("heads", opt.heads)
        return opt

    def init(self, args=""):
        class Struct:
            def __init__(self, entries):
                for k, v in entries.items():
                    self.__setattr__(k, v)

        opt = self.parse(args)
        info = {"default_resolution": [512, 1024], "num_classes": 8,
                "mean": [0.284, 0.323, 0.282], "std": [0.04, 0.04, 0.04],
                "dataset": "cityscapes"}
        dataset = Struct(info)
        opt.dataset = dataset.dataset
        return self.update_dataset_info_and_set_heads(opt, dataset)

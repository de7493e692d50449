"""PolydetLoss / PolydetTrainer (reference: src/lib/trains/polydet.py:20-237).

Aggregation, weights, returned stats keys and the in-place activation of
output['hm'] follow the reference; every term is a HIP kernel and stays a device
scalar.  `--dense_poly` (:107-110) is a dense masked-L1 kernel; `--cat_spec_poly` (:103-106) hands PolyLoss a
[B, M, C*2N] mask whose rows `if mask[batch][i]:` cannot reduce to a bool -- the reference raises RuntimeError on the first
object (models/losses.py:870) and so does this mirror.  The `--eval_oracle_*` switches (:49-70, numba BFS maps of an
evaluation-protocol experiment, all default off) raise NotImplementedError.
"""
import torch

from ..models.decode import polydet_decode
from ..models.losses import (FocalLoss, MSELoss, PolyLoss, RegL1Loss, RegLoss, dense_poly_l1_loss,
                             sigmoid_focal_loss)
from ..utils.post_process import polydet_post_process
from .base_trainer import BaseTrainer

_UNSUPPORTED = ("eval_oracle_hm", "eval_oracle_border_hm", "eval_oracle_offset",
                "eval_oracle_poly", "eval_oracle_pseudo_depth")


class PolydetLoss(torch.nn.Module):
    def __init__(self, opt):
        super(PolydetLoss, self).__init__()
        for flag in _UNSUPPORTED:
            if getattr(opt, flag, False):
                raise NotImplementedError("--%s is outside the accelerated polydet path" % flag)
        if getattr(opt, "reg_loss", "l1") not in ("l1", "sl1"):
            raise NotImplementedError("--reg_loss must be l1 or sl1 (the reference leaves crit_reg None otherwise)")
        self.mse = bool(getattr(opt, "mse_loss", False))
        self.crit = MSELoss() if self.mse else FocalLoss()
        self.crit_reg = RegL1Loss() if getattr(opt, "reg_loss", "l1") == "l1" else RegLoss()
        self.crit_poly = PolyLoss(opt)
        self.opt = opt

    def forward(self, outputs, batch):
        opt = self.opt
        hm_loss = off_loss = poly_loss = depth_loss = order_loss = 0
        for s in range(opt.num_stacks):
            output = outputs[s]
            depth_loss = depth_loss + self.crit_reg(
                output["pseudo_depth"], batch["reg_mask"], batch["ind"],
                batch["pseudo_depth"]) / opt.num_stacks
            if self.mse:                     # --mse_loss: MSE on the raw head, no activation (:44-46,84)
                hm_l = self.crit(output["hm"], batch["hm"])
            else:                            # _sigmoid + FocalLoss fused; output['hm'] becomes the activated map
                hm_l, output["hm"] = sigmoid_focal_loss(output["hm"], batch["hm"])
            hm_loss = hm_loss + hm_l / opt.num_stacks
            if getattr(opt, "cat_spec_poly", False):
                # :103-106 -> PolyLoss.forward with mask = batch['cat_spec_mask'] [B, M, C*2N]: its per-object test
                # `if mask[batch][i]:` (models/losses.py:870) is the truth value of a C*2N-element tensor
                if batch["cat_spec_mask"][0][0].numel() > 1:
                    raise RuntimeError("Boolean value of Tensor with more than one value is ambiguous")
                r = self.crit_poly(output["poly"], batch["cat_spec_mask"][:, :, 0], batch["ind"], batch["cat_spec_poly"],
                                   hm=output["hm"])
                poly_loss = poly_loss + (r[0] if opt.poly_order else r) / opt.num_stacks
            elif getattr(opt, "dense_poly", False):
                poly_loss = poly_loss + dense_poly_l1_loss(output["poly"], batch["dense_poly"],
                                                           batch["dense_poly_mask"], 1e-4) / opt.num_stacks
            else:
                r = self.crit_poly(output["poly"], batch["reg_mask"], batch["ind"], batch["poly"],
                                   freq_mask=batch.get("freq_mask"), peak=batch.get("peak"),
                                   hm=output["hm"])
                if opt.poly_order:
                    poly_loss = poly_loss + r[0] / opt.num_stacks
                    order_loss = order_loss + r[1] / opt.num_stacks
                else:
                    poly_loss = poly_loss + r / opt.num_stacks
            if opt.reg_offset and opt.off_weight > 0:
                off_loss = off_loss + self.crit_reg(output["reg"], batch["reg_mask"], batch["ind"],
                                                    batch["reg"]) / opt.num_stacks
        if opt.poly_order:
            loss = opt.hm_weight * hm_loss + opt.off_weight * off_loss \
                + opt.poly_weight * (poly_loss + order_loss) + opt.depth_weight * depth_loss
            stats = {"loss": loss, "hm_l": hm_loss, "off_l": off_loss, "poly_l": poly_loss,
                     "order_l": order_loss, "depth_l": depth_loss}
        else:
            loss = opt.hm_weight * hm_loss + opt.off_weight * off_loss \
                + opt.poly_weight * poly_loss + opt.depth_weight * depth_loss
            stats = {"loss": loss, "hm_l": hm_loss, "off_l": off_loss, "poly_l": poly_loss,
                     "depth_l": depth_loss}
        for k, v in stats.items():
            if not torch.is_tensor(v):
                stats[k] = batch["hm"].new_tensor(float(v))
        return loss, stats


class PolydetTrainer(BaseTrainer):
    def __init__(self, opt, model, optimizer=None):
        super(PolydetTrainer, self).__init__(opt, model, optimizer=optimizer)

    def _get_losses(self, opt):
        if opt.task != "polydet":
            raise NotImplementedError
        if opt.poly_order:
            loss_states = ["loss", "hm_l", "off_l", "poly_l", "order_l", "depth_l"]
        else:
            loss_states = ["loss", "hm_l", "off_l", "poly_l", "depth_l"]
        return loss_states, PolydetLoss(opt)

    def prepare_batch(self, batch):
        """--device_targets: the loader delivered packed raw annotations (bbox, poly, cls_id, ...,
        trans_output); build hm / ind / reg / poly / ... on the GPU (the reference does this per
        object in the sampler, src/lib/datasets/sample/polydet.py:160-405)."""
        if "trans_output" not in batch:
            return batch
        from ..datasets.sample.polydet import build_inputs, build_targets
        opt = self.opt
        if "image_u8" in batch:
            # real datasets: the loader delivered 8-bit images + the drawn augmentation; warp, colour
            # augmentation and normalisation run here (sample/polydet.py:106-136 of the reference)
            hw = batch["input_hw"][0].tolist()
            batch["input"] = build_inputs(batch["image_u8"], batch["trans_input"].cpu().numpy(),
                                          batch["color"].cpu().numpy(), opt.mean, opt.std, hw[0], hw[1])
        h, w = batch["input"].shape[2] // opt.down_ratio, batch["input"].shape[3] // opt.down_ratio
        targets = build_targets(batch, h, w, opt.num_classes, rep=opt.rep,
                                no_reorder_flip=getattr(opt, "no_reorder_flip", False),
                                with_border_hm=False, dense_poly=getattr(opt, "dense_poly", False),
                                cat_spec_poly=getattr(opt, "cat_spec_poly", False))
        out = {k: v for k, v in batch.items() if k in ("input", "meta")}
        out.update(targets)
        return out

    def debug(self, batch, output, iter_id):
        raise NotImplementedError("the reference's debug() reads output['wh'], which polydet "
                                  "never produces (trains/polydet.py:185-188)")

    def save_result(self, output, batch, results):
        reg = output["reg"] if self.opt.reg_offset else None
        dets = polydet_decode(output["hm"], output["poly"], output["pseudo_depth"], reg=reg,
                              cat_spec_poly=self.opt.cat_spec_poly, K=self.opt.K, rep=self.opt.rep)
        dets = dets.detach().cpu().numpy().reshape(1, -1, dets.shape[2])
        dets_out = polydet_post_process(
            dets.copy(), batch["meta"]["c"].cpu().numpy(), batch["meta"]["s"].cpu().numpy(),
            output["hm"].shape[2], output["hm"].shape[3], output["hm"].shape[1])
        results[batch["meta"]["img_id"].cpu().numpy()[0]] = dets_out[0]

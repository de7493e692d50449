"""create_model / load_model / save_model (reference: src/lib/models/model.py:14-142).

Only the architectures on the polydet hot path are registered ('dla', 'hourglass',
'smallhourglass'); checkpoint format is the reference's
{'epoch', 'state_dict', ['optimizer']} so its .pth files load unchanged.
"""
import torch

from .networks.large_hourglass import get_large_hourglass_net, get_small_hourglass_net
from .networks.pose_dla_dcn import get_pose_net as get_dla_dcn

_model_factory = {
    "dla": get_dla_dcn,
    "hourglass": get_large_hourglass_net,
    "smallhourglass": get_small_hourglass_net,
}


def create_model(arch, heads, head_conv):
    name, _, layers = arch.partition("_")
    num_layers = int(layers) if layers else 0
    if name not in _model_factory:
        raise KeyError("arch %r is outside the accelerated path (have: %s)"
                       % (arch, ", ".join(sorted(_model_factory))))
    return _model_factory[name](num_layers=num_layers, heads=heads, head_conv=head_conv)


def _strip_module_prefix(sd):
    out = {}
    for k, v in sd.items():
        if k.startswith("module") and not k.startswith("module_list"):
            k = k[7:]
        out[k] = v
    return out


def load_model(model, model_path, optimizer=None, resume=False, lr=None, lr_step=None):
    checkpoint = torch.load(model_path, map_location="cpu")
    print("loaded {}, epoch {}".format(model_path, checkpoint["epoch"]))
    incoming = _strip_module_prefix(checkpoint["state_dict"])
    own = model.state_dict()
    for k in list(incoming):
        if k not in own:
            print("Drop parameter {}.".format(k))
        elif incoming[k].shape != own[k].shape:
            print("Skip loading parameter {}, required shape{}, loaded shape{}.".format(
                k, own[k].shape, incoming[k].shape))
            incoming[k] = own[k]
    for k in own:
        if k not in incoming:
            print("No param {}.".format(k))
            incoming[k] = own[k]
    model.load_state_dict(incoming, strict=False)

    start_epoch = 0
    if optimizer is not None and resume:
        if "optimizer" in checkpoint:
            optimizer.load_state_dict(checkpoint["optimizer"])
            start_epoch = checkpoint["epoch"]
            start_lr = lr
            for step in lr_step:
                if start_epoch >= step:
                    start_lr *= 0.1
            for group in optimizer.param_groups:
                group["lr"] = start_lr
            print("Resumed optimizer with start lr", start_lr)
        else:
            print("No optimizer parameters in checkpoint.")
    if optimizer is not None:
        return model, optimizer, start_epoch
    return model


def save_model(path, epoch, model, optimizer=None):
    if isinstance(model, (torch.nn.DataParallel, torch.nn.parallel.DistributedDataParallel)):
        model = model.module
    data = {"epoch": epoch, "state_dict": model.state_dict()}
    if optimizer is not None:
        data["optimizer"] = optimizer.state_dict()
    torch.save(data, path)

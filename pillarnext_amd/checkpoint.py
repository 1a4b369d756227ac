"""Checkpoint wire format of the reference trainer (trainer/utils/checkpoint.py:8-44 load, :62-89 save): a torch.save'd dict
{"meta", "state_dict"[, "optimizer", "scheduler"]} whose keys may carry DistributedDataParallel's "module." prefix; spconv weight
layouts are mapped onto the dense stand-in by the modules' own _load_from_state_dict (models._SpConv2d)."""
from collections import OrderedDict

import torch


def extract_state_dict(checkpoint):
    if isinstance(checkpoint, OrderedDict):
        sd = checkpoint
    elif isinstance(checkpoint, dict) and "state_dict" in checkpoint:
        sd = checkpoint["state_dict"]
    elif isinstance(checkpoint, dict) and "model" in checkpoint:
        sd = checkpoint["model"]
    else:
        raise RuntimeError("no state_dict in the checkpoint")
    if len(sd) and next(iter(sd)).startswith("module."):
        sd = OrderedDict((k[7:], v) for k, v in sd.items())
    return sd


def load_checkpoint(model, filename, map_location=None, strict=False):
    import os

    if not os.path.isfile(filename):  # trainer/utils/checkpoint.py:21-22
        raise IOError(f"{filename} is not a checkpoint file")
    checkpoint = torch.load(filename, map_location=map_location, weights_only=False)
    target = model.module if hasattr(model, "module") else model
    # files written by save_checkpoint say which layout their sparse-conv weights are in; the shapes alone cannot when Cin == kH == kW
    layout = checkpoint.get("meta", {}).get("pnx_conv_layout") if isinstance(checkpoint, dict) and isinstance(checkpoint.get("meta"), dict) else None
    from .models import _SpConv2d

    convs = [m for m in target.modules() if isinstance(m, _SpConv2d)] if layout in ("spconv", "dense") else []
    for m in convs:
        m.assume_layout = layout
    try:
        missing, unexpected = target.load_state_dict(extract_state_dict(checkpoint), strict=strict)
    finally:
        for m in convs:
            del m.assume_layout
    return checkpoint, missing, unexpected


def save_checkpoint(model, filename, optimizer=None, scheduler=None, meta=None, layout="spconv"):
    """layout="spconv" (default): the weights of the dense stand-ins of SubMConv2d / SparseConv2d (models._SpConv2d) are written in
    spconv >= 2.2's (Cout, kH, kW, Cin) layout, so that the REFERENCE's modules load the file (their load_state_dict checks shapes);
    this package reads either layout back.  layout="dense" keeps nn.Conv2d's (Cout, Cin, kH, kW)."""
    if meta is not None and not isinstance(meta, dict):
        raise TypeError("meta must be a dict or None")
    if layout not in ("spconv", "dense"):
        raise ValueError("layout must be 'spconv' or 'dense'")
    target = model.module if hasattr(model, "module") else model
    sd = OrderedDict((k, v.cpu()) for k, v in target.state_dict().items())
    if layout == "spconv":
        from .models import _SpConv2d

        for name, m in target.named_modules():
            if isinstance(m, _SpConv2d):
                k = (name + "." if name else "") + "weight"
                sd[k] = sd[k].permute(0, 2, 3, 1).contiguous()
    meta = dict(meta or {})
    meta["pnx_conv_layout"] = layout      # (Cout, kH, kW, Cin) and (Cout, Cin, kH, kW) cannot be told apart by shape when Cin == kH == kW
    ck = {"meta": meta, "state_dict": sd}
    if optimizer is not None:
        ck["optimizer"] = optimizer.state_dict()
    if scheduler is not None:
        ck["scheduler"] = scheduler.state_dict()
    torch.save(ck, filename)

"""CPU: B1 surface beyond import paths --
 * an own-written YAML with the reference's key structure (configs/pillarnext_b_nusc.yaml) resolves `${...}` interpolations and
   instantiates through `_target_` into the MI355X modules, with the parameter names of the published checkpoints;
 * a checkpoint shaped like the reference trainer's (dict with "state_dict", DDP "module." prefix, spconv weight layouts,
   num_batches_tracked) round-trips through pillarnext_amd.checkpoint."""
import os

import pytest

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build():
    from pillarnext_amd import config

    cfg = config.load(os.path.join(ROOT, "configs", "pillarnext_b_nusc.yaml"))
    assert cfg["model"]["backbone"]["num_input_features"] == 64                      # ${model.reader.num_filters[1]}
    assert cfg["model"]["head"]["tasks"][1] == ["truck", "construction_vehicle"]     # ${_tasks}
    assert cfg["model"]["post_processing"]["pc_range"] == cfg["model"]["reader"]["pc_range"]
    return cfg, config.instantiate(cfg["model"])


def test_yaml_instantiates_the_detector():
    from pillarnext_amd import models, reader

    cfg, det = build()
    assert isinstance(det, models.SingleStageDetector) and isinstance(det.reader, reader.PillarFeatureNet)
    assert isinstance(det.backbone, models.SparseResNet) and isinstance(det.neck, models.ASPPNeck) and isinstance(det.head, models.CenterHead)
    assert list(det.reader.grid_size) == [1344, 1344] and len(det.head.tasks) == 6
    assert det.post_processing["nms"]["nms_post_max_size"] == 83                      # plain dict, as hydra hands it over
    keys = set(det.state_dict())
    for k in ("reader.pfn_layers.0.linear.weight", "reader.pfn_layers.1.norm.num_batches_tracked", "backbone.blocks.0.0.conv.weight",
              "backbone.blocks.3.2.norm2.running_var", "backbone.mapping.1.weight", "neck.pre_conv.block1.conv.conv.weight", "neck.weight",
              "head.shared_conv.0.weight", "head.tasks.5.hm.3.bias", "head.tasks.0.deblock.conv.conv.weight"):
        assert k in keys, k
    assert sum(p.numel() for p in det.parameters()) > 10_000_000                       # PillarNeXt-B is ~10.4 M parameters


def test_reference_shaped_checkpoint_round_trip(tmp_path):
    from pillarnext_amd import checkpoint

    _, det = build()
    sd = det.state_dict()
    torch.manual_seed(1)
    ck = {}
    for k, v in sd.items():
        w = torch.randn_like(v) if v.is_floating_point() else v + 7
        if k.startswith("backbone.") and k.endswith("conv.weight") and w.dim() == 4 or k == "backbone.mapping.0.weight" or k.endswith("conv2.weight") and k.startswith("backbone."):
            w = w.permute(0, 2, 3, 1).contiguous()                                     # spconv >= 2.2 layout (Cout, kH, kW, Cin)
        ck["module." + k] = w                                                          # saved from a DDP-wrapped model
    path = os.path.join(tmp_path, "epoch_20.pth")
    torch.save({"meta": {"epoch": 20}, "state_dict": ck, "optimizer": {}}, path)
    _, det2 = build()
    loaded, missing, unexpected = checkpoint.load_checkpoint(det2, path, map_location="cpu", strict=True)
    assert loaded["meta"]["epoch"] == 20 and not missing and not unexpected
    sd2 = det2.state_dict()
    assert torch.equal(sd2["reader.pfn_layers.1.norm.num_batches_tracked"], sd["reader.pfn_layers.1.norm.num_batches_tracked"] + 7)
    assert torch.equal(sd2["backbone.blocks.1.0.conv.weight"], ck["module.backbone.blocks.1.0.conv.weight"].permute(0, 3, 1, 2))
    assert torch.equal(sd2["head.tasks.2.hm.3.weight"], ck["module.head.tasks.2.hm.3.weight"])
    # and back out in the reference's format
    out = os.path.join(tmp_path, "resaved.pth")
    checkpoint.save_checkpoint(torch.nn.DataParallel(det2) if False else det2, out, meta={"epoch": 21})
    again = torch.load(out, weights_only=False)
    assert set(again) == {"meta", "state_dict"} and list(again["state_dict"]) == list(sd2)
    assert all(not v.is_cuda for v in again["state_dict"].values())


def test_saved_checkpoint_carries_spconv_layout_and_loads_back(tmp_path):
    """save_checkpoint writes the sparse-conv stand-ins in spconv's (Cout, kH, kW, Cin) layout -- what the reference's modules check on
    load (trainer/utils/checkpoint.py:8-44) -- and this package reads the file back bit for bit."""
    import torch

    from pillarnext_amd import checkpoint
    from pillarnext_amd.models import SparseResNet, _SpConv2d

    torch.manual_seed(3)
    net = SparseResNet([1, 1], [1, 2], [8, 16], 8, kernel_size=(3, 3), out_channels=16)
    f = str(tmp_path / "ck.pth")
    checkpoint.save_checkpoint(net, f, meta={"epoch": 1})
    raw = torch.load(f, weights_only=False)["state_dict"]
    for name, m in net.named_modules():
        if isinstance(m, _SpConv2d):
            co, ci, kh, kw = m.weight.shape
            assert tuple(raw[name + ".weight"].shape) == (co, kh, kw, ci)
    net2 = SparseResNet([1, 1], [1, 2], [8, 16], 8, kernel_size=(3, 3), out_channels=16)
    _, missing, unexpected = checkpoint.load_checkpoint(net2, f, strict=True)
    assert not missing and not unexpected
    for (k, a), (_, b) in zip(net.state_dict().items(), net2.state_dict().items()):
        assert torch.equal(a, b), k
    import pytest

    with pytest.raises(IOError):
        checkpoint.load_checkpoint(net2, str(tmp_path / "nope.pth"))


def test_waymo_yaml_instantiates_with_the_iou_head():
    """configs/pillarnext_b_waymo.yaml = the model block of the reference's waymo_det_pp18_aspp_iou_car_sp.yaml (:11, :43-73): 2 tasks, an
    `iou` branch in every SepHead, IoU-rectified scores, NMS pre 4096 / post 500 / thresholds [[0.7], [0.2, 0.25]], 2048 x 2048 pillars."""
    from pillarnext_amd import config, models

    cfg = config.load(os.path.join(ROOT, "configs", "pillarnext_b_waymo.yaml"))
    det = config.instantiate(cfg["model"])
    assert isinstance(det, models.SingleStageDetector) and len(det.head.tasks) == 2 and det.head.with_iou and det.head.with_reg_iou
    assert list(det.reader.grid_size) == [2048, 2048] and det.head.weight == 1
    assert det.head.rectifier == [[0.68], [0.71, 0.65]] and det.head.num_classes == [1, 2]
    pp = det.post_processing
    assert pp["nms"]["nms_pre_max_size"] == 4096 and pp["nms"]["nms_post_max_size"] == 500 and pp["nms"]["nms_iou_threshold"] == [[0.7], [0.2, 0.25]]
    assert pp["post_center_limit_range"][3] == 80.0
    keys = set(det.state_dict())
    for k in ("head.tasks.0.iou.0.weight", "head.tasks.1.iou.3.bias", "head.tasks.1.hm.3.weight"):
        assert k in keys, k
    assert det.state_dict()["head.tasks.1.hm.3.weight"].shape[0] == 2


def test_checkpoint_records_its_conv_layout(tmp_path):
    """A 3x3 sparse-conv layer over 3 input features: (Cout, Cin, kH, kW) and spconv's (Cout, kH, kW, Cin) have the same shape.  A file
    written by save_checkpoint says which one it holds and loads back exactly in both layouts; a bare state_dict of that shape is refused."""
    import torch

    from pillarnext_amd import checkpoint
    from pillarnext_amd.models import SparseResNet

    torch.manual_seed(1)
    bb = SparseResNet([1], [1], [8], 3)                       # blocks.0.0.conv.weight is (8, 3, 3, 3)
    w = bb.state_dict()["blocks.0.0.conv.weight"].clone()
    assert tuple(w.shape) == (8, 3, 3, 3) and not torch.equal(w, w.permute(0, 2, 3, 1))
    for layout in ("spconv", "dense"):
        f = str(tmp_path / f"{layout}.pth")
        checkpoint.save_checkpoint(bb, f, layout=layout)
        ck = torch.load(f, weights_only=False)
        assert ck["meta"]["pnx_conv_layout"] == layout
        stored = ck["state_dict"]["blocks.0.0.conv.weight"]
        assert torch.equal(stored, w.permute(0, 2, 3, 1) if layout == "spconv" else w)
        bb2 = SparseResNet([1], [1], [8], 3)
        checkpoint.load_checkpoint(bb2, f)
        assert torch.equal(bb2.state_dict()["blocks.0.0.conv.weight"], w)
        assert not hasattr(bb2.blocks[0][0].conv, "assume_layout")
    with pytest.raises(RuntimeError, match="layout"):
        SparseResNet([1], [1], [8], 3).load_state_dict(bb.state_dict())

"""CPU: every `_target_` the reference's PillarNeXt YAMLs name resolves through the det3d alias package, and the classes
accept the YAML's constructor keys (configs/models/**, configs/experiments/nusc_det_pp18_aspp_iou_sp.yaml)."""
import importlib

import pytest

TARGETS = [
    "det3d.models.readers.pillar_encoder.PillarFeatureNet",
    "det3d.models.backbones.sparse_resnet.SparseResNet",
    "det3d.models.necks.aspp.ASPPNeck",
    "det3d.models.heads.centerhead.CenterHead",
    "det3d.models.detectors.single_stage.SingleStageDetector",
    "det3d.models.readers.mvf_encoder.MVFFeatureNet",          # configs/models/reader/mvf_encoder.yaml
    "det3d.models.readers.voxel_encoder.VoxelFeatureNet",      # configs/models/reader/voxel_encoder.yaml
]


def resolve(t):
    mod, name = t.rsplit(".", 1)
    return getattr(importlib.import_module(mod), name)


@pytest.mark.parametrize("target", TARGETS)
def test_target_resolves(target):
    assert callable(resolve(target))


def test_constructors_take_the_yaml_keys():
    reader = resolve(TARGETS[0])(num_input_features=5, num_filters=[64, 64], voxel_size=[0.075, 0.075, 8], pc_range=[-50.4, -50.4, -5.0, 50.4, 50.4, 3.0])
    assert list(reader.state_dict()) [:2] == ["pfn_layers.0.linear.weight", "pfn_layers.0.norm.weight"]
    assert list(reader.grid_size) == [1344, 1344]
    backbone = resolve(TARGETS[1])(layer_nums=[2, 2, 2, 2], ds_layer_strides=[1, 2, 2, 2], ds_num_filters=[64, 128, 256, 256], num_input_features=64)
    keys = list(backbone.state_dict())
    assert "blocks.0.0.conv.weight" in keys and "blocks.0.1.block1.conv.weight" in keys and "blocks.3.2.norm2.running_var" in keys and "mapping.0.weight" in keys
    neck = resolve(TARGETS[2])(in_channels=256)
    assert {"weight", "conv1x1.weight", "pre_conv.block1.conv.conv.weight", "post_conv.norm.weight"} <= set(neck.state_dict())
    tasks = [["car"], ["truck", "construction_vehicle"]]
    head = resolve(TARGETS[3])(in_channels=256, tasks=tasks, weight=0.25, code_weights=[1.0] * 10,
                              common_heads={"reg": [2, 2], "height": [1, 2], "dim": [3, 2], "rot": [2, 2], "vel": [2, 2]}, strides=[2, 2],
                              rectifier=[[0.5], [0.5, 0.5]], with_reg_iou=True, voxel_size=[0.075, 0.075, 8],
                              pc_range=[-50.4, -50.4, -5.0, 50.4, 50.4, 3.0], out_size_factor=[4, 4])
    hk = set(head.state_dict())
    assert {"shared_conv.0.weight", "tasks.0.deblock.conv.conv.weight", "tasks.1.hm.3.bias", "tasks.0.reg.0.weight"} <= hk
    det = resolve(TARGETS[4])(reader=reader, backbone=backbone, neck=neck, head=head, post_processing={}, sync_batchnorm=True)
    assert det.reader is reader


def test_spconv_checkpoint_layouts_load():
    """spconv stores (Cout,kH,kW,Cin) (>=2.2) or (kH,kW,Cin,Cout): both must load into the dense stand-in."""
    import torch

    bb = resolve(TARGETS[1])(layer_nums=[1, 1], ds_layer_strides=[1, 2], ds_num_filters=[8, 16], num_input_features=4)
    sd = bb.state_dict()
    w = sd["blocks.1.0.conv.weight"].clone()                      # (16, 8, 3, 3)
    for layout in (w.permute(0, 2, 3, 1).contiguous(), w.permute(2, 3, 1, 0).contiguous()):
        sd2 = dict(sd)
        sd2["blocks.1.0.conv.weight"] = layout
        bb.load_state_dict(sd2)
        assert torch.equal(bb.state_dict()["blocks.1.0.conv.weight"], w)

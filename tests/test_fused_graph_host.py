"""Host side of models.FusedPillarNeXt, checked on the CPU (no kernel runs): which layers go to which kernel, BN folding against
the module it replaces, and the block-diagonal weight of the merged SepHead (the GPU tests compare the graphs' outputs)."""
import pytest

torch = pytest.importorskip("torch")


def _detector():
    from pillarnext_amd import synth
    from pillarnext_amd.models import build_pillarnext_b

    cfg = synth.CONFIGS["C1"]
    torch.manual_seed(5)
    det = build_pillarnext_b(cfg["pc_range"], cfg["voxel_size"], tasks=[["car"], ["truck", "bus"]], with_iou_head=True).eval()
    with torch.no_grad():
        for m in det.modules():
            if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
                m.running_mean.uniform_(-0.2, 0.2)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.uniform_(0.6, 1.4)
                m.bias.uniform_(-0.2, 0.2)
    return det


def test_kernel_routing_and_workspace_policy():
    from pillarnext_amd import models

    fused = models.FusedPillarNeXt(_detector(), hip_conv=True)
    kinds = [[type(m).__name__ for m in st] for st in fused.stages]
    assert all(k == "_HipConv3x3" for k in kinds[0] + kinds[1]), kinds          # 64- and 128-channel stages: HIP kernels
    assert all(k == "_HipConv3x3" for k in kinds[2] + kinds[3]), kinds          # 256-channel stages, strided entry convs included
    assert [type(m).__name__ for m in fused.task_conv1] == ["_HipConv3x3"] * 2   # merged SepHead conv 64 -> 64 * branches
    assert [type(m).__name__ for m in fused.task_conv2] == ["_HipSepHeadOut"] * 2
    assert fused.task_chans == [16, 16]
    # sparse workspaces only for all-HIP stages, and only when enabled
    mask = torch.zeros((2, 8, 8), dtype=torch.uint8)
    ws2 = fused._stage_workspace(2, fused.stages[2], mask)
    assert len(ws2) == 3 and tuple(ws2[0][0].shape) == (2, 256, 8, 8)           # for the four HIP convs of the stage
    fused.sparse_ws = False
    assert fused._stage_workspace(0, fused.stages[0], mask) is None
    off = models.FusedPillarNeXt(_detector(), hip_conv=False)
    assert all(type(m).__name__ == "_FusedConv" for st in off.stages for m in st)
    assert all(type(m).__name__ == "_FusedConv" for m in list(off.task_conv1) + list(off.task_conv2))


def test_bn_fold_equals_conv_then_bn():
    from pillarnext_amd.models import _fold_bn

    torch.manual_seed(0)
    conv = torch.nn.Conv2d(8, 12, 3, padding=1, bias=True)
    bn = torch.nn.BatchNorm2d(12, eps=1e-3).eval()
    with torch.no_grad():
        bn.running_mean.uniform_(-0.5, 0.5)
        bn.running_var.uniform_(0.3, 2.0)
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.3, 0.3)
    x = torch.randn(2, 8, 9, 7)
    w, b = _fold_bn(conv.weight, bn, conv.bias)
    torch.testing.assert_close(torch.nn.functional.conv2d(x, w, b, 1, 1), bn(conv(x)), rtol=1e-5, atol=1e-5)
    # transposed convolution (the head's deblock): output channels sit on dim 1 of the weight
    dc = torch.nn.ConvTranspose2d(8, 12, 2, stride=2, bias=False)
    w, b = _fold_bn(dc.weight, bn, transposed=True)
    torch.testing.assert_close(torch.nn.functional.conv_transpose2d(x, w, b, 2), bn(dc(x)), rtol=1e-5, atol=1e-5)


def test_merged_sephead_weights_are_block_diagonal():
    """Branch j of the merged output convolution may only see its own 64 channels; unused outputs have zero weight and bias."""
    from pillarnext_amd import models

    det = _detector()
    fused = models.FusedPillarNeXt(det, hip_conv=False, dtype=torch.float32)     # _FusedConv keeps the dense weights around
    for t, task in enumerate(det.head.tasks):
        names, outs = fused.task_split[t]
        W2, B2 = fused.task_conv2[t].weight, fused.task_conv2[t].bias
        o = 0
        for j, (nme, k) in enumerate(zip(names, outs)):
            fc = getattr(task, nme)
            blk = W2[o:o + k]
            torch.testing.assert_close(blk[:, j * 64:(j + 1) * 64], fc[3].weight.detach().float())
            rest = torch.cat([blk[:, :j * 64], blk[:, (j + 1) * 64:]], 1)
            assert float(rest.abs().max()) == 0.0
            torch.testing.assert_close(B2[o:o + k], fc[3].bias.detach().float())
            o += k
        assert float(W2[o:].abs().max()) == 0.0 and float(B2[o:].abs().max()) == 0.0


def test_aspp_fold_equals_the_module():
    """post_conv distributed over the six ASPP branches (models.fold_aspp) against ASPPNeck itself (aspp.py:19-32), fp64 on the CPU."""
    import torch.nn.functional as F

    from pillarnext_amd.models import ASPPNeck, fold_aspp

    torch.manual_seed(3)
    nk = ASPPNeck(16).eval()
    with torch.no_grad():
        for m in nk.modules():
            if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
                m.running_mean.uniform_(-0.2, 0.2)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.uniform_(0.6, 1.4)
                m.bias.uniform_(-0.2, 0.2)
        nk.weight.mul_(0.1)
    nk = nk.double()
    x = torch.randn(2, 16, 40, 44, dtype=torch.float64)
    with torch.no_grad():
        want = nk(x)
        xin = nk.pre_conv(x)
        wds, shift = fold_aspp(nk)
        got = 0
        for wd, d in zip(wds, (1, 6, 12, 18)):
            got = got + F.conv2d(xin, wd.double(), None, 1, d, d)
        got = torch.relu(got + shift.double().view(1, -1, 1, 1))
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)   # the fold itself is computed in fp32

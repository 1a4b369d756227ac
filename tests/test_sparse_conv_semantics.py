"""The masked-dense backbone blocks (models.SparseConvBlock / SparseBasicBlock / SparseResNet) against a rulebook restatement of spconv's
SubMConv2d / SparseConv2d semantics (oracle/sparse_conv_ref.py) -- spconv itself is absent and unpinned (docker/Dockerfile:18), so this
checks the H2 rule "SubM -> conv * mask, strided sparse conv -> mask_out = maxpool(mask)", not spconv's binaries.  CPU, fp64."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
from oracle import sparse_conv_ref as ref  # noqa: E402  (tests/conftest.py puts the repo root on sys.path)


def _randomise_bn(mod, g):
    with torch.no_grad():
        for m in mod.modules():
            if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
                m.running_mean.copy_(torch.rand(m.running_mean.shape, generator=g) * 0.4 - 0.2)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
                m.weight.copy_(torch.rand(m.weight.shape, generator=g) * 0.8 + 0.6)
                m.bias.copy_(torch.rand(m.bias.shape, generator=g) * 0.4 - 0.2)


def _bn(norm):
    return dict(mean=norm.running_mean.numpy(), var=norm.running_var.numpy(), gamma=norm.weight.detach().numpy(), beta=norm.bias.detach().numpy(),
                eps=norm.eps)


def _sparse_input(g, B, C, H, W, density):
    mask = (torch.rand((B, 1, H, W), generator=g) < density).double()
    x = torch.randn((B, C, H, W), generator=g, dtype=torch.float64) * mask
    idx = torch.nonzero(mask[:, 0] > 0)                       # (N,3) [b,y,x], sorted
    x[idx[0, 0], :, idx[0, 1], idx[0, 2]] = 0.0                # an ACTIVE site whose features are all zero stays active
    feats = x[idx[:, 0], :, idx[:, 1], idx[:, 2]]
    return x, mask, idx.numpy(), feats.numpy()


def _dense_to_sites(y, mask):
    idx = torch.nonzero(mask[:, 0] > 0)
    return idx.numpy(), y[idx[:, 0], :, idx[:, 1], idx[:, 2]].detach().numpy()


@pytest.mark.parametrize("stride,subm", [(1, True), (1, False), (2, False)])
def test_conv_block_equals_the_rulebook(stride, subm):
    from pillarnext_amd.models import SparseConvBlock

    g = torch.Generator().manual_seed(stride * 2 + subm)
    blk = SparseConvBlock(5, 7, 3, stride, use_subm=subm).double().eval()
    _randomise_bn(blk, g)
    x, mask, idx, feats = _sparse_input(g, 2, 5, 13, 18, 0.15)
    with torch.no_grad():
        y, mask_out = blk(x, mask)
    p = dict(weight=blk.conv.weight.detach().numpy(), **_bn(blk.norm))
    want_idx, want, hw = ref.conv_block(idx, feats, (13, 18), p, stride, subm)
    got_idx, got = _dense_to_sites(y, mask_out)
    assert tuple(y.shape[2:]) == hw
    assert np.array_equal(got_idx, want_idx)                                   # same active set, site by site
    np.testing.assert_allclose(got, want, rtol=1e-10, atol=1e-10)
    assert float((y * (1 - mask_out)).abs().max()) == 0.0                      # exact zeros everywhere else


def test_basic_block_and_a_small_resnet_equal_the_rulebook():
    from pillarnext_amd.models import SparseBasicBlock, SparseResNet

    g = torch.Generator().manual_seed(9)
    blk = SparseBasicBlock(6, 3).double().eval()
    _randomise_bn(blk, g)
    x, mask, idx, feats = _sparse_input(g, 2, 6, 11, 12, 0.2)
    with torch.no_grad():
        y, _ = blk(x, mask)
    p1 = dict(weight=blk.block1.conv.weight.detach().numpy(), **_bn(blk.block1.norm))
    p2 = dict(weight=blk.conv2.weight.detach().numpy(), **_bn(blk.norm2))
    want = ref.basic_block(idx, feats, (11, 12), p1, p2)
    np.testing.assert_allclose(_dense_to_sites(y, mask)[1], want, rtol=1e-10, atol=1e-10)

    # two stages of the reference's SparseResNet (sparse_resnet.py:50-68): non-SubM entry conv (stride 1, then 2) + one residual block each, 1x1 mapping
    net = SparseResNet([1, 1], [1, 2], [6, 8], 4, kernel_size=(3, 3), out_channels=5).double().eval()
    _randomise_bn(net, g)
    x, mask, idx, feats = _sparse_input(g, 2, 4, 14, 16, 0.08)
    with torch.no_grad():
        y = net.forward_dense(x, mask)
    hw = (14, 16)
    for si, blkseq in enumerate(net.blocks):
        first, res = blkseq[0], blkseq[1]
        idx, feats, hw = ref.conv_block(idx, feats, hw, dict(weight=first.conv.weight.detach().numpy(), **_bn(first.norm)), first.stride, False)
        feats = ref.basic_block(idx, feats, hw, dict(weight=res.block1.conv.weight.detach().numpy(), **_bn(res.block1.norm)),
                                dict(weight=res.conv2.weight.detach().numpy(), **_bn(res.norm2)))
    wmap = net.mapping[0].weight.detach().numpy()[:, :, 0, 0]
    want = np.maximum(ref.bn_eval(feats @ wmap.T, **_bn(net.mapping[1])), 0)
    assert tuple(y.shape[2:]) == hw
    dense = np.zeros((2, 5) + hw)
    dense[idx[:, 0], :, idx[:, 1], idx[:, 2]] = want
    np.testing.assert_allclose(y.numpy(), dense, rtol=1e-9, atol=1e-9)        # values at the active sites AND zeros elsewhere

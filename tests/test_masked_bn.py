"""CPU: MaskedBatchNorm (the dense stand-in for the reference's BatchNorm1d over the features of a SparseConvTensor,
det3d/models/utils/sparse_conv.py:31-37,57-60) against an independent statement: torch.nn.BatchNorm1d applied to the GATHERED
active sites -- train mode (batch statistics, running-statistics update with momentum 0.01 and the unbiased variance) and eval."""
import torch

from pillarnext_amd.models import MaskedBatchNorm


def _pair(C=12):
    torch.manual_seed(0)
    a = MaskedBatchNorm(C, eps=1e-3, momentum=0.01)
    b = torch.nn.BatchNorm1d(C, eps=1e-3, momentum=0.01)
    with torch.no_grad():
        for m in (a, b):
            m.weight.copy_(torch.linspace(0.5, 1.5, C))
            m.bias.copy_(torch.linspace(-0.3, 0.3, C))
            m.running_mean.copy_(torch.linspace(-1, 1, C))
            m.running_var.copy_(torch.linspace(0.5, 2.0, C))
    return a, b


def test_train_mode_equals_batchnorm1d_over_active_sites():
    a, b = _pair()
    a.train(), b.train()
    g = torch.Generator().manual_seed(3)
    x = torch.randn((3, 12, 9, 11), generator=g, requires_grad=True)
    mask = (torch.rand((3, 1, 9, 11), generator=g) < 0.35).float()
    y = a(x, mask)
    sel = mask[:, 0].bool()                                    # (B,H,W)
    feats = x.detach().permute(0, 2, 3, 1)[sel].clone().requires_grad_(True)   # (n_active, C): what spconv's .features holds
    yb = b(feats)
    torch.testing.assert_close(y.permute(0, 2, 3, 1)[sel], yb, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(a.running_mean, b.running_mean, rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(a.running_var, b.running_var, rtol=1e-6, atol=1e-7)
    assert int(a.num_batches_tracked) == 1
    # gradients: only active sites feed the statistics; inactive sites are zeroed by the block's `* mask` afterwards
    w = torch.randn(yb.shape, generator=g)
    (y.permute(0, 2, 3, 1)[sel] * w).sum().backward()
    (yb * w).sum().backward()
    torch.testing.assert_close(x.grad.permute(0, 2, 3, 1)[sel], feats.grad, rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(a.weight.grad, b.weight.grad, rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(a.bias.grad, b.bias.grad, rtol=1e-4, atol=1e-6)


def test_eval_mode_is_the_running_stat_affine():
    a, b = _pair()
    a.eval(), b.eval()
    x = torch.randn((2, 12, 5, 7))
    mask = (torch.rand((2, 1, 5, 7)) < 0.5).float()
    sel = mask[:, 0].bool()
    torch.testing.assert_close(a(x, mask).permute(0, 2, 3, 1)[sel], b(x.permute(0, 2, 3, 1)[sel]), rtol=1e-6, atol=1e-6)

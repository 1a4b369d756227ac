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


def test_fused_bn_act_node_equals_the_module_composition():
    """models.masked_bn_act (one autograd node: masked BN + residual + ReLU + mask, recomputing in the backward) against the composition
    of MaskedBatchNorm, add, relu and the mask product: outputs, running statistics, and the gradients of x, residual, gamma, beta."""
    import torch.nn.functional as F

    from pillarnext_amd.models import masked_bn_act

    for with_res in (False, True):
        a, _ = _pair()
        b, _ = _pair()
        a.train(), b.train()
        g = torch.Generator().manual_seed(11)
        x1 = torch.randn((3, 12, 9, 11), generator=g, requires_grad=True)
        r1 = torch.randn((3, 12, 9, 11), generator=g, requires_grad=True) if with_res else None
        mask = (torch.rand((3, 1, 9, 11), generator=g) < 0.4).float()
        x2 = x1.detach().clone().requires_grad_(True)
        r2 = r1.detach().clone().requires_grad_(True) if with_res else None
        y1 = masked_bn_act(x1, mask, a, residual=r1)
        pre = b(x2, mask)
        y2 = F.relu(pre + r2 if with_res else pre) * mask
        torch.testing.assert_close(y1, y2, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(a.running_mean, b.running_mean, rtol=1e-6, atol=1e-7)
        torch.testing.assert_close(a.running_var, b.running_var, rtol=1e-6, atol=1e-7)
        w = torch.randn(y1.shape, generator=g)
        (y1 * w).sum().backward()
        (y2 * w).sum().backward()
        torch.testing.assert_close(x1.grad, x2.grad, rtol=1e-4, atol=1e-6)
        torch.testing.assert_close(a.weight.grad, b.weight.grad, rtol=1e-4, atol=1e-6)
        torch.testing.assert_close(a.bias.grad, b.bias.grad, rtol=1e-4, atol=1e-6)
        if with_res:
            torch.testing.assert_close(r1.grad, r2.grad, rtol=1e-5, atol=1e-7)

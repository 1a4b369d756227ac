"""GPU: the fused masked BatchNorm + residual + ReLU + mask kernels (csrc/masked_bn.hip behind models.masked_bn_act) against the torch
statement of the same autograd node (PNX_MASKED_BN_HIP=0), which tests/test_masked_bn.py pins to BatchNorm1d over the gathered active sites:
outputs, running statistics, gradients of x, the residual, gamma and beta; fp32 and bf16 channels_last maps; inactive sites exactly zero."""
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _norm(C):
    from pillarnext_amd.models import MaskedBatchNorm

    n = MaskedBatchNorm(C, eps=1e-3, momentum=0.01).cuda().train()
    with torch.no_grad():
        n.weight.copy_(torch.linspace(0.5, 1.5, C))
        n.bias.copy_(torch.linspace(-0.3, 0.3, C))
        n.running_mean.copy_(torch.linspace(-1, 1, C))
        n.running_var.copy_(torch.linspace(0.5, 2.0, C))
    return n


@pytest.mark.parametrize("C,shape", [(64, (3, 37, 53)), (128, (2, 20, 31)), (256, (2, 9, 11)), (16, (1, 8, 8))])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("with_res,relu", [(False, True), (True, True), (True, False)])
def test_fused_masked_bn_matches_the_torch_node(monkeypatch, C, shape, dtype, with_res, relu):
    from pillarnext_amd.models import masked_bn_act

    B, H, W = shape
    g = torch.Generator(device="cuda").manual_seed(C + H)
    mask = (torch.rand((B, 1, H, W), device="cuda", generator=g) < 0.3).float()
    mask[0, 0, 0, :3] = 1.0
    base = (torch.randn((B, C, H, W), device="cuda", generator=g) * 1.7 + 0.4).to(dtype).contiguous(memory_format=torch.channels_last)
    rbase = torch.randn((B, C, H, W), device="cuda", generator=g).to(dtype).contiguous(memory_format=torch.channels_last)
    wgt = torch.randn((B, C, H, W), device="cuda", generator=g).to(dtype).contiguous(memory_format=torch.channels_last)
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("PNX_MASKED_BN_HIP", mode)
        n = _norm(C)
        x = base.clone().requires_grad_(True)
        r = rbase.clone().requires_grad_(True) if with_res else None
        y = masked_bn_act(x, mask, n, residual=r, relu=relu)
        (y.float() * wgt.float()).sum().backward()
        res[mode] = (y.detach().float(), x.grad.float(), r.grad.float() if with_res else None, n.weight.grad.clone(), n.bias.grad.clone(),
                     n.running_mean.clone(), n.running_var.clone(), int(n.num_batches_tracked))
    a, b = res["1"], res["0"]
    lo = dtype == torch.bfloat16
    off = (mask == 0).expand(B, C, H, W)
    assert bool((a[0][off] == 0).all()) and bool((a[1][off] == 0).all())
    torch.testing.assert_close(a[0], b[0], rtol=1.6e-2 if lo else 1e-4, atol=1.6e-2 if lo else 1e-4)
    torch.testing.assert_close(a[1], b[1], rtol=3e-2 if lo else 2e-4, atol=3e-2 if lo else 2e-4)
    if with_res:
        torch.testing.assert_close(a[2], b[2], rtol=1.6e-2 if lo else 1e-5, atol=1.6e-2 if lo else 1e-5)
    scale = float(b[3].abs().max()) + 1e-6
    torch.testing.assert_close(a[3], b[3], rtol=2e-2 if lo else 1e-3, atol=(2e-2 if lo else 1e-3) * scale)
    torch.testing.assert_close(a[4], b[4], rtol=2e-2 if lo else 1e-3, atol=(2e-2 if lo else 1e-3) * (float(b[4].abs().max()) + 1e-6))
    torch.testing.assert_close(a[5], b[5], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(a[6], b[6], rtol=1e-4, atol=1e-5)
    assert a[7] == b[7] == 1


@pytest.mark.parametrize("dtype,c,shape", [(torch.float32, 64, (2, 33, 41)), (torch.bfloat16, 64, (2, 40, 24)), (torch.float32, 256, (1, 9, 17))])
def test_dense_batchnorm_relu_on_the_masked_node(dtype, c, shape):
    """models.dense_bn_act: nn.BatchNorm2d + ReLU of the head / neck in training on the masked node with every site active, against the modules themselves
    (output, input / weight / bias gradients, running statistics, num_batches_tracked)."""
    import copy

    import torch.nn.functional as F

    from pillarnext_amd.models import dense_bn_act

    B, H, W = shape
    gen = torch.Generator(device="cuda").manual_seed(c + H)
    bn = torch.nn.BatchNorm2d(c).cuda().train()
    with torch.no_grad():
        bn.weight.copy_(torch.rand(c, device="cuda", generator=gen) + 0.5)
        bn.bias.copy_(torch.randn(c, device="cuda", generator=gen) * 0.3)
        bn.running_mean.copy_(torch.randn(c, device="cuda", generator=gen) * 0.1)
    ref = copy.deepcopy(bn).double()     # the modules in fp64 (MIOpen's fp32 BatchNorm backward is itself 2e-4 off on these shapes, and segfaults on channels_last fp32 input)
    x0 = (torch.randn((B, c, H, W), device="cuda", generator=gen) * 2 + 0.5).to(dtype).contiguous(memory_format=torch.channels_last)
    g0 = torch.randn((B, c, H, W), device="cuda", generator=gen).to(dtype).contiguous(memory_format=torch.channels_last)
    x = x0.clone().requires_grad_(True)
    y = dense_bn_act(bn, x)
    assert type(y.grad_fn).__name__.startswith("_MaskedBNActFn") and y.dtype == dtype
    y.backward(g0)
    xr = x0.double().contiguous().requires_grad_(True)
    yr = F.relu(ref(xr))
    yr.backward(g0.double().contiguous())
    tol = 2e-2 if dtype == torch.bfloat16 else 2e-5
    assert float((y.double() - yr.detach()).abs().max()) <= tol * max(1.0, float(yr.detach().abs().max()))
    assert float((x.grad.double() - xr.grad).abs().max()) <= tol * max(1.0, float(xr.grad.abs().max()))
    for a, r in ((bn.weight.grad, ref.weight.grad), (bn.bias.grad, ref.bias.grad)):
        assert float((a.double() - r).abs().max()) <= (3e-2 if dtype == torch.bfloat16 else 1e-4) * max(1.0, float(r.abs().max()))
    assert torch.allclose(bn.running_mean.double(), ref.running_mean, rtol=1e-5, atol=1e-6) and torch.allclose(bn.running_var.double(), ref.running_var, rtol=1e-5, atol=1e-6)
    assert int(bn.num_batches_tracked) == int(ref.num_batches_tracked) == 1
    bn.eval()
    assert not type(dense_bn_act(bn, x).grad_fn).__name__.startswith("_MaskedBNActFn")

"""GPU: the training-mode masked 3x3 convolution node (models._MaskedConv3x3Fn: forward, dgrad and the stride-1 wgrad on the product's HIP
kernels, the stride-2 backward on MIOpen) against torch's own autograd of F.conv2d in FP32 on the same bf16-rounded operands, on LiDAR-like masks (sparse_conv.py:16-63:
SubMConv2d / SparseConv2d compute at the active sites only).  bf16 tolerances: outputs and input gradients to 2 bf16 ulps of a K = 9*Cin
fp32 sum, weight gradients relative to the largest entry."""
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _lidar_mask(B, H, W, gen, p=0.12):
    """clustered occupancy: a few dense blobs + ring-like rows, ~p of the cells"""
    m = torch.rand((B, 1, H, W), device="cuda", generator=gen) < p * 0.3
    yy, xx = torch.meshgrid(torch.arange(H, device="cuda"), torch.arange(W, device="cuda"), indexing="ij")
    for b in range(B):
        for _ in range(6):
            cy, cx = (torch.rand(2, device="cuda", generator=gen) * torch.tensor([H, W], device="cuda")).tolist()
            r = 3 + 0.12 * min(H, W) * float(torch.rand(1, device="cuda", generator=gen))
            m[b, 0] |= ((yy - cy) ** 2 + (xx - cx) ** 2 < r * r) & (torch.rand((H, W), device="cuda", generator=gen) < 0.6)
    return m.float()


@pytest.mark.parametrize("cin,cout,stride,subm,shape", [(64, 64, 1, True, (2, 70, 97)), (64, 64, 1, False, (2, 48, 64)), (128, 128, 1, True, (2, 41, 70)),
                                                       (256, 256, 1, True, (2, 23, 33)), (64, 128, 2, False, (2, 50, 66)), (128, 256, 2, False, (1, 37, 41)),
                                                       (256, 256, 2, False, (2, 24, 64)), (64, 128, 2, False, (1, 130, 7))])
def test_masked_conv_node_matches_fp32_autograd(cin, cout, stride, subm, shape):
    import torch.nn.functional as F

    from pillarnext_amd.models import _SpConv2d, masked_conv

    B, H, W = shape
    gen = torch.Generator(device="cuda").manual_seed(cin + cout + H)
    mask_in = _lidar_mask(B, H, W, gen)
    mask_out = mask_in if subm else F.max_pool2d(mask_in, 3, stride, 1)
    conv = _SpConv2d(cin, cout, 3, stride=stride, padding=1, bias=False).cuda().train()
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, device="cuda", generator=gen) * (2.0 / (9 * cin)) ** 0.5)
    xb = (torch.randn((B, cin, H, W), device="cuda", generator=gen) * mask_in).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    Ho, Wo = mask_out.shape[2:]
    gb = (torch.randn((B, cout, Ho, Wo), device="cuda", generator=gen) * mask_out).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)

    x = xb.clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = masked_conv(conv, x, mask_out, mask_in)
    assert y.dtype == torch.bfloat16 and type(y.grad_fn).__name__.startswith("_MaskedConv3x3Fn")   # the HIP node, not conv(x)
    y.backward(gb)
    dw, dx = conv.weight.grad.clone(), x.grad.float()
    conv.weight.grad = None

    # reference: fp32 autograd of mask_out * conv2d on the same bf16-rounded x, W and upstream gradient
    xr = xb.float().contiguous().requires_grad_(True)
    wr = conv.weight.detach().to(torch.bfloat16).float().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, stride, 1) * mask_out
    yr.backward(gb.float())
    ulp = 2.0 ** -8
    # stride 1: two bf16 ulps of the fp32 sum; the strided kernel sums its input slabs in another order: the bound of tests/test_gpu_dense_ops.py
    tol_y = (2 * ulp * yr.detach().abs().clamp(min=1e-2) + 2e-3) if stride == 1 else (1.6e-2 * yr.detach().abs() + 2e-2)
    err_y = (y.float() - yr.detach()).abs()
    assert bool((err_y <= tol_y).all()), (float(err_y.max()), float((err_y / yr.detach().abs().clamp(min=1e-2)).max()))
    assert bool((y.float()[(mask_out == 0).expand_as(y)] == 0).all())
    dxr = xr.grad * mask_in          # an inactive input site is a constant zero: its gradient is thrown away by the previous layer's mask
    dxm = dx * mask_in
    assert bool(((dxm - dxr).abs() <= 2 * ulp * dxr.abs().clamp(min=1e-2) + 4e-3).all())
    assert bool((dx[(mask_in == 0).expand_as(dx)] == 0).all())       # stride 2 as well since round 6 (csrc/conv_dgrad_s2.h writes zeros at inactive input sites)
    scale = float(wr.grad.abs().max())
    assert float((dw.float() - wr.grad).abs().max()) <= 2e-2 * scale


@pytest.mark.parametrize("cin,cout,stride,subm,shape", [(64, 64, 1, True, (2, 70, 97)), (64, 64, 1, False, (2, 48, 64)), (128, 128, 1, True, (2, 41, 70)),
                                                       (256, 256, 1, True, (2, 23, 33)), (64, 128, 2, False, (2, 50, 66)), (128, 256, 2, False, (1, 37, 41)),
                                                       (256, 256, 2, False, (2, 24, 64)), (64, 64, 1, True, (1, 16, 32)), (128, 128, 1, True, (1, 8, 33))])
def test_fp32_node_on_three_bf16_products_against_fp64(cin, cout, stride, subm, shape):
    """models._MaskedConv3x3F32Fn (pnx_split_f32 + pnx_conv3x3_x3 + three wgrad calls) against the fp64 autograd of mask_out * conv2d on the SAME fp32
    operands.  Tolerance: 2^-14 of the sum of |terms| (each product carries 2^-16 from the dropped low x low term and the halves' rounding; the bound is
    on the absolute sum because cancellation does not shrink the error), and MIOpen's own fp32 result must not be more than 64 x closer on average."""
    import torch.nn.functional as F

    from pillarnext_amd.models import _SpConv2d, masked_conv

    B, H, W = shape
    gen = torch.Generator(device="cuda").manual_seed(3 * cin + cout + H)
    mask_in = _lidar_mask(B, H, W, gen)
    mask_out = mask_in if subm else F.max_pool2d(mask_in, 3, stride, 1)
    conv = _SpConv2d(cin, cout, 3, stride=stride, padding=1, bias=False).cuda().train()
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, device="cuda", generator=gen) * (2.0 / (9 * cin)) ** 0.5)
    x0 = (torch.randn((B, cin, H, W), device="cuda", generator=gen) * mask_in).contiguous(memory_format=torch.channels_last)
    Ho, Wo = mask_out.shape[2:]
    g0 = (torch.randn((B, cout, Ho, Wo), device="cuda", generator=gen) * mask_out).contiguous(memory_format=torch.channels_last)

    x = x0.clone().requires_grad_(True)
    y = masked_conv(conv, x, mask_out, mask_in)
    assert y.dtype == torch.float32 and type(y.grad_fn).__name__.startswith("_MaskedConv3x3F32Fn")
    assert y.is_contiguous(memory_format=torch.channels_last)
    y.backward(g0)
    dw, dx = conv.weight.grad.clone(), x.grad.clone()
    conv.weight.grad = None

    xr = x0.double().contiguous().requires_grad_(True)
    wr = conv.weight.detach().double().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, stride, 1) * mask_out.double()
    yr.backward(g0.double())
    rel = 2.0 ** -14
    # bounds from the absolute sums: |x| * |W| etc.
    ya = F.conv2d(x0.double().abs(), wr.detach().abs(), None, stride, 1)
    assert bool(((y.double() - yr.detach()).abs() <= rel * ya + 1e-30).all())
    assert bool((y[(mask_out == 0).expand_as(y)] == 0).all())
    dxa = torch.nn.grad.conv2d_input(xr.shape, wr.detach().abs(), g0.double().abs(), stride=stride, padding=1)
    dxm, dxr = dx.double() * mask_in, xr.grad * mask_in
    assert bool(((dxm - dxr).abs() <= rel * dxa + 1e-30).all())
    assert bool((dx[(mask_in == 0).expand_as(dx)] == 0).all())
    dwa = torch.nn.grad.conv2d_weight(x0.double().abs(), wr.shape, g0.double().abs(), stride=stride, padding=1)
    assert bool(((dw.double() - wr.grad).abs() <= rel * dwa + 1e-30).all())
    # how close in relative Frobenius norm (printed by -rP; asserted loosely: an accuracy regression of the split shows here first)
    for name, a, r in (("y", y, yr.detach()), ("dx", dxm, dxr), ("dw", dw, wr.grad)):
        e = float((a.double() - r).norm() / r.norm())
        print(f"fp32 node {cin}->{cout} s{stride} {name}: relative error {e:.2e}")
        assert e < 3e-5, (name, e)


@pytest.mark.parametrize("c,shape,shared", [(64, (2, 40, 72), True), (256, (1, 20, 33), False), (128, (1, 9, 31), False)])
def test_dense_fp32_layers_on_the_three_product_node(c, shape, shared):
    """models.x3_conv: the head's / neck's dense 3x3 nn.Conv2d (bias, every site active) on the same node, against fp64; `shared`: halves split once
    by the caller (SepHead.forward)."""
    import torch.nn.functional as F

    from pillarnext_amd import ops
    from pillarnext_amd.models import x3_conv

    B, H, W = shape
    gen = torch.Generator(device="cuda").manual_seed(c + H)
    conv = torch.nn.Conv2d(c, c, 3, padding=1, bias=True).cuda().train()
    with torch.no_grad():
        conv.bias.copy_(torch.randn(c, device="cuda", generator=gen))
    x0 = torch.randn((B, c, H, W), device="cuda", generator=gen).contiguous(memory_format=torch.channels_last)
    g0 = torch.randn((B, c, H, W), device="cuda", generator=gen).contiguous(memory_format=torch.channels_last)
    x = x0.clone().requires_grad_(True)
    y = x3_conv(conv, x, ops.split_f32(x.detach()) if shared else None)
    assert type(y.grad_fn).__name__.startswith("_MaskedConv3x3F32Fn")
    y.backward(g0)
    xr, wr, br = x0.double().requires_grad_(True), conv.weight.detach().double().requires_grad_(True), conv.bias.detach().double().requires_grad_(True)
    yr = F.conv2d(xr, wr, br, 1, 1)
    yr.backward(g0.double())
    for name, a, r in (("y", y, yr.detach()), ("dx", x.grad, xr.grad), ("dw", conv.weight.grad, wr.grad), ("db", conv.bias.grad, br.grad)):
        e = float((a.double() - r).norm() / r.norm())
        assert e < 3e-5, (name, e)
    conv.weight.grad = conv.bias.grad = None
    x2 = x0.clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):     # bf16 autocast: the same layer on the bf16 kernels, every site active
        yb = x3_conv(conv, x2)
    assert yb.dtype == torch.bfloat16 and type(yb.grad_fn).__name__.startswith("_MaskedConv3x3Fn")
    yb.backward(g0.to(torch.bfloat16))
    for name, a, r in (("y", yb, yr.detach()), ("dx", x2.grad, xr.grad), ("dw", conv.weight.grad, wr.grad), ("db", conv.bias.grad, br.grad)):
        e = float((a.double() - r).norm() / r.norm())
        assert e < 1.5e-2, (name, e)     # bf16 operands, fp32 accumulation
    conv.eval()
    assert not type(x3_conv(conv, x).grad_fn).__name__.startswith("_MaskedConv3x3F32Fn")


@pytest.mark.parametrize("k,shape,dtype", [(1, (2, 37, 50), torch.float32), (2, (1, 64, 64), torch.float32), (3, (2, 21, 19), torch.float32),
                                           (4, (1, 8, 16), torch.float32), (2, (2, 40, 33), torch.bfloat16), (3, (1, 30, 47), torch.bfloat16),
                                           (1, (1, 5, 3), torch.bfloat16)])
def test_sephead_output_convolution_kernels(k, shape, dtype):
    """csrc/head_train.hip (models._SmallKConv3x3Fn) against the fp64 autograd of F.conv2d on the same operands: fp32 maps to 1e-5 of the sum of |terms|,
    bf16 maps (fp32 accumulation of bf16 inputs, output rounded once) to one bf16 ulp on top."""
    import torch.nn.functional as F

    from pillarnext_amd.models import smallk_conv

    B, H, W = shape
    gen = torch.Generator(device="cuda").manual_seed(11 * k + H)
    conv = torch.nn.Conv2d(64, k, 3, padding=1, bias=True).cuda().train()
    with torch.no_grad():
        conv.bias.copy_(torch.randn(k, device="cuda", generator=gen))
    x0 = torch.randn((B, 64, H, W), device="cuda", generator=gen).to(dtype).contiguous(memory_format=torch.channels_last)
    g0 = torch.randn((B, k, H, W), device="cuda", generator=gen).to(dtype)
    x = x0.clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=dtype == torch.bfloat16):
        y = smallk_conv(conv, x)
    assert y.dtype == dtype and type(y.grad_fn).__name__.startswith("_SmallKConv3x3Fn") and tuple(y.shape) == (B, k, H, W)
    y.backward(g0)
    xr, wr, br = x0.double().requires_grad_(True), conv.weight.detach().double().requires_grad_(True), conv.bias.detach().double().requires_grad_(True)
    yr = F.conv2d(xr, wr, br, 1, 1)
    yr.backward(g0.double())
    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 0.0
    ya = F.conv2d(x0.double().abs(), wr.detach().abs(), br.detach().abs(), 1, 1)
    assert bool(((y.double() - yr.detach()).abs() <= 1e-5 * ya + ulp * yr.detach().abs()).all())
    dwa = torch.nn.grad.conv2d_weight(x0.double().abs(), wr.shape, g0.double().abs(), stride=1, padding=1)
    assert bool(((conv.weight.grad.double() - wr.grad).abs() <= 1e-5 * dwa).all())
    assert bool(((conv.bias.grad.double() - br.grad).abs() <= 1e-5 * g0.double().abs().sum(dim=(0, 2, 3))).all())
    # the data gradient is MIOpen's (fp32: its fp32 kernels; bf16: weights rounded to bf16 like any autocast convolution)
    e = float((x.grad.double() - xr.grad).norm() / xr.grad.norm())
    assert e < (2e-2 if dtype == torch.bfloat16 else 1e-5), e
    conv.eval()
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=dtype == torch.bfloat16):
        assert not type(smallk_conv(conv, x).grad_fn).__name__.startswith("_SmallKConv3x3Fn")


@pytest.mark.parametrize("amp", [False, True])
def test_sephead_training_path_against_its_modules(amp, monkeypatch):
    """models.SepHead.forward in training on a CUDA channels_last map (first convolutions on the three-product / bf16 nodes, BatchNorm + ReLU on the masked node,
    output convolutions on csrc/head_train.hip) against the same head with every switch off (the nn.Sequential statement of centerhead.py:12-59 on MIOpen):
    outputs, input gradient, every parameter gradient, running statistics."""
    import copy

    from pillarnext_amd.models import SepHead

    torch.manual_seed(4)
    heads = {"reg": (2, 2), "height": (1, 2), "dim": (3, 2), "rot": (2, 2), "vel": (2, 2), "hm": (2, 2)}
    head = SepHead(64, heads, stride=1, head_conv=64, final_kernel=3, bn=True).cuda().train().to(memory_format=torch.channels_last)
    ref = copy.deepcopy(head)
    x0 = torch.randn((2, 64, 40, 56), device="cuda").contiguous(memory_format=torch.channels_last)
    gs = {h: torch.randn((2, heads[h][0], 40, 56), device="cuda") for h in heads}

    def run(m, x):
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
            out = m(x)
        sum((out[h].float() * gs[h]).sum() for h in heads).backward()
        return out

    xa = x0.clone().requires_grad_(True)
    out = run(head, xa)
    kinds = {type(out[h].grad_fn).__name__ for h in heads}
    assert kinds == {"_SmallKConv3x3FnBackward"}, kinds                       # the HIP path ran
    # the reference: the modules themselves in fp64 (MIOpen's fp32 BatchNorm backward is itself ~2e-4 off, tests/test_gpu_masked_bn.py)
    for k in ("PNX_TRAIN_F32_HIP", "PNX_TRAIN_DENSE_HIP", "PNX_TRAIN_HEAD_HIP", "PNX_TRAIN_DENSE_BN_HIP"):
        monkeypatch.setenv(k, "0")
    ref = ref.double().to(memory_format=torch.contiguous_format)
    xb = x0.double().contiguous().requires_grad_(True)
    want = ref(xb)
    assert not any(type(want[h].grad_fn).__name__.startswith("_SmallK") for h in heads)
    sum((want[h] * gs[h].double()).sum() for h in heads).backward()
    # Outputs: bf16 operands | the three-product node's 4e-6 per layer, two layers and a BatchNorm deep.  Gradients pass a ReLU gate: a pre-activation within
    # the forward error of zero flips its gate, a discrete change of one term of a sum -- measured 7e-4 (dx) and 3e-3 (a BatchNorm bias) for the fp32 graph
    # on the three-product node, 2.5e-7 / 3.6e-7 for MIOpen's fp32 kernels (whose forward error is 20 x smaller), a few percent for bf16.
    tol_out, tol_grad = (4e-2, 8e-2) if amp else (5e-5, 1e-2)

    def close(a, b, what, tol):
        e = float((a.detach().double() - b.detach()).norm() / b.detach().norm().clamp(min=1e-12))
        assert e <= tol, (what, e)

    for h in heads:
        close(out[h], want[h], h, tol_out)
    close(xa.grad, xb.grad, "dx", tol_grad)
    big = max(float(r.grad.norm()) for r in ref.parameters())
    for (n, p), (_, q) in zip(head.named_parameters(), ref.named_parameters()):
        if float(q.grad.norm()) > 1e-3 * big:                                  # biases in front of a BatchNorm: zero in exact arithmetic
            close(p.grad, q.grad, n, tol_grad)
    for (n, p), (_, q) in zip(head.named_buffers(), ref.named_buffers()):
        assert torch.allclose(p.double(), q.double(), rtol=1e-2 if amp else 1e-4, atol=1e-3 if amp else 1e-5), n


def test_split_f32_halves():
    from pillarnext_amd import ops

    gen = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn((3, 64, 17, 8), device="cuda", generator=gen) * torch.logspace(-20, 20, 8, device="cuda")
    x[0, 0, 0, :4] = torch.tensor([0.0, -0.0, float("inf"), 1e-42], device="cuda")
    x = x.contiguous(memory_format=torch.channels_last)
    hi, lo = ops.split_f32(x)
    assert hi.dtype == torch.bfloat16 and hi.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(hi, x.to(torch.bfloat16))                                        # round to nearest even, like torch's cast
    fin = torch.isfinite(x)
    assert torch.equal(lo[fin], (x - hi.float()).to(torch.bfloat16)[fin])
    err = ((hi.float() + lo.float()) - x)[fin].abs()
    assert bool((err <= x[fin].abs() * 2.0 ** -16 + 1e-38).all())
    # with a mask: inactive sites are not read (their content does not matter), zeros are written
    m = (torch.rand((3, 17, 8), device="cuda", generator=gen) < 0.4).to(torch.uint8)
    xm = torch.where(m[:, None].bool(), x, torch.full_like(x, float("nan")))
    hm, lm = ops.split_f32(xm, m)
    keep = m[:, None].bool().expand_as(x)
    assert torch.equal(hm[keep], hi[keep]) and torch.equal(lm[keep & fin], lo[keep & fin])
    assert bool((hm[~keep] == 0).all()) and bool((lm[~keep] == 0).all())
    with pytest.raises(Exception):
        ops.split_f32(x.contiguous(), m)           # the mask form is for channels_last maps


def test_masked_conv_falls_back_outside_training_shapes():
    from pillarnext_amd.models import _SpConv2d, masked_conv

    conv = _SpConv2d(64, 64, 3, stride=1, padding=1, bias=False).cuda().train()
    x = torch.randn((1, 64, 16, 16), device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
    m = torch.ones((1, 1, 16, 16), device="cuda")
    y = masked_conv(conv, x, m, m)             # fp32, no autocast: the three-product node
    assert y.dtype == torch.float32 and type(y.grad_fn).__name__.startswith("_MaskedConv3x3F32Fn")
    conv.eval()
    y = masked_conv(conv, x, m, m)             # eval: MIOpen
    assert not type(y.grad_fn).__name__.startswith("_MaskedConv3x3")
    conv2 = _SpConv2d(64, 96, 3, stride=1, padding=1, bias=False).cuda().train()
    y = masked_conv(conv2, x, m, m)            # a shape without a kernel: MIOpen
    assert y.shape[1] == 96 and not type(y.grad_fn).__name__.startswith("_MaskedConv3x3")


@pytest.mark.parametrize("cin,cout,shape,p,stride", [(64, 64, (2, 70, 97), 0.12, 1), (64, 64, (1, 33, 31), 0.5, 1), (128, 128, (2, 41, 70), 0.12, 1),
                                                     (256, 256, (2, 23, 33), 0.2, 1), (64, 128, (1, 19, 40), 0.3, 1), (64, 128, (2, 50, 66), 0.12, 2),
                                                     (128, 256, (1, 37, 41), 0.3, 2), (256, 256, (2, 24, 64), 0.2, 2), (64, 64, (1, 9, 131), 0.4, 2)])
def test_wgrad_kernel_matches_fp32_conv2d_weight(cin, cout, shape, p, stride):
    """pnx_conv3x3_wgrad_bf16 (csrc/conv_wgrad.hip: K = pixels through the transposing LDS read, fp32 accumulation, fixed-order reduction)
    against torch.nn.grad.conv2d_weight in fp32 on the same bf16 operands; bit-identical from call to call; an upstream gradient that is
    NOT zero outside the mask must not contribute (the kernel applies the mask itself)."""
    from pillarnext_amd import ops

    import torch.nn.functional as F

    B, H, W = shape
    gen = torch.Generator(device="cuda").manual_seed(cin * 3 + cout + H)
    m_in = _lidar_mask(B, H, W, gen, p)
    m = m_in if stride == 1 else F.max_pool2d(m_in, 3, stride, 1)          # SparseConv2d: the output set is the pooled input set
    Ho, Wo = m.shape[2:]
    x = (torch.randn((B, cin, H, W), device="cuda", generator=gen) * m_in).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    g_all = torch.randn((B, cout, Ho, Wo), device="cuda", generator=gen).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    mu8 = (m[:, 0] != 0).to(torch.uint8).contiguous()
    dw = ops.conv3x3_wgrad(x, g_all, mu8, stride=stride)
    ref = torch.nn.grad.conv2d_weight(x.float().contiguous(), (cout, cin, 3, 3), (g_all.float() * m).contiguous(), stride=stride, padding=1)
    scale = float(ref.abs().max())
    assert dw.shape == ref.shape and dw.dtype == torch.float32
    assert float((dw - ref).abs().max()) <= 2e-4 * scale, float((dw - ref).abs().max()) / scale
    assert torch.equal(dw, ops.conv3x3_wgrad(x, g_all, mu8, stride=stride))
    empty = torch.zeros_like(mu8)
    assert float(ops.conv3x3_wgrad(x, g_all, empty, stride=stride).abs().max()) == 0.0


@pytest.mark.parametrize("co,ci", [(64, 64), (128, 64), (256, 256), (320, 64)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_pack_weights_kernel_equals_the_torch_statement(co, ci, dtype):
    """pnx_conv3x3_pack_weights (one launch, plain and transposed = data-gradient weights) bit for bit against the host statement of the layout."""
    from pillarnext_amd import ops

    w = torch.randn((co, ci, 3, 3), generator=torch.Generator().manual_seed(co + ci)).to(dtype)
    for transposed in (False, True):
        want = ops.conv3x3_pack_weights(w, transposed=transposed)              # CPU tensor: the torch statement
        got = ops.conv3x3_pack_weights(w.cuda(), transposed=transposed)        # CUDA tensor: the kernel
        assert want.dtype == got.dtype == torch.bfloat16 and want.shape == got.shape
        assert torch.equal(got.cpu().view(torch.int16), want.view(torch.int16)), (transposed,)

"""GPU: the IEEE-half twins of the convolution kernels (csrc/conv3x3.hip built with -DPNX_CONV_F16: pnx_conv3x3_f16, pnx_deconv2x2_f16,
pnx_sephead_out_f16, pnx_sephead_lazy_f16) -- BASELINE configs[4] (C5: the Waymo 3-sweep network in fp16) runs on them instead of on dense
MIOpen.  Each kernel against the fp32 torch statement of the same op on fp16-representable inputs (one fp16 rounding of an fp32-accumulated
sum: rtol 2e-3), the launch plans against the Python launch loop, and the whole fp16 detector against itself on MIOpen (PNX_HIP_CONV=0)."""
import os

import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu
F16 = torch.float16
TOL = dict(rtol=2e-3, atol=2e-3)


def _cl(t):
    return t.contiguous(memory_format=torch.channels_last)


@pytest.mark.parametrize("cin,cout,stride,residual", [(64, 64, 1, False), (64, 64, 1, True), (64, 128, 2, False), (128, 128, 1, True), (64, 384, 1, False),
                                                      (256, 256, 1, True), (256, 64, 1, False), (128, 256, 2, False), (256, 256, 2, False), (64, 128, 1, False)])
def test_conv3x3_f16_matches_torch(cin, cout, stride, residual):
    from pillarnext_amd import ops

    g = torch.Generator(device="cuda").manual_seed(cin + cout + stride)
    B, H, W = 2, 45, 70
    x = _cl((torch.randn((B, cin, H, W), device="cuda", generator=g) * (torch.rand((B, 1, H, W), device="cuda", generator=g) > 0.5)).to(F16))
    w = (torch.randn((cout, cin, 3, 3), device="cuda", generator=g) / (3 * cin ** 0.5)).to(F16)
    bias = torch.randn((cout,), device="cuda", generator=g)
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    mask = (torch.rand((B, Ho, Wo), device="cuda", generator=g) > 0.4).to(torch.uint8)
    mask[0, :9] = 0            # whole tiles without an active site
    mask[1, :, 33:] = 0
    res = _cl(torch.randn((B, cout, Ho, Wo), device="cuda", generator=g).to(F16)) if residual else None
    ref = torch.nn.functional.conv2d(x.float(), w.float(), None, stride, 1) + bias.view(1, -1, 1, 1)
    if residual:
        ref = ref + res.float()
    ref = torch.relu(ref) * mask.unsqueeze(1).float()
    wf = ops.conv3x3_pack_weights(w, dtype=F16)
    assert wf.dtype == F16
    got = ops.conv3x3_masked(x, wf, bias, cout, stride, mask, res, True)
    assert got.dtype == F16 and got.shape == ref.shape
    got = got.float()
    assert bool((got[(mask == 0).unsqueeze(1).expand_as(got)] == 0).all())
    torch.testing.assert_close(got, ref, **TOL)
    got2 = ops.conv3x3_masked(x, wf, bias, cout, stride, None, None, False).float()       # unmasked / no relu
    torch.testing.assert_close(got2, torch.nn.functional.conv2d(x.float(), w.float(), None, stride, 1) + bias.view(1, -1, 1, 1), **TOL)
    with pytest.raises(ops.PnxError):                                                     # bf16 weights under fp16 activations: refused, not reinterpreted
        ops.conv3x3_masked(x, ops.conv3x3_pack_weights(w.to(torch.bfloat16)), bias, cout, stride, mask, None, True)


def test_conv3x3_f16_persistent_workspace_and_tile_list():
    """row_dirty workspaces + tile lists (what the backbone plans use) in fp16: three different masks through the same buffers."""
    from pillarnext_amd import ops

    g = torch.Generator(device="cuda").manual_seed(11)
    B, H, W = 2, 83, 101
    w = (torch.randn((64, 64, 3, 3), device="cuda", generator=g) / 24).to(F16)
    bias = torch.randn((64,), device="cuda", generator=g)
    wf = ops.conv3x3_pack_weights(w, dtype=F16)
    ws = ops.conv3x3_workspace(B, 64, H, W, "cuda", F16)
    rows = ops.conv_tile_rows(64, 64, 1)
    for frame in range(3):
        mask = (torch.rand((B, H, W), device="cuda", generator=g) > (0.6, 0.97, 0.8)[frame]).to(torch.uint8)
        mask[frame % 2, 20:60] = 0
        x = _cl((torch.randn((B, 64, H, W), device="cuda", generator=g) * mask.unsqueeze(1)).to(F16))
        res = _cl(torch.randn((B, 64, H, W), device="cuda", generator=g).to(F16))
        ref = torch.relu(torch.nn.functional.conv2d(x.float(), w.float(), None, 1, 1) + bias.view(1, -1, 1, 1) + res.float()) * mask.unsqueeze(1).float()
        tiles = ops.conv_tile_list(mask, [ws[1]], rows)
        got = ops.conv3x3_masked(x, wf, bias, 64, 1, mask, res, True, out=ws, tiles=tiles).float()
        assert bool((got[(mask == 0).unsqueeze(1).expand_as(got)] == 0).all()), frame
        torch.testing.assert_close(got, ref, **TOL)


def test_deconv2x2_and_sephead_out_f16_match_torch():
    from pillarnext_amd import ops

    g = torch.Generator(device="cuda").manual_seed(3)
    B, H, W = 2, 37, 50
    x = _cl(torch.randn((B, 64, H, W), device="cuda", generator=g).to(F16))
    w = (torch.randn((64, 64, 2, 2), device="cuda", generator=g) / 8).to(F16)
    bias = torch.randn((64,), device="cuda", generator=g)
    got = ops.deconv2x2(x, ops.deconv2x2_pack_weights(w, dtype=F16), bias, 64, True)
    assert got.dtype == F16
    ref = torch.relu(torch.nn.functional.conv_transpose2d(x.float(), w.float(), None, 2) + bias.view(1, -1, 1, 1))
    torch.testing.assert_close(got.float(), ref, **TOL)
    for nb in (2, 6, 7):
        outs = [2, 1, 3, 2, 2, 1, 2][:nb]
        xx = _cl(torch.relu(torch.randn((B, nb * 64, H, W), device="cuda", generator=g)).to(F16))
        W2 = torch.zeros((16, nb * 64, 3, 3), device="cuda")
        o = 0
        for j, k in enumerate(outs):
            W2[o:o + k, 64 * j:64 * (j + 1)] = torch.randn((k, 64, 3, 3), device="cuda", generator=g) / 24
            o += k
        W2 = W2.to(F16)
        b2 = torch.randn((16,), device="cuda", generator=g)
        y = ops.sephead_out(xx, ops.sephead_pack_weights(W2, dtype=F16), b2)
        assert y.dtype == F16
        torch.testing.assert_close(y.float(), torch.nn.functional.conv2d(xx.float(), W2.float(), b2, padding=1), **TOL)


def test_lazy_kernel_f16_equals_dense_convolutions_at_the_cells():
    from test_gpu_lazy_head import _w2m, _weights
    from pillarnext_amd import ops

    B, H, W, pre_max = 2, 40, 36, 200
    g = torch.Generator(device="cuda").manual_seed(5)
    tasks, dense = [], []
    for ti in range(2):
        W1, b1, W2, b2 = _weights(3 + ti)
        W1, W2 = W1.to(F16).float(), W2.to(F16).float()
        up = _cl(torch.randn((B, 64, H, W), device="cuda", generator=g).to(F16))
        t = torch.relu(torch.nn.functional.conv2d(up.float(), W1, b1, padding=1)).to(F16).float()
        dense.append(torch.nn.functional.conv2d(t, W2, b2, padding=1).to(F16).float().permute(0, 2, 3, 1).reshape(-1, 10))
        tasks.append((up, ops.conv3x3_pack_weights(W1, dtype=F16), b1, ops.sephead_lazy_pack_w2(_w2m(W2)), b2))
    S = 2 * B
    local = torch.randint(0, B * H * W, (S, pre_max), device="cuda", generator=g)
    local[:, :4] = torch.tensor([0, W - 1, (H - 1) * W, B * H * W - 1], device="cuda")
    seg_len = torch.tensor([pre_max, 77, pre_max, 0], dtype=torch.int32, device="cuda")
    got = ops.sephead_lazy(tasks, [0, 1], B, local, seg_len, pre_max)
    valid = torch.arange(pre_max, device="cuda")[None, :] < seg_len[:, None]
    ref = torch.stack([dense[s % 2][local[s]] for s in range(S)]) * valid[..., None]
    assert bool((got[~valid] == 0).all())
    torch.testing.assert_close(got, ref, rtol=4e-3, atol=4e-3)      # an intermediate may round to the neighbouring fp16 value
    assert float((got == ref).float().mean()) > 0.8
    assert torch.equal(got, ops.sephead_lazy(tasks, [0, 1], B, local, seg_len, pre_max))


def test_fp16_detector_runs_on_the_hip_kernels_and_matches_miopen():
    """C5's network (fp16) through FusedPillarNeXt: every backbone / head convolution is a HIP module (no dense-MIOpen fallback), launch plans and
    the lazy head are on; detections equal the Python launch loop's bit for bit, and the head maps agree with the same fp16 network on MIOpen
    (PNX_HIP_CONV=0) to fp16 rounding of reordered fp32 sums."""
    from test_gpu_configs import _waymo_fused
    from pillarnext_amd import models, synth

    det, fused = _waymo_fused("C4", F16)      # C4 geometry (1504^2), fp16 network; C5 differs in the point count only
    assert all(isinstance(m, models._HipConv3x3) for mods in fused.stages for m in mods) and isinstance(fused.shared, models._HipConv3x3)
    assert all(isinstance(d, models._HipDeconv2x2) for d in fused.task_deblock) and fused.lazy_head and fused._plan_ok() and fused._head_plan_ok()
    assert fused.stages[0][0].wfrag.dtype == F16
    B = 2
    pts = torch.from_numpy(synth.make_batch("C4", B, "sweep")).cuda()
    ex = {"points": pts, "token": ["a", "b"], "batch_size": B}
    with torch.no_grad():
        d1 = fused(ex)
        fused.use_plan = False
        d2 = fused(ex)
        fused.use_plan = True
        os.environ["PNX_HIP_CONV"] = "0"
        try:
            ref_net = models.FusedPillarNeXt(det, dtype=F16).cuda().eval()
        finally:
            os.environ.pop("PNX_HIP_CONV", None)
        assert not any(isinstance(m, models._HipConv3x3) for mods in ref_net.stages for m in mods)
        pa = fused.forward_preds(pts, B, lazy=False)
        pb = ref_net.forward_preds(pts, B)
    for tok in ("a", "b"):
        assert len(d1[tok]["scores"]) > 0
        assert torch.equal(d1[tok]["scores"], d2[tok]["scores"]) and torch.equal(d1[tok]["box3d_lidar"], d2[tok]["box3d_lidar"])
    for a, b in zip(pa, pb):
        for k in a:
            x, y = a[k].float(), b[k].float()
            assert x.shape == y.shape and bool(torch.isfinite(x).all())
            # 24 layers of fp16 storage in a different summation order: compare in the scale of the map
            assert float((x - y).abs().max()) <= 0.03 * float(y.abs().max()) + 1e-3, k

"""GPU: the fused dense epilogue / mask-pool kernels vs their PyTorch statement, and the point uploader."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("residual", [False, True])
@pytest.mark.parametrize("masked", [False, True])
@pytest.mark.parametrize("relu", [False, True])
def test_bias_act_mask_matches_torch(residual, masked, relu):
    from pillarnext_amd import ops

    g = torch.Generator(device="cuda").manual_seed(0)
    B, C, H, W = 2, 24, 37, 53
    x = torch.randn((B, C, H, W), device="cuda", generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    res = torch.randn((B, C, H, W), device="cuda", generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last) if residual else None
    bias = torch.randn((C,), device="cuda", generator=g)
    mask = (torch.rand((B, H, W), device="cuda", generator=g) > 0.6).to(torch.uint8) if masked else None
    ref = x.float() + bias.view(1, -1, 1, 1)
    if residual:
        ref = ref + res.float()
    if relu:
        ref = torch.relu(ref)
    ref = ref.to(torch.bfloat16)                       # one rounding, as in the kernel
    if masked:
        ref = ref * mask.unsqueeze(1).to(torch.bfloat16)
    got = ops.bias_act_mask_(x.clone(memory_format=torch.channels_last), bias, mask, res, relu)
    assert torch.equal(got, ref)


@pytest.mark.parametrize("masked", [False, True])
def test_bias_act_mask_residual_after_relu(masked):
    """relu mode 2: (relu(x + b) + residual) * mask -- the neck's BasicBlock, whose identity joins after block2's own ReLU."""
    from pillarnext_amd import ops

    g = torch.Generator(device="cuda").manual_seed(3)
    B, C, H, W = 2, 24, 37, 53
    x = torch.randn((B, C, H, W), device="cuda", generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    res = torch.randn((B, C, H, W), device="cuda", generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    bias = torch.randn((C,), device="cuda", generator=g)
    mask = (torch.rand((B, H, W), device="cuda", generator=g) > 0.6).to(torch.uint8) if masked else None
    ref = (torch.relu(x.float() + bias.view(1, -1, 1, 1)) + res.float()).to(torch.bfloat16)
    if masked:
        ref = ref * mask.unsqueeze(1).to(torch.bfloat16)
    got = ops.bias_act_mask_(x.clone(memory_format=torch.channels_last), bias, mask, res, 2)
    assert torch.equal(got, ref)


@pytest.mark.parametrize("hw", [(45, 62), (45, 64), (37, 120), (8, 8), (5, 4), (37, 128), (9, 32), (5, 16), (33, 96)])   # W, Wo multiples of 4 / of 16: the 4- and 16-sites-per-thread kernels
@pytest.mark.parametrize("stride", [1, 2])
def test_mask_pool3_matches_maxpool(stride, hw):
    from pillarnext_amd import ops

    g = torch.Generator(device="cuda").manual_seed(1)
    m = (torch.rand((3,) + hw, device="cuda", generator=g) > 0.9).to(torch.uint8)
    m = m * torch.randint(1, 256, m.shape, device="cuda", generator=g).to(torch.uint8)      # any non-zero byte marks an active site
    m[0, :, -1] = 128                                                                       # right border
    m[1, -1, :] = 0
    ref = torch.nn.functional.max_pool2d((m != 0).float().unsqueeze(1), 3, stride, 1).squeeze(1).to(torch.uint8)
    assert torch.equal(ops.mask_pool3(m, stride), ref)


def test_point_uploader_roundtrip():
    from pillarnext_amd.io import PointUploader, collate_points

    rng = np.random.default_rng(0)
    up = PointUploader(10_000)
    for it in range(5):
        clouds = [rng.standard_normal((int(rng.integers(100, 2000)), 5)).astype(np.float32) for _ in range(3)]
        dev, B = up.upload(clouds)
        assert B == 3
        ref = collate_points(clouds)
        assert np.array_equal(dev.cpu().numpy(), ref)
        assert np.array_equal(np.unique(ref[:, 0]), [0, 1, 2])


@pytest.mark.parametrize("residual", [False, True])
def test_conv3x3_64_stale_workspace(residual):
    """The 64 -> 64 row kernel on dense-ish, sparse and empty tiles, with a workspace that goes stale (row_dirty)."""
    from pillarnext_amd import ops

    g = torch.Generator(device="cuda").manual_seed(3)
    B, H, W = 2, 83, 101
    w = (torch.randn((64, 64, 3, 3), device="cuda", generator=g) / 24).to(torch.bfloat16)
    wf = ops.conv3x3_pack_weights(w)
    bias = torch.randn((64,), device="cuda", generator=g)
    ws = ops.conv3x3_workspace(B, 64, H, W, "cuda")
    for frame in range(3):
        mask = (torch.rand((B, H, W), device="cuda", generator=g) > (0.2, 0.9, 0.97)[frame]).to(torch.uint8)   # > 256 active pixels per tile, then sparse
        mask[0, 16 * frame:16 * frame + 20] = 0
        x = (torch.randn((B, 64, H, W), device="cuda", generator=g) * mask.unsqueeze(1)).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        res = torch.randn((B, 64, H, W), device="cuda", generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last) if residual else None
        ref = torch.nn.functional.conv2d(x.float(), w.float(), None, 1, 1) + bias.view(1, -1, 1, 1)
        if residual:
            ref = ref + res.float()
        ref = torch.relu(ref) * mask.unsqueeze(1).float()
        tiles = ops.conv_tile_list(mask, [ws[1]], ops.conv_tile_rows(64, 64, 1))
        got = ops.conv3x3_masked(x, wf, bias, 64, 1, mask, res, True, out=ws, tiles=tiles).float()
        assert bool((got[(mask == 0).unsqueeze(1).expand_as(got)] == 0).all()), frame
        torch.testing.assert_close(got, ref, rtol=1.6e-2, atol=2e-2)


@pytest.mark.parametrize("cin,cout,stride", [(64, 64, 1), (64, 128, 2), (128, 128, 1), (64, 64, 2), (64, 384, 1), (64, 320, 1), (256, 256, 1), (256, 64, 1), (128, 256, 2),
                                               (256, 256, 2)])
@pytest.mark.parametrize("residual", [False, True])
def test_conv3x3_masked_matches_torch(cin, cout, stride, residual):
    from pillarnext_amd import ops

    if residual and (stride != 1 or cin != cout):
        pytest.skip("residual only on submanifold blocks")
    g = torch.Generator(device="cuda").manual_seed(cin + cout + stride)
    B, H, W = 2, 45, 70
    x = (torch.randn((B, cin, H, W), device="cuda", generator=g) * (torch.rand((B, 1, H, W), device="cuda", generator=g) > 0.5)).to(torch.bfloat16)
    x = x.contiguous(memory_format=torch.channels_last)
    w = (torch.randn((cout, cin, 3, 3), device="cuda", generator=g) / (3 * cin ** 0.5)).to(torch.bfloat16)
    bias = torch.randn((cout,), device="cuda", generator=g)
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    mask = (torch.rand((B, Ho, Wo), device="cuda", generator=g) > 0.4).to(torch.uint8)
    mask[0, :9] = 0            # whole tiles without an active site
    mask[1, :, 33:] = 0
    res = torch.randn((B, cout, Ho, Wo), device="cuda", generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last) if residual else None
    ref = torch.nn.functional.conv2d(x.float(), w.float(), None, stride, 1) + bias.view(1, -1, 1, 1)
    if residual:
        ref = ref + res.float()
    ref = torch.relu(ref) * mask.unsqueeze(1).float()
    got = ops.conv3x3_masked(x, ops.conv3x3_pack_weights(w), bias, cout, stride, mask, res, True).float()
    assert got.shape == ref.shape
    assert bool((got[(mask == 0).unsqueeze(1).expand_as(got)] == 0).all())
    torch.testing.assert_close(got, ref, rtol=1.6e-2, atol=2e-2)      # one bf16 rounding of an fp32-accumulated sum
    # unmasked / no relu
    got2 = ops.conv3x3_masked(x, ops.conv3x3_pack_weights(w), bias, cout, stride, None, None, False).float()
    ref2 = torch.nn.functional.conv2d(x.float(), w.float(), None, stride, 1) + bias.view(1, -1, 1, 1)
    torch.testing.assert_close(got2, ref2, rtol=1.6e-2, atol=2e-2)


@pytest.mark.parametrize("cin,cout,residual", [(64, 64, False), (64, 64, True), (64, 384, False), (128, 128, False), (128, 128, True), (256, 64, False), (256, 256, False),
                                               (256, 256, True)])
def test_conv3x3_sparse_rows(cin, cout, residual):
    """Few active rows per 16-row tile: the per-row-count (NR = 1..3) paths, dummy rows and zero-filled rows of the LDS kernel."""
    from pillarnext_amd import ops

    g = torch.Generator(device="cuda").manual_seed(7 + cout)
    B, H, W = 2, 83, 101
    mask = torch.zeros((B, H, W), dtype=torch.uint8, device="cuda")
    for r in (0, 3, 17, 18, 19, 40, 41, 42, 43, 44, 45, 46, 63, 82):      # 1, 1+3, 7 (-> waves with 2 and 1 rows), 1, last row
        mask[0, r] = (torch.rand((W,), device="cuda", generator=g) > 0.5).to(torch.uint8)
    mask[1] = (torch.rand((H, W), device="cuda", generator=g) > 0.97).to(torch.uint8)
    x = (torch.randn((B, cin, H, W), device="cuda", generator=g) * mask.unsqueeze(1)).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn((cout, cin, 3, 3), device="cuda", generator=g) / 24).to(torch.bfloat16)
    bias = torch.randn((cout,), device="cuda", generator=g)
    res = torch.randn((B, cout, H, W), device="cuda", generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last) if residual else None
    ref = torch.nn.functional.conv2d(x.float(), w.float(), None, 1, 1) + bias.view(1, -1, 1, 1)
    if residual:
        ref = ref + res.float()
    ref = torch.relu(ref) * mask.unsqueeze(1).float()
    y = torch.full((B, cout, H, W), 7.0, dtype=torch.bfloat16, device="cuda").contiguous(memory_format=torch.channels_last)  # noqa: F841 (poison)
    got = ops.conv3x3_masked(x, ops.conv3x3_pack_weights(w), bias, cout, 1, mask, res, True).float()
    assert bool((got[(mask == 0).unsqueeze(1).expand_as(got)] == 0).all())
    torch.testing.assert_close(got, ref, rtol=1.6e-2, atol=2e-2)


@pytest.mark.parametrize("nb", [5, 6, 7])
def test_sephead_out_matches_torch(nb):
    """k_sephead_out vs the dense conv over the block-diagonal weight (what the merged SepHead's last convolutions compute)."""
    from pillarnext_amd import ops

    g = torch.Generator(device="cuda").manual_seed(nb)
    B, H, W = 2, 45, 70
    outs = [2, 1, 3, 2, 2, 1, 2][:nb]
    x = torch.relu(torch.randn((B, nb * 64, H, W), device="cuda", generator=g)).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    W2 = torch.zeros((16, nb * 64, 3, 3), device="cuda")
    o = 0
    for j, k in enumerate(outs):
        W2[o:o + k, j * 64:(j + 1) * 64] = torch.randn((k, 64, 3, 3), device="cuda", generator=g) / 24
        o += k
    W2 = W2.to(torch.bfloat16)
    bias = torch.zeros((16,), device="cuda")
    bias[:o] = torch.randn((o,), device="cuda", generator=g)
    ref = torch.nn.functional.conv2d(x.float(), W2.float(), None, 1, 1) + bias.view(1, -1, 1, 1)
    got = ops.sephead_out(x, ops.sephead_pack_weights(W2), bias).float()
    assert got.shape == ref.shape
    assert bool((got[:, o:] == 0).all())
    torch.testing.assert_close(got, ref, rtol=1.6e-2, atol=2e-2)


@pytest.mark.parametrize("cin,cout,stride", [(64, 64, 1), (128, 128, 1), (64, 128, 2), (256, 256, 1), (128, 256, 2), (256, 256, 2)])
def test_conv3x3_row_dirty_workspace(cin, cout, stride):
    """Persistent output buffer + row_dirty flags (pnx.h): three frames with different active sets through ONE workspace must
    equal the stateless result every time -- stale rows of an earlier frame are cleared, untouched rows stay zero."""
    from pillarnext_amd import ops

    g = torch.Generator(device="cuda").manual_seed(cin + stride)
    B, H, W = 2, 70, 99
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    w = (torch.randn((cout, cin, 3, 3), device="cuda", generator=g) / 24).to(torch.bfloat16)
    wf = ops.conv3x3_pack_weights(w)
    bias = torch.randn((cout,), device="cuda", generator=g)
    ws = ops.conv3x3_workspace(B, cout, Ho, Wo, "cuda")
    for frame in range(3):
        mask = torch.zeros((B, Ho, Wo), dtype=torch.uint8, device="cuda")
        rows = torch.randperm(Ho, device="cuda", generator=g)[: 6 + 5 * frame]
        mask[0, rows] = (torch.rand((rows.numel(), Wo), device="cuda", generator=g) > 0.6).to(torch.uint8)
        if frame != 1:
            mask[1, 5 * frame:5 * frame + 20, 10:40] = (torch.rand((20, 30), device="cuda", generator=g) > 0.8).to(torch.uint8)
        x = torch.randn((B, cin, H, W), device="cuda", generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        ref = ops.conv3x3_masked(x, wf, bias, cout, stride, mask, None, True)
        got = ops.conv3x3_masked(x, wf, bias, cout, stride, mask, None, True, out=ws)
        assert got.data_ptr() == ws[0].data_ptr()
        assert torch.equal(got, ref), f"frame {frame}"
        seg_active = torch.nn.functional.max_pool1d(mask.float(), 32, 32, ceil_mode=True) > 0
        assert torch.equal(ws[1] != 0, seg_active), "row_dirty must equal the active row segments after a call"


@pytest.mark.gpu
@pytest.mark.parametrize("n,relu", [(1, True), (5, True), (5, False), (8, True)])
def test_sum_bias_act_matches_torch(n, relu):
    from pillarnext_amd import ops

    torch.manual_seed(n)
    parts = [torch.randn(2, 32, 19, 23, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last) for _ in range(n)]
    bias = torch.randn(32, device="cuda")
    got = ops.sum_bias_act(parts, bias, relu)
    want = sum(p.float() for p in parts) + bias.view(1, -1, 1, 1)
    if relu:
        want = torch.relu(want)
    assert got.is_contiguous(memory_format=torch.channels_last)
    torch.testing.assert_close(got.float(), want.to(torch.bfloat16).float(), rtol=0, atol=0)   # fp32 sum in the same order, one rounding


@pytest.mark.parametrize("hw", [(45, 70), (7, 33), (180, 180)])
def test_deconv2x2_matches_torch(hw):
    """k_deconv2x2_64 vs F.conv_transpose2d (kernel 2, stride 2) + bias + ReLU."""
    from pillarnext_amd import ops

    H, W = hw
    g = torch.Generator(device="cuda").manual_seed(H * W)
    x = torch.randn((2, 64, H, W), device="cuda", generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn((64, 64, 2, 2), device="cuda", generator=g) / 8).to(torch.bfloat16)
    bias = torch.randn((64,), device="cuda", generator=g)
    for relu in (True, False):
        ref = torch.nn.functional.conv_transpose2d(x.float(), w.float(), None, 2) + bias.view(1, -1, 1, 1)
        if relu:
            ref = torch.relu(ref)
        got = ops.deconv2x2(x, ops.deconv2x2_pack_weights(w), bias, 64, relu)
        assert got.shape == ref.shape and got.is_contiguous(memory_format=torch.channels_last)
        torch.testing.assert_close(got.float(), ref, rtol=1.6e-2, atol=2e-2)


@pytest.mark.parametrize("W", [200, 224])     # the list kernel reads the mask bytewise / as aligned 16-byte pieces
@pytest.mark.parametrize("cin,cout", [(64, 64), (128, 128), (256, 256)])
def test_conv3x3_tile_list_equals_full_walk(cin, cout, W):
    """Three frames through one workspace, once walking all tiles and once the listed ones (active site or stale row): same bytes,
    same row_dirty.  The list is built from the mask and the workspace's flags BEFORE the conv, as a backbone stage does."""
    from pillarnext_amd import ops

    g = torch.Generator(device="cuda").manual_seed(cin)
    B, H = 2, 150
    w = (torch.randn((cout, cin, 3, 3), device="cuda", generator=g) / 24).to(torch.bfloat16)
    wf = ops.conv3x3_pack_weights(w)
    bias = torch.randn((cout,), device="cuda", generator=g)
    rows = ops.conv_tile_rows(cin, cout, 1)
    assert rows in (8, 16)
    ws_a, ws_b = ops.conv3x3_workspace(B, cout, H, W, "cuda"), ops.conv3x3_workspace(B, cout, H, W, "cuda")
    for frame in range(3):
        mask = torch.zeros((B, H, W), dtype=torch.uint8, device="cuda")
        y0, x0 = 20 + 40 * frame, 30 * frame                       # a blob that moves: earlier tiles go stale
        mask[0, y0:y0 + 35, x0:x0 + 70] = (torch.rand((35, 70), device="cuda", generator=g) > 0.7).to(torch.uint8)
        mask[1, 5 * frame + 3, :] = 1
        x = (torch.randn((B, cin, H, W), device="cuda", generator=g) * mask.unsqueeze(1)).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        res = torch.randn((B, cout, H, W), device="cuda", generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        tiles = ops.conv_tile_list(mask, [ws_b[1]], rows)
        n_list = int(tiles[1].item())
        seg = torch.nn.functional.max_pool2d((mask | torch.repeat_interleave(ws_b[1], 32, dim=2)[:, :, :W]).float().unsqueeze(1), (rows, 32), (rows, 32),
                                             ceil_mode=True)
        assert n_list == int(seg.sum().item())                      # exactly the tiles with an active site or a stale row
        assert sorted(tiles[0][:n_list].tolist()) == torch.nonzero(seg.flatten() > 0).flatten().tolist()
        n_all = B * ((H + rows - 1) // rows) * ((W + 31) // 32)
        assert 0 < n_list < n_all // 2
        ref = ops.conv3x3_masked(x, wf, bias, cout, 1, mask, res, True, out=ws_a)
        got = ops.conv3x3_masked(x, wf, bias, cout, 1, mask, res, True, out=ws_b, tiles=tiles)
        assert torch.equal(got, ref), f"frame {frame}"
        assert torch.equal(ws_a[1], ws_b[1])
        want = torch.relu(torch.nn.functional.conv2d(x.float(), w.float(), None, 1, 1) + bias.view(1, -1, 1, 1) + res.float()) * mask.unsqueeze(1)
        torch.testing.assert_close(got.float(), want, rtol=1.6e-2, atol=2e-2)


def test_producer_consumer_kernel_on_every_shape_it_is_built_for():
    """k_conv3x3_pc serves 64 -> 64 by default; its 128 / 256-channel and 64 -> 320 / 384 / 448 forms are selected by PNX_CONV_PC bits 1 and 2, which the library
    reads once per process -- so they run in a child (tools/conv_pc_check.py: dense, half-dense and sparse masks, ragged sizes, residual, tile lists, stale
    workspaces, against fp32 torch)."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PNX_CONV_PC="7")
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "conv_pc_check.py")], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and p.stdout.strip().endswith("OK"), (p.stdout[-1500:], p.stderr[-500:])

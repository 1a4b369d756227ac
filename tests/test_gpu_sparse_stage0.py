"""GPU: the sparse first backbone stage (feature rows + occupancy words instead of the dense canvas):
  * pnx_reader_forward_rows against pnx_reader_forward's canvas / occupancy (same pillars, same values, rank = position in (b, xi, yi) order)
  * pnx_subm64_sparse_bf16 against torch's dense convolution restricted to the active sites (spconv's SubMConv2d semantics)
  * pnx_conv3x3_s2_sparse_bf16 against the dense HIP kernel on the densified tensor: the same arithmetic, bit for bit
  * the fused detector with and without the sparse stage: same detections up to the bf16 rounding of reordered fp32 sums."""
import os

import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _ranks(wfull, wpr, b, xi, yi, gx):
    """rank of cells (b, xi, yi) from the occupancy words, in torch."""
    w = (b * gx + xi) * wpr + (yi >> 5)
    bits = wfull[w, 0].long() & 0xFFFFFFFF
    below = bits & ((1 << (yi & 31)) - 1)
    pop = torch.zeros_like(below)
    for k in range(32):
        pop += (below >> k) & 1
    return wfull[w, 1].long() + pop


def test_reader_rows_equal_the_dense_canvas():
    from pillarnext_amd import synth
    from test_gpu_reader import make_net

    cfg = synth.CONFIGS["C2"]
    net = make_net(cfg["pc_range"], cfg["voxel_size"], synth.pfn_params(5, (64, 64), 0)).eval()
    B = 2
    pts = torch.from_numpy(synth.make_batch("C2", B, "sweep")).cuda()
    ny, nx = (int(v) for v in net.grid_size)
    occ_d = torch.empty((B, ny, nx), dtype=torch.uint8, device="cuda")
    canvas = net.forward_dense(pts, B, dtype=torch.bfloat16, occupancy=occ_d)            # (B, 64, ny, nx) channels_last
    occ_s = torch.empty_like(occ_d)
    counts = torch.zeros((2,), dtype=torch.int32, device="cuda")
    rows, wfull, wpr = net.forward_rows(pts, B, occupancy=occ_s, counts=counts)
    assert torch.equal(occ_d, occ_s)
    P = int(counts[0])
    assert P == int(occ_d.sum()) and rows.shape[0] >= P
    b, yi, xi = torch.nonzero(occ_d, as_tuple=True)
    r = _ranks(wfull, wpr, b, xi, yi, nx)
    assert int(r.min()) == 0 and int(r.max()) == P - 1 and r.unique().numel() == P         # a permutation of 0..P-1
    key = (b * nx + xi) * (wpr * 32) + yi
    assert torch.equal(torch.argsort(key), torch.argsort(r))                                # rank order = (b, xi, yi) order
    assert torch.equal(rows[r], canvas.permute(0, 2, 3, 1)[b, yi, xi])                      # the same bf16 lines
    # the words: bits == occupancy, prefix == exclusive popcount scan
    ws = ops_words(occ_d, wpr)
    assert torch.equal(wfull[:, 0], ws[0]) and torch.equal(wfull[:, 1], ws[1])


def ops_words(occ, wpr):
    from pillarnext_amd import ops

    w, wpr2 = ops.sparse_index_from_mask(occ.transpose(1, 2).contiguous())
    assert wpr2 == wpr
    return w[:, 0], w[:, 1]


def _sparse_case(B, gx, gy, density, seed, dense_blob=True):
    from pillarnext_amd import ops

    g = torch.Generator(device="cuda").manual_seed(seed)
    mt = (torch.rand((B, gx, gy), device="cuda", generator=g) < density)
    if dense_blob:
        mt[0, 3:19, 5:37] = torch.rand((16, 32), device="cuda", generator=g) < 0.8       # a tile with more than 256 active cells: two rounds
    mt[B - 1, gx - 1, gy - 1] = True
    mt[0, 0, 0] = True
    wfull, wpr = ops.sparse_index_from_mask(mt)
    P = int(mt.sum())
    rows = torch.randn((P + 7, 64), device="cuda", generator=g).to(torch.bfloat16)
    b, xi, yi = torch.nonzero(mt, as_tuple=True)                                          # (b, xi, yi) order == rank order
    dense = torch.zeros((B, gy, gx, 64), dtype=torch.bfloat16, device="cuda")
    dense[b, yi, xi] = rows[:P]
    return mt, wfull, wpr, rows, dense.permute(0, 3, 1, 2), (b, xi, yi), P, g


@pytest.mark.parametrize("shape,density", [((2, 50, 70), 0.08), ((1, 16, 32), 0.5), ((3, 37, 129), 0.03)])
@pytest.mark.parametrize("residual", [False, True])
@pytest.mark.parametrize("with_tiles", [False, True])
def test_subm64_sparse_matches_dense_conv(shape, density, residual, with_tiles):
    from pillarnext_amd import ops

    B, gx, gy = shape
    mt, wfull, wpr, rows, dense, (b, xi, yi), P, g = _sparse_case(B, gx, gy, density, 11, dense_blob=gx >= 19 and gy >= 37)
    w = (torch.randn((64, 64, 3, 3), device="cuda", generator=g) / 24).to(torch.bfloat16)
    bias = torch.randn((64,), device="cuda", generator=g)
    res = torch.randn(rows.shape, device="cuda", generator=g).to(torch.bfloat16) if residual else None
    ref = torch.nn.functional.conv2d(dense.float(), w.float(), None, 1, 1) + bias.view(1, -1, 1, 1)
    ref = ref.permute(0, 2, 3, 1)[b, yi, xi]
    if residual:
        ref = ref + res[:P].float()
    ref = torch.relu(ref)
    tiles = ops.sparse_tile_list(wfull, B, gx, wpr) if with_tiles else None
    if with_tiles:
        n = int(tiles[1])
        occ_t = torch.nn.functional.max_pool2d(torch.nn.functional.pad(mt.float(), (0, wpr * 32 - gy)).unsqueeze(1), (16, 32), (16, 32), ceil_mode=True)
        assert n == int(occ_t.sum()) and sorted(tiles[0][:n].tolist()) == torch.nonzero(occ_t.flatten() > 0).flatten().tolist()
    out = torch.full_like(rows, 7.0)
    got = ops.subm64_sparse(rows, wfull, B, gx, wpr, ops.conv3x3_pack_weights(w.transpose(2, 3)), bias, residual=res, relu=True, out=out, tiles=tiles)
    torch.testing.assert_close(got[:P].float(), ref, rtol=1.6e-2, atol=2e-2)              # one bf16 rounding of an fp32-accumulated sum
    assert bool((got[P:] == 7.0).all())                                                     # nothing written behind the last pillar


@pytest.mark.parametrize("shape,density", [((2, 70, 50), 0.08), ((1, 129, 67), 0.2)])
def test_conv3x3_s2_sparse_equals_the_dense_kernel(shape, density):
    from pillarnext_amd import ops

    B, gx, gy = shape
    mt, wfull, wpr, rows, dense, _, P, g = _sparse_case(B, gx, gy, density, 5, dense_blob=False)
    w = (torch.randn((128, 64, 3, 3), device="cuda", generator=g) / 24).to(torch.bfloat16)
    bias = torch.randn((128,), device="cuda", generator=g)
    wf = ops.conv3x3_pack_weights(w)
    xd = dense.contiguous(memory_format=torch.channels_last)                               # (B, 64, gy, gx)
    mask = ops.mask_pool3(mt.transpose(1, 2).contiguous().to(torch.uint8), 2)
    ref = ops.conv3x3_masked(xd, wf, bias, 128, 2, mask, None, True)
    got = ops.conv3x3_s2_sparse(rows, wfull, B, gy, gx, wpr, wf, bias, 128, mask=mask, relu=True)
    assert got.shape == ref.shape and torch.equal(got, ref)


def test_fused_detector_with_and_without_the_sparse_stage():
    from pillarnext_amd import synth
    from pillarnext_amd.models import FusedPillarNeXt, build_pillarnext_b

    cfg = synth.CONFIGS["C2"]
    torch.manual_seed(0)
    det = build_pillarnext_b(cfg["pc_range"], cfg["voxel_size"]).cuda().eval()
    # PillarNeXt-B opens stage 0 with a SparseConv2d (sparse_resnet.py:53-54: use_subm=False, the active set dilates); the sparse path covers
    # backbones whose first layer is submanifold, so this test runs such a variant through both graphs
    det.backbone.blocks[0][0].subm = True
    sparse = FusedPillarNeXt(det).cuda().eval()
    os.environ["PNX_SPARSE_STAGE0"] = "0"
    try:
        dense = FusedPillarNeXt(det).cuda().eval()
    finally:
        os.environ.pop("PNX_SPARSE_STAGE0", None)
    assert sparse.sparse0 and not dense.sparse0
    B = 2
    pts = torch.from_numpy(synth.make_batch("C2", B, "sweep")).cuda()
    with torch.no_grad():
        for frame in range(2):                                                              # twice: the persistent buffers are reused
            ps, pd = sparse.forward_preds(pts, B), dense.forward_preds(pts, B)
            for a, b_ in zip(ps, pd):
                for k in a:
                    x, y = a[k].float(), b_[k].float()
                    assert float((x - y).abs().max()) <= 0.05 * float(y.abs().max()) + 1e-3, (frame, k)
                    assert float((x - y).abs().mean()) <= 2e-3 * float(y.abs().mean()) + 1e-4, (frame, k)
        ex = {"points": pts, "token": ["a", "b"], "batch_size": B}
        ds, dd = sparse(ex), dense(ex)
    for tok in ("a", "b"):
        assert abs(len(ds[tok]["scores"]) - len(dd[tok]["scores"])) <= max(3, len(dd[tok]["scores"]) // 50)

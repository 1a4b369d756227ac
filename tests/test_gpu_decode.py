"""GPU parity: CenterHead.predict (decode + per-class rotated NMS) vs the golden produced by running the
reference's CenterHead.predict (centerhead.py:231-384) on the same head outputs."""
import numpy as np
import pytest

from conftest import load_golden

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def test_decode_two_tasks_matches_reference():
    from pillarnext_amd.models import CenterHead

    g = load_golden("decode_2task")
    tasks = [["car"], ["truck", "construction_vehicle"]]
    common = {"reg": (2, 2), "height": (1, 2), "dim": (3, 2), "rot": (2, 2), "vel": (2, 2), "iou": (1, 2)}
    head = CenterHead(16, tasks, 0.25, [1.0] * 10, common, [2, 2], share_conv_channel=16, rectifier=[[0.5], [0.68, 0.2]]).cuda()
    preds = []
    for t in range(2):
        preds.append({k: torch.from_numpy(g[f"t{t}_{k}"]).cuda() for k in ("reg", "height", "dim", "rot", "vel", "iou", "hm")})
    test_cfg = dict(post_center_limit_range=list(g["post_center_limit_range"]), score_threshold=float(g["score_threshold"]),
                    nms=dict(nms_pre_max_size=int(g["pre_max"]), nms_post_max_size=int(g["post_max"]), nms_iou_threshold=[[0.2], [0.2, 0.25]]),
                    out_size_factor=[int(v) for v in g["out_size_factor"]], voxel_size=list(g["voxel_size"]), pc_range=list(g["pc_range"]))
    res = head.predict({"token": ["a", "b"]}, preds, test_cfg)
    assert [r["token"] for r in res] == ["a", "b"]
    for i, r in enumerate(res):
        assert np.array_equal(r["label_preds"].cpu().numpy(), g[f"s{i}_labels"])       # same boxes kept, same order
        np.testing.assert_allclose(r["scores"].cpu().numpy(), g[f"s{i}_scores"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(r["box3d_lidar"].cpu().numpy(), g[f"s{i}_boxes"], rtol=1e-4, atol=1e-4)


def test_detector_end_to_end_runs_and_is_deterministic():
    from pillarnext_amd import synth
    from pillarnext_amd.models import build_pillarnext_b

    cfg = synth.CONFIGS["C1"]
    torch.manual_seed(0)
    torch.backends.cudnn.deterministic = True  # MIOpen: deterministic conv algorithms (the HIP path is deterministic by construction)
    model = build_pillarnext_b(cfg["pc_range"], cfg["voxel_size"], tasks=[["car"]]).cuda().eval()
    model.backbone.to(memory_format=torch.channels_last, dtype=torch.bfloat16)
    model.neck.to(memory_format=torch.channels_last, dtype=torch.bfloat16)
    model.head.to(memory_format=torch.channels_last, dtype=torch.bfloat16)
    pts = torch.from_numpy(synth.make_batch("C1", 2, "sweep", n=20_000)).cuda()
    ex = {"points": pts, "token": ["f0", "f1"], "batch_size": 2}
    d1 = model(ex)
    d2 = model(ex)
    assert set(d1) == {"f0", "f1"}
    for k in d1:
        assert d1[k]["box3d_lidar"].shape[1] == 9 and d1[k]["box3d_lidar"].shape[0] <= 83
        assert torch.equal(d1[k]["box3d_lidar"], d2[k]["box3d_lidar"]) and torch.equal(d1[k]["scores"], d2[k]["scores"])


def test_fused_inference_graph_matches_module_graph():
    """BN folding + fused HIP epilogues + merged SepHead branches == the module-by-module network (bf16 tolerance)."""
    from pillarnext_amd import synth
    from pillarnext_amd.models import FusedPillarNeXt, build_pillarnext_b

    cfg = synth.CONFIGS["C1"]
    torch.manual_seed(1)
    torch.backends.cudnn.deterministic = True
    model = build_pillarnext_b(cfg["pc_range"], cfg["voxel_size"], tasks=[["car"], ["truck", "bus"]], with_iou_head=True).cuda().eval()
    with torch.no_grad():  # non-trivial BN statistics everywhere
        for m in model.modules():
            if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
                m.running_mean.uniform_(-0.2, 0.2)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.uniform_(0.6, 1.4)
                m.bias.uniform_(-0.2, 0.2)
    fused = FusedPillarNeXt(model).cuda().eval()
    pts = torch.from_numpy(synth.make_batch("C1", 2, "sweep", n=30_000)).cuda()
    # reference side: the module-by-module graph in FP32 (no oracle exists for the spconv backbone, SURVEY 8c: the masked-dense
    # modules in full precision are the statement the fused bf16 graph is held against), stage by stage
    taps, ref_taps = {}, {}
    with torch.no_grad():
        ny, nx = (int(v) for v in model.reader.grid_size)
        occ = torch.empty((2, ny, nx), dtype=torch.uint8, device="cuda")
        canvas = model.reader.forward_dense(pts, 2, dtype=torch.float32, occupancy=occ)
        x, mask = canvas, occ.unsqueeze(1).float()
        for si, blk in enumerate(model.backbone.blocks):
            x, mask = blk(x, mask)
            ref_taps[f"stage{si}"] = x
        x = torch.relu(model.backbone.mapping[1](model.backbone.mapping[0](x), mask)) * mask
        ref_taps["neck"] = model.neck(x)
        ref = model.head(ref_taps["neck"])
        got = fused.forward_preds(pts, 2, taps=taps)
    # bf16 storage between ~30 layers.  Bounds = 1.25 x what an MI355X run of round 3 measured (printed below; run with -s): stages 0.0031 /
    # 0.0028 / 0.0027 / 0.0028, neck 0.0037; head maps: relative L2 <= 0.0103, max error <= 0.0136 of the map's scale
    bound = {"stage0": 0.0039, "stage1": 0.0035, "stage2": 0.0034, "stage3": 0.0035, "neck": 0.0047}
    for k, tol in bound.items():
        a, b = ref_taps[k].float(), taps[k].float()
        rel = ((a - b).norm() / (a.norm() + 1e-6)).item()
        print(f"[fused-vs-module] {k}: rel {rel:.5f} (bound {tol})")
        assert rel <= tol, (k, rel)
        assert bool(((a == 0) == (b == 0))[..., ::1].float().mean() > 0.97), k  # the same active-site pattern (exact zeros elsewhere)
    assert len(ref) == len(got) == 2
    for r, g in zip(ref, got):
        assert set(r) == set(g)
        for k in r:
            a, b = r[k].float(), g[k].float()
            assert a.shape == b.shape
            rel = ((a - b).norm() / (a.norm() + 1e-6)).item()
            err = (a - b).abs().max().item()
            scale = a.abs().max().item() + 1e-3
            print(f"[fused-vs-module] head {k}: rel {rel:.5f} max err {err:.5f} scale {scale:.4f}")
            assert rel <= 0.0128 and err <= 0.017 * scale, (k, rel, err, scale)
            assert torch.corrcoef(torch.stack([a.flatten(), b.flatten()]))[0, 1] > 0.997, k
    d = fused({"points": pts, "token": ["a", "b"], "batch_size": 2})
    assert set(d) == {"a", "b"} and d["a"]["box3d_lidar"].shape[1] == 9


def test_fused_graph_is_stateless_across_frames():
    """The persistent sparse-in-dense stage workspaces (row_dirty) must not leak one frame into the next: frame B after frame A
    through ONE FusedPillarNeXt == frame B through a fresh one (same kernels, so bit-equal), also with the workspaces disabled."""
    from pillarnext_amd import synth
    from pillarnext_amd.models import FusedPillarNeXt, build_pillarnext_b

    cfg = synth.CONFIGS["C1"]
    torch.manual_seed(3)
    torch.backends.cudnn.deterministic = True
    model = build_pillarnext_b(cfg["pc_range"], cfg["voxel_size"], tasks=[["car"], ["truck", "bus"]], with_iou_head=True).cuda().eval()
    frames = [torch.from_numpy(synth.make_batch("C1", 2, d, n=n, frame0=f)).cuda() for d, n, f in (("sweep", 30_000, 0), ("uniform", 4_000, 7),
                                                                                                  ("sweep", 20_000, 3))]
    seq = FusedPillarNeXt(model).cuda().eval()
    assert seq.sparse_ws
    outs_seq = [[p.clone() for p in _packed(seq, f)] for f in frames]
    assert seq._ws, "stage workspaces were not used"
    for i, f in enumerate(frames):
        fresh = FusedPillarNeXt(model).cuda().eval()
        for a, b in zip(outs_seq[i], _packed(fresh, f)):
            _same(a, b, f"frame {i}: workspace state leaked into the result")
    dense = FusedPillarNeXt(model).cuda().eval()
    dense.sparse_ws = False
    for a, b in zip(outs_seq[2], _packed(dense, frames[2])):
        _same(a, b, "sparse workspaces vs freshly allocated outputs")


def _same(a, b, what):
    # same kernels on the same inputs: normally bit-equal; the tolerance only absorbs a different MIOpen solver pick between two
    # module instances (a leaked row of stale activations would be an O(1) error)
    err = (a.float() - b.float()).abs().max().item()
    assert err <= 2e-2 * (b.float().abs().max().item() + 1.0), (what, err)


def _packed(fused, pts):
    packed = []
    with torch.no_grad():
        fused.forward_preds(pts, 2, packed_out=packed, lazy=False)
    torch.cuda.synchronize()
    return packed


def test_packed_hip_decoder_matches_reference_golden():
    """csrc/decode.hip + one sort + one batched NMS == the reference's CenterHead.predict on the golden head outputs."""
    from pillarnext_amd.decode import PackedDecoder

    g = load_golden("decode_2task")
    test_cfg = dict(post_center_limit_range=list(g["post_center_limit_range"]), score_threshold=float(g["score_threshold"]),
                    nms=dict(nms_pre_max_size=int(g["pre_max"]), nms_post_max_size=int(g["post_max"]), nms_iou_threshold=[[0.2], [0.2, 0.25]]),
                    out_size_factor=[int(v) for v in g["out_size_factor"]], voxel_size=list(g["voxel_size"]), pc_range=list(g["pc_range"]))
    packed = []
    for t, ncls in enumerate((1, 2)):
        parts = [g[f"t{t}_{k}"] for k in ("reg", "height", "dim", "rot", "vel", "iou", "hm")]
        x = np.concatenate(parts, axis=1)                                  # (B, 11+ncls, H, W)
        pad = (-x.shape[1]) % 8
        x = np.concatenate([x, np.zeros((x.shape[0], pad) + x.shape[2:], np.float32)], axis=1)
        packed.append(torch.from_numpy(x).cuda().contiguous(memory_format=torch.channels_last))
    dec = PackedDecoder([1, 2], [[0.5], [0.68, 0.2]], test_cfg, True, [p.shape[1] for p in packed])
    res = dec(packed, ["a", "b"])
    for i, r in enumerate(res):
        assert np.array_equal(r["label_preds"].numpy(), g[f"s{i}_labels"])
        np.testing.assert_allclose(r["scores"].numpy(), g[f"s{i}_scores"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(r["box3d_lidar"].numpy(), g[f"s{i}_boxes"], rtol=1e-4, atol=1e-4)


def test_head_loss_matches_reference():
    """CenterHead.loss (focal + L1 reg + DIoU + IoU-head loss with the HIP aligned IoU) vs the reference's loss and gradients."""
    from pillarnext_amd.models import CenterHead

    g = load_golden("head_loss_2task")
    tasks = [["car"], ["pedestrian", "cyclist"]]
    common = {"reg": (2, 2), "height": (1, 2), "dim": (3, 2), "rot": (2, 2), "vel": (2, 2), "iou": (1, 2)}
    head = CenterHead(16, tasks, 0.25, [1.0] * 6 + [0.2, 0.2, 1.0, 1.0], common, [2, 2], share_conv_channel=16, with_reg_iou=True,
                      voxel_size=list(g["voxel_size"]), pc_range=list(g["pc_range"]), out_size_factor=[4, 4]).cuda()
    preds, example = [], {k: [] for k in ("hm", "ind", "mask", "cat", "anno_box", "gt_boxes")}
    for t in range(2):
        preds.append({k: torch.from_numpy(g[f"t{t}_{k}"]).cuda().requires_grad_(True) for k in ("reg", "height", "dim", "rot", "vel", "iou", "hm")})
        for k in example:
            example[k].append(torch.from_numpy(g[f"t{t}_label_{k}"]).cuda())
    loss, rets = head.loss(example, preds)
    loss.backward()
    assert abs(loss.item() - float(g["total_loss"])) <= 1e-4 * abs(float(g["total_loss"])) + 1e-5
    for t, r in enumerate(rets):
        for k in ("loss", "hm_loss", "loc_loss", "iou_loss", "iou_reg_loss"):
            assert abs(float(r[k]) - float(g[f"t{t}_out_{k}"])) <= 1e-4 * abs(float(g[f"t{t}_out_{k}"])) + 1e-5, (t, k)
        np.testing.assert_allclose(r["loc_loss_elem"].numpy(), g[f"t{t}_out_loc_loss_elem"], rtol=1e-4, atol=1e-6)
        for k in ("reg", "height", "dim", "rot", "vel", "iou", "hm"):
            np.testing.assert_allclose(preds[t][k].grad.cpu().numpy(), g[f"t{t}_grad_{k}"], rtol=2e-3, atol=2e-6, err_msg=f"{t}/{k}")


def test_training_step_end_to_end():
    """One optimizer step of the whole detector in train mode: HIP voxelizer + scatter-max autograd -> masked-dense backbone ->
    ASPP (checkpointed) -> CenterHead.loss; every trainable parameter receives a finite gradient."""
    from pillarnext_amd import synth
    from pillarnext_amd.models import build_pillarnext_b

    cfg = synth.CONFIGS["C1"]
    torch.manual_seed(0)
    tasks = [["car"], ["pedestrian", "cyclist"]]
    model = build_pillarnext_b(cfg["pc_range"], cfg["voxel_size"], tasks=tasks).cuda().train()
    B, M = 2, 16
    pts = torch.from_numpy(synth.make_batch("C1", B, "sweep", n=8000)).cuda()
    H = W = 512 // 4
    gen = torch.Generator(device="cuda").manual_seed(3)
    ex = {"points": pts, "batch_size": B, "hm": [], "ind": [], "mask": [], "cat": [], "anno_box": [], "gt_boxes": []}
    for names in tasks:
        hm = torch.rand((B, len(names), H, W), device="cuda", generator=gen) * 0.2
        ind = torch.randint(0, H * W, (B, M), device="cuda", generator=gen)
        mask = torch.zeros((B, M), dtype=torch.uint8, device="cuda")
        mask[:, :6] = 1
        cat = torch.randint(0, len(names), (B, M), device="cuda", generator=gen)
        anno = torch.randn((B, M, 10), device="cuda", generator=gen) * 0.3
        gtb = torch.rand((B, M, 7), device="cuda", generator=gen) + torch.tensor([0, 0, -1, 1.5, 0.6, 1.2, 0], device="cuda")
        for k, v in zip(("hm", "ind", "mask", "cat", "anno_box", "gt_boxes"), (hm, ind, mask, cat, anno, gtb)):
            ex[k].append(v)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, betas=(0.9, 0.99), weight_decay=0.01)
    loss, rets = model(ex)
    assert torch.isfinite(loss) and len(rets) == 2
    opt.zero_grad()
    loss.backward()
    torch.nn.utils.clip_grad_norm_(model.parameters(), 35)
    missing = [n for n, p in model.named_parameters() if p.requires_grad and (p.grad is None or not torch.isfinite(p.grad).all())]
    assert not missing, missing[:5]
    opt.step()
    loss2, _ = model(ex)
    assert torch.isfinite(loss2)


def test_training_graph_on_the_fused_masked_bn_kernels(monkeypatch):
    """The training graph with the masked BatchNorm + residual + ReLU node on the HIP kernels (csrc/masked_bn.hip; channels_last maps) against
    the same detector with the node in its torch statement:
      fp32:  channels_last + HIP kernels (masked BN node; 3x3 convolutions on the three-bf16-product node, models._MaskedConv3x3F32Fn)  vs  NCHW + torch node +
             MIOpen fp32 convolutions -- same loss to 1e-5, every parameter gradient within 8 % norm-wise (this freshly initialised net amplifies ANY rounding difference ~1e4-1e5 x: what is left
             is MIOpen running other fp32 solvers for NHWC than for NCHW, measured 0.5-2 %, and the three-product node's 4e-6 per layer, measured 2.5-3.9 %)
      bf16:  autocast + channels_last, HIP kernels  vs  the torch node in the same graph -- loss within 0.5 %, gradients within the run-to-run
             spread of two bf16 graphs that round differently (measured values are printed with -s)
    and, for the record, bf16 against fp32 (a freshly initialised 30-layer net with batch statistics over two frames amplifies bf16 rounding:
    the loss agrees to 2 %, gradient directions to cos ~0.8; reported, bounded loosely).  Gradients whose norm is below 1e-3 of the largest one
    (convolution biases in front of a BatchNorm: exactly zero in exact arithmetic) are skipped."""
    import copy

    from pillarnext_amd import synth
    from pillarnext_amd.models import build_pillarnext_b

    cfg = synth.CONFIGS["C1"]
    torch.manual_seed(0)
    tasks = [["car"], ["pedestrian", "cyclist"]]
    ref = build_pillarnext_b(cfg["pc_range"], cfg["voxel_size"], tasks=tasks).cuda().train()
    B, M = 2, 16
    pts = torch.from_numpy(synth.make_batch("C1", B, "sweep", n=12000)).cuda()
    H = W = 512 // 4
    gen = torch.Generator(device="cuda").manual_seed(3)
    ex = {"points": pts, "batch_size": B, "hm": [], "ind": [], "mask": [], "cat": [], "anno_box": [], "gt_boxes": []}
    for names in tasks:
        ex["hm"].append(torch.rand((B, len(names), H, W), device="cuda", generator=gen) * 0.2)
        ex["ind"].append(torch.randint(0, H * W, (B, M), device="cuda", generator=gen))
        m = torch.zeros((B, M), dtype=torch.uint8, device="cuda")
        m[:, :6] = 1
        ex["mask"].append(m)
        ex["cat"].append(torch.randint(0, len(names), (B, M), device="cuda", generator=gen))
        ex["anno_box"].append(torch.randn((B, M, 10), device="cuda", generator=gen) * 0.3)
        ex["gt_boxes"].append(torch.rand((B, M, 7), device="cuda", generator=gen) + torch.tensor([0, 0, -1, 1.5, 0.6, 1.2, 0], device="cuda"))

    def run(model, amp):
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
            loss, _ = model(ex)
        loss.backward()
        return float(loss), {n: p.grad.double().flatten() for n, p in model.named_parameters() if p.grad is not None}

    def compare(tag, ga, gb):
        big = max(float(v.norm()) for v in ga.values())
        worst_rel, worst_cos = 0.0, 1.0
        for n, a in ga.items():
            b = gb[n]
            assert bool(torch.isfinite(b).all()), n
            if float(a.norm()) < 1e-3 * big:
                continue
            rel = float((a - b).norm()) / float(a.norm())
            cos = float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-30))
            worst_rel, worst_cos = max(worst_rel, rel), min(worst_cos, cos)
        va, vb = torch.cat([ga[n] for n in ga]), torch.cat([gb[n] for n in ga])   # all parameters as one vector: the stable statistic
        g_rel = float((va - vb).norm() / va.norm())
        g_cos = float(torch.dot(va, vb) / (va.norm() * vb.norm() + 1e-30))
        print(f"[train-graph] {tag}: worst relative L2 error of a gradient tensor {worst_rel:.4f}, worst cosine {worst_cos:.5f}; whole gradient: rel {g_rel:.4f}, cosine {g_cos:.5f}")
        return worst_rel, worst_cos, g_rel, g_cos

    monkeypatch.setenv("PNX_TRAIN_F32_HIP", "0")
    l_ref, g_ref = run(ref, False)                                            # fp32, NCHW: torch node, every convolution on MIOpen's fp32 kernels
    monkeypatch.delenv("PNX_TRAIN_F32_HIP")
    l_n, g_n = run(copy.deepcopy(ref).to(memory_format=torch.channels_last), False)   # fp32, channels_last: HIP kernels, 3x3 layers on the three-product node
    print(f"[train-graph] loss fp32 NCHW {l_ref:.6f}, fp32 channels_last/HIP {l_n:.6f}")
    assert abs(l_n - l_ref) <= 1e-5 * abs(l_ref)
    r, c, _, _ = compare("fp32 HIP vs fp32 torch", g_ref, g_n)
    assert r <= 0.08 and c >= 0.997
    l_a, g_a = run(copy.deepcopy(ref).to(memory_format=torch.channels_last), True)    # bf16 autocast: HIP kernels
    monkeypatch.setenv("PNX_MASKED_BN_HIP", "0")
    l_t, g_t = run(copy.deepcopy(ref).to(memory_format=torch.channels_last), True)    # bf16 autocast: torch node
    monkeypatch.delenv("PNX_MASKED_BN_HIP")
    print(f"[train-graph] loss bf16 HIP {l_a:.5f}, bf16 torch node {l_t:.5f}")
    assert abs(l_a - l_t) <= 5e-3 * abs(l_t) and abs(l_a - l_ref) <= 2e-2 * abs(l_ref)
    # bf16: a freshly initialised 30-layer net with batch statistics over two frames amplifies rounding differences chaotically; which solvers MIOpen's find
    # pass picked on this box and the HIP convolutions' summation order move the WORST gradient tensor's cosine anywhere between 0.45 and 0.8 (printed), for the
    # torch-node graph as for the HIP one.  What is asserted is the whole gradient vector: the HIP graph is as close to fp32 as the torch-node graph is, and the
    # two bf16 graphs are as close to each other as either is to fp32.
    _, _, gr, gc = compare("bf16 HIP vs bf16 torch", g_t, g_a)
    _, _, gr2, gc2 = compare("bf16 torch vs fp32", g_ref, g_t)
    _, _, gr3, gc3 = compare("bf16 HIP vs fp32", g_ref, g_a)
    assert gc3 >= gc2 - 0.05 and gr3 <= gr2 * 1.25 + 0.02, (gc3, gc2, gr3, gr2)
    assert gc >= min(gc2, gc3) - 0.05, (gc, gc2, gc3)


@pytest.mark.parametrize("layout,pre_max,n", [("mixed", 1000, 900_000), ("blocked", 1000, 900_000), ("blocked", 4096, 1_300_000), ("blocked", 83, 5_000),
                                               ("all_equal", 1000, 400_000), ("all_valid", 500, 300_000)])
def test_segmented_topk_equals_the_full_stable_sort(layout, pre_max, n):
    """pnx_decode_topk (exact radix select on (score key, index) -> collect -> LDS sort) must select and order exactly what one stable sort of
    all keys + the [:pre_max] cut gives, including ties (bf16 head outputs quantise the scores), segments with fewer / far more candidates
    than pre_max, an empty segment and a segment whose scores are ALL equal.  layout "mixed": every chunk holds keys of all 23 segments (the
    global-atomic overflow path of the histogram pass); "blocked": segments own contiguous key ranges, as pnx_decode_keys lays them out;
    "all_equal": every valid key of every segment carries the same score (the index digits decide everything); "all_valid": a fresh head."""
    from pillarnext_amd._lib import check, lib, ptr, stream_ptr

    L = lib()
    g = torch.Generator(device="cuda").manual_seed(4)
    S = 23
    if layout == "mixed":
        seg = torch.randint(0, S, (n,), device="cuda", generator=g)
    else:                                                                    # (sample, task) blocks of two classes each, like the head's maps
        blk = torch.arange(n, device="cuda") * 12 // n
        seg = (blk * 2 + torch.randint(0, 2, (n,), device="cuda", generator=g)).clamp(max=S - 1)
    seg[seg == 5] = 6                                                        # an empty segment
    sc = torch.rand((n,), device="cuda", generator=g) * 0.9 + 0.1
    sc = sc.to(torch.bfloat16).float()                                       # heavy ties
    sc[seg == 7] = 0.5                                                       # one segment: ALL scores equal (tens of thousands of ties)
    if layout == "all_equal":
        sc[:] = 0.25
    few = (seg == 9).nonzero().flatten()
    valid = torch.rand((n,), device="cuda", generator=g) < (1.1 if layout in ("all_valid", "all_equal") else 0.08)
    valid[few[40:]] = False                                                  # a segment with fewer candidates than pre_max
    low = (0xFFFFFFFF - sc.view(torch.int32).to(torch.int64))
    keys = torch.where(valid, (seg.to(torch.int64) << 32) | low, torch.full_like(low, -1))
    # reference: stable sort as unsigned
    skeys, order = torch.sort(keys ^ (-0x8000000000000000), stable=True)
    skeys = skeys ^ (-0x8000000000000000)
    bounds = (torch.arange(S + 1, device="cuda", dtype=torch.int64) << 32) ^ (-0x8000000000000000)
    st = torch.searchsorted(skeys ^ (-0x8000000000000000), bounds)
    tot = st[1:] - st[:-1]
    ln = torch.clamp(tot, max=pre_max)
    out_k = torch.empty((S * pre_max,), dtype=torch.int64, device="cuda")
    out_o = torch.empty((S * pre_max,), dtype=torch.int64, device="cuda")
    out_s = torch.empty((S,), dtype=torch.int64, device="cuda")
    out_l = torch.empty((S,), dtype=torch.int32, device="cuda")
    out_t = torch.empty((S,), dtype=torch.int32, device="cuda")
    ws = torch.empty(int(L.pnx_decode_topk_workspace_bytes(n, S)) + 256, dtype=torch.uint8, device="cuda")
    for rep in range(2):                                                     # the workspace is reused from call to call
        check(L.pnx_decode_topk(ptr(keys), n, S, pre_max, ptr(out_k), ptr(out_o), ptr(out_s), ptr(out_l), ptr(out_t), ptr(ws), ws.numel(), stream_ptr()),
              "pnx_decode_topk")
        assert torch.equal(out_l.long(), ln) and int(ln[5]) == 0 and int(ln[9]) <= 40
        assert torch.equal(out_t.long(), tot)
        for s in range(S):
            k = int(ln[s])
            a0 = int(st[s])
            assert torch.equal(out_o[s * pre_max: s * pre_max + k], order[a0: a0 + k]), (rep, s)
            assert torch.equal(out_k[s * pre_max: s * pre_max + k], skeys[a0: a0 + k]), (rep, s)
            assert int(out_s[s]) == s * pre_max


@pytest.mark.parametrize("S", [23, 31, 32, 80])
def test_bit_ranged_key_sort_equals_the_full_stable_sort(S):
    """pnx_sort_keys sorts only the 32 + bit_length(S) bits that can differ; keys, order (stability under heavy ties) and the place of the
    invalid keys must equal one generic stable 64-bit sort.  S = 31 / 32: the all-ones segment pattern just above / one bit above."""
    from pillarnext_amd._lib import check, lib, ptr, stream_ptr

    L = lib()
    g = torch.Generator(device="cuda").manual_seed(S)
    n = 700_001
    seg = torch.randint(0, S, (n,), device="cuda", generator=g)
    sc = (torch.rand((n,), device="cuda", generator=g) * 0.9 + 0.1).to(torch.bfloat16).float()        # heavy ties
    valid = torch.rand((n,), device="cuda", generator=g) < 0.5
    low = (0xFFFFFFFF - sc.view(torch.int32).to(torch.int64))
    keys = torch.where(valid, (seg.to(torch.int64) << 32) | low, torch.full_like(low, -1))
    want_k, want_o = torch.sort(keys ^ (-0x8000000000000000), stable=True)
    want_k = want_k ^ (-0x8000000000000000)
    got_k, got_o = torch.empty_like(keys), torch.empty_like(keys)
    ws = torch.empty(int(L.pnx_sort_keys_workspace_bytes(n)), dtype=torch.uint8, device="cuda")
    check(L.pnx_sort_keys(ptr(keys), n, S, ptr(got_k), ptr(got_o), ptr(ws), ws.numel(), stream_ptr()), "pnx_sort_keys")
    assert torch.equal(got_k, want_k) and torch.equal(got_o, want_o)


def test_decoder_sort_paths_agree(monkeypatch):
    """PackedDecoder with pnx_sort_keys (default) and with the generic torch.sort: identical detections."""
    from pillarnext_amd.decode import PackedDecoder

    g = torch.Generator(device="cuda").manual_seed(11)
    B, H, W = 2, 64, 64
    cfg = dict(nms=dict(nms_pre_max_size=200, nms_post_max_size=40, nms_iou_threshold=[[0.2], [0.2, 0.2]]), score_threshold=0.1,
               pc_range=[-25.6, -25.6], voxel_size=[0.1, 0.1], out_size_factor=[8, 8], post_center_limit_range=[-30, -30, -10, 30, 30, 10])
    packed = [(torch.randn((B, 16, H, W), device="cuda", generator=g) * 0.7).to(torch.bfloat16).contiguous(memory_format=torch.channels_last) for _ in range(2)]
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("PNX_DECODE_TORCH_SORT", mode)
        dec = PackedDecoder([1, 2], [[0.5], [0.5, 0.5]], cfg, True, [16, 16])
        res[mode] = dec(packed)
    for a, b in zip(res["0"], res["1"]):
        assert a["box3d_lidar"].shape[0] > 0
        for k in ("box3d_lidar", "scores", "label_preds"):
            assert torch.equal(a[k], b[k]), k


def test_launch_plans_equal_the_python_launch_loop(monkeypatch):
    """FusedPillarNeXt with launch plans (plan.py / pnx_enqueue: backbone and head as one C call each, pnx_decode_lazy_enqueue: the decoder as
    one) against the same network issuing every launch from Python: detections bit for bit, in a serving-loop order (launch i + 1 before
    the result of i is read), over batches whose active sets differ (the persistent workspaces go stale in between)."""
    from pillarnext_amd import synth
    from pillarnext_amd.models import FusedPillarNeXt, build_pillarnext_b

    cfg = synth.CONFIGS["C1"]
    torch.manual_seed(5)
    model = build_pillarnext_b(cfg["pc_range"], cfg["voxel_size"], tasks=[["car"], ["truck", "bus"]], with_iou_head=True).cuda().eval()
    monkeypatch.setenv("PNX_PLAN", "0")
    plain = FusedPillarNeXt(model).cuda().eval()
    monkeypatch.setenv("PNX_PLAN", "1")
    fused = FusedPillarNeXt(model).cuda().eval()
    assert fused.use_plan and not plain.use_plan and fused._plan_ok() and fused._head_plan_ok()
    B = 2
    exs = []
    for i, (d, n, f) in enumerate((("sweep", 30_000, 0), ("uniform", 20_000, 5), ("sweep", 25_000, 9), ("sweep", 30_000, 2))):
        exs.append({"points": torch.from_numpy(synth.make_batch("C1", B, d, n=n, frame0=f)).cuda(), "batch_size": B, "token": [f"f{i}a", f"f{i}b"]})
    with torch.no_grad():
        want = [plain(e) for e in exs]
        got, pend = [], None
        for e in exs:
            nxt = fused.forward_async(e)
            if pend is not None:
                got.append(fused.detections(pend.result()))
            pend = nxt
        got.append(fused.detections(pend.result()))
    assert ("plan_bb", B, exs[0]["points"].device) in fused._ws and any(k[0] == "lazy_fused" for k in fused.decoder()._dev)
    assert any(isinstance(k, tuple) and k[0] == "plan_head" for k in fused._ws)
    assert sum(len(v["scores"]) for a in want for v in a.values()) > 0
    for a, b in zip(got, want):
        assert set(a) == set(b)
        for k in a:
            assert torch.equal(a[k]["scores"], b[k]["scores"]) and torch.equal(a[k]["box3d_lidar"], b[k]["box3d_lidar"]), k
            assert torch.equal(a[k]["label_preds"], b[k]["label_preds"]), k

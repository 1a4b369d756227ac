#!/usr/bin/env python3
"""gen_golden.py -- TEST INFRASTRUCTURE.  Generates tests/golden/*.npz by RUNNING THE REFERENCE.

Run only in the build container (needs /root/reference); the GPU box receives the .npz fixtures,
never reference code.  Fixtures are pure data: inputs and the reference's outputs.

What runs the reference, and with which third-party restatements:
  * reader     : det3d/models/readers/pillar_encoder.py imported unmodified.  It imports
                 `torch_scatter` (third-party CUDA/C++ package, un-vendored, version UNPINNED in
                 docker/Dockerfile:16).  We restate its two published ops on torch primitives:
                 scatter_max(src,index,dim=0)[0] = per-index maximum, scatter_mean = per-index
                 sum / count.  Call sites: pillar_encoder.py:43,113,180.
  * IoU        : det3d/core/iou3d_nms/src/iou3d_cpu.cpp compiled in place (oracle/_ref).
  * NMS        : the reference has no CPU NMS; reference CPU IoU matrix + the greedy rule of
                 iou3d_nms.cpp:144-155.
  * decode     : det3d/models/heads/centerhead.py CenterHead.predict imported unmodified, with
                 `det3d.core.iou3d_nms.iou3d_nms_cuda.nms_gpu` provided by the NMS above and
                 `numba` (imported by det3d/core/__init__.py only for CPU augmentation code that is
                 never called here) replaced by identity decorators.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from pillarnext_amd import synth  # noqa: E402

REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")


# --------------------------------------------------------------------------- third-party restatements
def install_torch_scatter():
    m = types.ModuleType("torch_scatter")

    def scatter_max(src, index, dim=0):
        n = int(index.max()) + 1
        idx = index.view(-1, 1).expand_as(src)
        out = torch.full((n, src.shape[1]), float("-inf"), dtype=src.dtype).scatter_reduce(0, idx, src, "amax", include_self=True)
        return out, None

    def scatter_mean(src, index, dim=0):
        n = int(index.max()) + 1
        idx = index.view(-1, 1).expand_as(src)
        s = torch.zeros((n, src.shape[1]), dtype=src.dtype).scatter_reduce(0, idx, src, "sum", include_self=True)
        cnt = torch.zeros((n,), dtype=src.dtype).scatter_reduce(0, index, torch.ones_like(index, dtype=src.dtype), "sum")
        return s / cnt.clamp(min=1).view(-1, 1)

    m.scatter_max, m.scatter_mean = scatter_max, scatter_mean
    sys.modules["torch_scatter"] = m


def install_numba_identity():
    m = types.ModuleType("numba")

    def _id(*a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return lambda f: f

    m.njit = m.jit = _id
    sys.modules["numba"] = m


def install_iou3d_nms():
    pkg = types.ModuleType("det3d.core.iou3d_nms")
    pkg.__path__ = [os.path.join(REF, "det3d/core/iou3d_nms")]
    ext = types.ModuleType("det3d.core.iou3d_nms.iou3d_nms_cuda")

    def nms_gpu(boxes, keep, thresh):
        k = O.ref_nms_rotated(boxes.detach().cpu().numpy(), float(thresh))
        keep[: len(k)] = torch.from_numpy(k)
        return len(k)

    ext.nms_gpu = nms_gpu
    pkg.iou3d_nms_cuda = ext
    sys.modules["det3d.core.iou3d_nms"] = pkg
    sys.modules["det3d.core.iou3d_nms.iou3d_nms_cuda"] = ext
    torch.Tensor.cuda = lambda self, *a, **k: self


# --------------------------------------------------------------------------- reader fixtures
def edge_rows(pc_range, voxel_size, batch_idx):
    """Boundary / NaN / -0.0 / out-of-range rows."""
    x0, y0, x1, y1 = pc_range[0], pc_range[1], pc_range[3], pc_range[4]
    vs = voxel_size[0]
    f32 = np.float32
    rows = [
        [x0, y0, 0.0], [np.nextafter(f32(x0), f32(-1e9)), y0, 0.1], [x1, 0.0, 0.2], [np.nextafter(f32(x1), f32(-1e9)), 0.0, 0.3],
        [0.0, y1, 0.0], [0.0, np.nextafter(f32(y1), f32(-1e9)), 0.0], [np.nan, 0.0, 0.0], [0.0, np.nan, 0.0],
        [x0 + 3 * vs, y0 + 5 * vs, 0.5], [x0 + f32(3) * f32(vs), y0 + 5 * vs, 0.6], [x0 - 0.0, y0, 99.0], [1e9, 0.0, 0.0],
        [-1e9, 0.0, 0.0], [x0 + 0.5 * vs, y0 + 0.5 * vs, -77.0], [x1 - 1e-4, y1 - 1e-4, 0.0], [0.0, 0.0, 0.0], [-0.0, -0.0, -0.0],
    ]
    # points sitting exactly on (computed) pillar edges all over the grid
    for k in (1, 7, 100, 333, 671):
        rows.append([x0 + k * vs, y0 + k * vs, 0.0])
        rows.append([f32(x0) + f32(k) * f32(vs), f32(y0) + f32(k) * f32(vs), 0.0])
    r = np.zeros((len(rows), 6), np.float32)
    r[:, 0] = batch_idx
    r[:, 1:4] = np.asarray(rows, np.float32)
    r[:, 4] = 0.5
    r[:, 5] = 0.1
    return r


def reader_case(name, pc_range, voxel_size, clouds, num_filters=(64, 64), seed=0, train=False):
    from det3d.models.readers.pillar_encoder import PillarFeatureNet

    pts = np.concatenate(clouds).astype(np.float32)
    F = pts.shape[1] - 1
    layers = synth.pfn_params(F, num_filters, seed)
    net = PillarFeatureNet(F, list(num_filters), list(voxel_size), list(pc_range))
    with torch.no_grad():
        for L, pfn in zip(layers, net.pfn_layers):
            pfn.linear.weight.copy_(torch.from_numpy(L["W"]))
            pfn.norm.weight.copy_(torch.from_numpy(L["gamma"]))
            pfn.norm.bias.copy_(torch.from_numpy(L["beta"]))
            pfn.norm.running_mean.copy_(torch.from_numpy(L["mean"]))
            pfn.norm.running_var.copy_(torch.from_numpy(L["var"]))
    tp = torch.from_numpy(pts)
    net.eval()
    with torch.no_grad():
        feats, coords_v, unq_inv, grid_v = net.voxelization(tp)
        feat_max, coords, grid = net(tp)
    assert torch.equal(coords, coords_v)
    out = dict(points=pts, pc_range=np.asarray(pc_range, np.float64), voxel_size=np.asarray(voxel_size, np.float64),
               num_filters=np.asarray(num_filters, np.int64), coords=coords.numpy().astype(np.int32),
               unq_inv=unq_inv.numpy().astype(np.int64), features=feats.numpy(), feat_max=feat_max.numpy(),
               grid=np.asarray(grid, np.int64), eps=np.float32(1e-3))
    for i, L in enumerate(layers):
        for k, v in L.items():
            out[f"l{i}_{k}"] = v
    if train:
        # train mode: batch statistics, running-stat update, and gradients for a fixed upstream grad
        net.train()
        for p in net.parameters():
            p.grad = None
        fm_t, _, _ = net(tp)
        g = torch.from_numpy(np.random.default_rng(seed + 7).standard_normal(tuple(fm_t.shape)).astype(np.float32))
        fm_t.backward(g)
        out["train_feat_max"] = fm_t.detach().numpy()
        out["train_upstream_grad"] = g.numpy()
        for i, pfn in enumerate(net.pfn_layers):
            out[f"train_l{i}_dW"] = pfn.linear.weight.grad.numpy().copy()
            out[f"train_l{i}_dgamma"] = pfn.norm.weight.grad.numpy().copy()
            out[f"train_l{i}_dbeta"] = pfn.norm.bias.grad.numpy().copy()
            out[f"train_l{i}_running_mean"] = pfn.norm.running_mean.numpy().copy()
            out[f"train_l{i}_running_var"] = pfn.norm.running_var.numpy().copy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"[golden] {name}: N={len(pts)} N'={len(out['unq_inv'])} P={len(out['coords'])} grid={out['grid']}")


def gen_reader():
    rng = np.random.default_rng(42)
    # (1) nuScenes reference-YAML geometry, B=2, uniform + clustered + edge rows, one fat pillar (>64 pts)
    pr, vs = synth.CONFIGS["C2ref"]["pc_range"], synth.CONFIGS["C2ref"]["voxel_size"]
    a = synth.uniform_cloud(1500, pr, 1000, 0)
    b = synth.sweep_cloud(1500, pr, 1001, 1)
    fat = np.zeros((150, 6), np.float32)
    fat[:, 0] = 1
    fat[:, 1] = 10.0 + rng.uniform(0, 0.07, 150)
    fat[:, 2] = -7.0 + rng.uniform(0, 0.07, 150)
    fat[:, 3] = rng.uniform(-3, 1, 150)
    fat[:, 4] = rng.uniform(0, 1, 150)
    clouds = [a, edge_rows(pr, vs, 0), b, fat, edge_rows(pr, vs, 1)]
    perm = rng.permutation(sum(len(c) for c in clouds))
    pts = np.concatenate(clouds)[perm]
    reader_case("reader_nusc_b2", pr, vs, [pts], train=True)
    # (2) BASELINE geometry (1440^2), single sample, dense little patch so pillars hold many points
    pr, vs = synth.CONFIGS["C2"]["pc_range"], synth.CONFIGS["C2"]["voxel_size"]
    c = synth.uniform_cloud(2500, pr, 1002, 0)
    d = np.zeros((1200, 6), np.float32)
    d[:, 1:3] = rng.uniform(-1.2, 1.2, (1200, 2)) + np.array([20.0, -13.0])
    d[:, 3] = rng.uniform(-2, 1, 1200)
    d[:, 4:6] = rng.uniform(0, 1, (1200, 2))
    reader_case("reader_c2_b1", pr, vs, [c, d, edge_rows(pr, vs, 0)])
    # (3) Tiny config C1 (0.2 m, 512^2): larger pillars, B=3 with an EMPTY middle sample (b=1 absent)
    pr, vs = synth.CONFIGS["C1"]["pc_range"], synth.CONFIGS["C1"]["voxel_size"]
    e = synth.sweep_cloud(3000, pr, 1003, 0)
    f = synth.sweep_cloud(2000, pr, 1004, 2)
    reader_case("reader_c1_b3_gap", pr, vs, [e, f, edge_rows(pr, vs, 2)])
    # (4) Waymo reference-YAML geometry (2048^2), F=5
    pr, vs = synth.CONFIGS["C5ref"]["pc_range"], synth.CONFIGS["C5ref"]["voxel_size"]
    g = synth.sweep_cloud(4000, pr, 1005, 0, beams=64, sweeps=3)
    reader_case("reader_waymo_b1", pr, vs, [g, edge_rows(pr, vs, 0)])
    # (5) all points out of range -> P = 0 is NOT supported by the reference (index.max() of empty);
    #     instead: a single point
    pr, vs = synth.CONFIGS["C2ref"]["pc_range"], synth.CONFIGS["C2ref"]["voxel_size"]
    one = np.array([[0, 1.0, 2.0, 0.5, 0.3, 0.0], [0, 1e6, 0.0, 0.0, 0.0, 0.0]], np.float32)
    reader_case("reader_single_point", pr, vs, [one])


def gen_reader_train():
    """Round 2: two more TRAIN-mode fixtures for the fused training PFN (forward with batch statistics, running statistics,
    gradients of both Linear weights and both BatchNorm affines for a fixed upstream gradient)."""
    rng = np.random.default_rng(43)
    pr, vs = synth.CONFIGS["C1"]["pc_range"], synth.CONFIGS["C1"]["voxel_size"]
    # (6) C1 geometry (0.2 m pillars), one sample, a pillar with 90 points (> 64: beyond one wave's lanes) and a few of 33..40
    a = synth.sweep_cloud(2500, pr, 1010, 0)
    fat = np.zeros((90, 6), np.float32)
    fat[:, 1] = 12.0 + rng.uniform(0, 0.19, 90)
    fat[:, 2] = 4.0 + rng.uniform(0, 0.19, 90)
    fat[:, 3] = rng.uniform(-3, 1, 90)
    fat[:, 4:6] = rng.uniform(0, 1, (90, 2))
    mid = np.zeros((110, 6), np.float32)
    mid[:, 1] = -20.0 + 0.2 * rng.integers(0, 3, 110) + rng.uniform(0, 0.19, 110)
    mid[:, 2] = 8.0 + rng.uniform(0, 0.19, 110)
    mid[:, 3] = rng.uniform(-3, 1, 110)
    mid[:, 4:6] = rng.uniform(0, 1, (110, 2))
    pts = np.concatenate([a, fat, mid])
    reader_case("reader_c1_train_fat", pr, vs, [pts[rng.permutation(len(pts))]], train=True, seed=1)
    # (7) B = 3 with an EMPTY middle sample (b = 1 absent), train mode
    e = synth.sweep_cloud(1800, pr, 1011, 0)
    f = synth.uniform_cloud(1200, pr, 1012, 2)
    pts = np.concatenate([e, f, edge_rows(pr, vs, 2)])
    reader_case("reader_c1_b3_gap_train", pr, vs, [pts[rng.permutation(len(pts))]], train=True, seed=2)


# --------------------------------------------------------------------------- IoU / NMS fixtures
def special_boxes():
    b = [
        [0, 0, 0, 4, 2, 1.5, 0.0], [0, 0, 0, 4, 2, 1.5, 0.0],  # identical
        [0, 0, 0.2, 4, 2, 1.5, np.pi / 2], [0, 0, 0, 2, 1, 1.0, 0.3],  # 90 deg rotated / contained
        [30, 30, 0, 4, 2, 1.5, 0.7], [4, 0, 0, 4, 2, 1.5, 0.0],  # disjoint / edge-touching
        [2, 0, 0, 4, 2, 1.5, 0.0], [2, 1, 0, 4, 2, 1.5, 1e-3],  # half overlap / near-collinear edges
        [0, 0, 0, 4, 2, 1.5, np.pi], [0, 0, 0, 4, 2, 1.5, -np.pi],  # same box, heading +-pi
        [0.5, 0.5, 0, 3, 3, 1.0, np.pi / 4], [0.5, 0.5, 0, 3, 3, 1.0, 0.0],  # octagon intersection
        [1, 1, 0, 1e-3, 1e-3, 1.0, 0.2], [0, 0, 0, 100, 100, 3, 2.0],  # tiny / huge
        [0, 0, 0, 4, 2, 1.5, 7.5], [0, 0, 0, 4, 2, 1.5, -12.25],  # headings outside [-pi, pi]
    ]
    return np.asarray(b, np.float32)


def min_margin(iou, thr):
    n = iou.shape[0]
    iu = np.triu_indices(n, 1)
    return np.abs(iou[iu] - np.float32(thr)).min()


def gen_iou_nms():
    rng = np.random.default_rng(7)
    sp = special_boxes()
    rnd, _ = synth.clustered_boxes(48, 11, spread=6.0, n_clusters=5)
    boxes = np.concatenate([sp, rnd])[:64]
    iou = O.ref_boxes_iou_bev(boxes, boxes)
    b2, _ = synth.clustered_boxes(40, 12, spread=6.0, n_clusters=4)
    iou_ab = O.ref_boxes_iou_bev(boxes, b2)
    al = O.ref_boxes_aligned_iou_bev(boxes[:40], b2)
    np.savez_compressed(os.path.join(OUT, "iou_bev_64.npz"), boxes_a=boxes, boxes_b=b2, iou_aa=iou, iou_ab=iou_ab, iou_aligned=al)
    print("[golden] iou_bev_64: nonzero frac", (iou > 0).mean())

    cases = {}
    for name, n, thr, seed in [("n256_t020", 256, 0.2, 21), ("n256_t070", 256, 0.7, 22), ("n1000_t020", 1000, 0.2, 23),
                               ("n1000_t025", 1000, 0.25, 24), ("n130_t020", 130, 0.2, 25)]:
        # re-draw until every pair IoU is >= 1e-4 away from the threshold
        s = seed
        while True:
            boxes, scores = synth.clustered_boxes(n, s)
            m = O.ref_boxes_iou_bev(boxes, boxes)
            if min_margin(m, thr) >= 1e-4:
                break
            s += 1000
        keep = O.ref_nms_rotated(boxes, thr)
        cases[name + "_boxes"] = boxes
        cases[name + "_scores"] = scores
        cases[name + "_thr"] = np.float32(thr)
        cases[name + "_keep"] = keep
        print(f"[golden] nms {name}: kept {len(keep)} of {n} (seed {s}, margin {min_margin(m, thr):.2e})")
    np.savez_compressed(os.path.join(OUT, "nms_rotated.npz"), **cases)


# --------------------------------------------------------------------------- decode fixture
def ns(**k):
    return types.SimpleNamespace(**k)


def gen_decode(name="decode_2task", bf16_reg=False):
    """bf16_reg: the five regression maps are rounded to bf16 before the reference sees them -- the lazy head (decode.PackedDecoder.
    launch_lazy + k_sephead_lazy) produces its regression values in bf16, so a test can reproduce these maps EXACTLY with identity
    convolutions and compare its detections with the reference's on equal inputs."""
    from det3d.models.heads.centerhead import CenterHead

    torch.manual_seed(3)
    tasks = [["car"], ["truck", "construction_vehicle"]]
    common = {"reg": (2, 2), "height": (1, 2), "dim": (3, 2), "rot": (2, 2), "vel": (2, 2), "iou": (1, 2)}
    pc_range = [-50.4, -50.4, -5.0, 50.4, 50.4, 3.0]
    voxel = [0.075, 0.075, 8]
    head = CenterHead(in_channels=16, tasks=tasks, weight=0.25, code_weights=[1.0] * 10, common_heads=common, strides=[2, 2],
                      share_conv_channel=16, rectifier=[[0.5], [0.68, 0.2]])
    B, H, W = 2, 24, 24
    rng = np.random.default_rng(5)
    preds = []
    for t, names in enumerate(tasks):
        d = {}
        for k, c in [("reg", 2), ("height", 1), ("dim", 3), ("rot", 2), ("vel", 2), ("iou", 1), ("hm", len(names))]:
            a = rng.standard_normal((B, c, H, W)).astype(np.float32)
            if k == "hm":
                a = a * 2.0 - 1.5
            if k == "dim":
                a = a * 0.3 + 0.8
            if k == "reg":
                a = rng.uniform(0, 1, (B, c, H, W)).astype(np.float32)
            if k == "iou":
                a = np.clip(a * 0.7, -1.3, 1.3)
            if bf16_reg and k not in ("hm", "iou"):
                a = torch.from_numpy(a).to(torch.bfloat16).float().numpy()
            d[k] = a
        # make the grid spread over real-world metres: out_size_factor 56 -> 24*56*0.075 = 100.8 m
        preds.append(d)
    test_cfg = ns(post_center_limit_range=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0], score_threshold=0.1,
                  nms=ns(nms_pre_max_size=1000, nms_post_max_size=83, nms_iou_threshold=[[0.2], [0.2, 0.25]]),
                  out_size_factor=[56, 56], voxel_size=voxel, pc_range=pc_range)
    tp = [{k: torch.from_numpy(v.copy()) for k, v in d.items()} for d in preds]
    res = head.predict({"token": ["a", "b"]}, tp, test_cfg)
    out = {}
    for t, d in enumerate(preds):
        for k, v in d.items():
            out[f"t{t}_{k}"] = v
    for i, r in enumerate(res):
        out[f"s{i}_boxes"] = r["box3d_lidar"].numpy()
        out[f"s{i}_scores"] = r["scores"].numpy()
        out[f"s{i}_labels"] = r["label_preds"].numpy()
        print(f"[golden] decode sample {i}: {len(r['scores'])} boxes")
    out["rectifier"] = np.asarray([[0.5, 0.0], [0.68, 0.2]], np.float32)
    out["num_classes"] = np.asarray([1, 2], np.int64)
    out["nms_thr"] = np.asarray([[0.2, 0.0], [0.2, 0.25]], np.float32)
    out["out_size_factor"] = np.asarray([56, 56], np.int64)
    out["pc_range"] = np.asarray(pc_range, np.float64)
    out["voxel_size"] = np.asarray(voxel, np.float64)
    out["post_center_limit_range"] = np.asarray(test_cfg.post_center_limit_range, np.float64)
    out["score_threshold"] = np.float32(0.1)
    out["pre_max"] = np.int64(1000)
    out["post_max"] = np.int64(83)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)


def gen_loss():
    """CenterHead.loss of the reference (centerhead.py:142-229 + loss/centerloss.py) on random head outputs and labels,
    with gradients.  The native aligned-overlap op it calls (iou3d_nms_utils.py:75) is served by the pinned oracle."""
    from det3d.models.heads.centerhead import CenterHead

    ext = sys.modules["det3d.core.iou3d_nms.iou3d_nms_cuda"]

    def boxes_aligned_overlap_bev_gpu(a, b, out):
        out[:, 0] = torch.from_numpy(O.boxes_aligned_overlap_bev(a.detach().numpy(), b.detach().numpy(), "libm"))
        return 1

    ext.boxes_aligned_overlap_bev_gpu = boxes_aligned_overlap_bev_gpu
    torch.cuda.FloatTensor = torch.FloatTensor
    torch.manual_seed(11)
    tasks = [["car"], ["pedestrian", "cyclist"]]
    common = {"reg": (2, 2), "height": (1, 2), "dim": (3, 2), "rot": (2, 2), "vel": (2, 2), "iou": (1, 2)}
    pc_range, voxel = [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0], [0.2, 0.2, 8]
    head = CenterHead(in_channels=16, tasks=tasks, weight=0.25, code_weights=[1.0] * 6 + [0.2, 0.2, 1.0, 1.0], common_heads=common,
                      strides=[2, 2], share_conv_channel=16, with_reg_iou=True, voxel_size=voxel, pc_range=pc_range, out_size_factor=[4, 4])
    B, H, W, M = 2, 32, 32, 20
    rng = np.random.default_rng(17)
    out, preds, feeds, example = {}, [], [], {"hm": [], "ind": [], "mask": [], "cat": [], "anno_box": [], "gt_boxes": []}
    for t, names in enumerate(tasks):
        d = {}
        for k, c in [("reg", 2), ("height", 1), ("dim", 3), ("rot", 2), ("vel", 2), ("iou", 1), ("hm", len(names))]:
            a = (rng.standard_normal((B, c, H, W)) * 0.5).astype(np.float32)
            out[f"t{t}_{k}"] = a
            d[k] = torch.from_numpy(a.copy()).requires_grad_(True)
        preds.append(d)
        feeds.append({k: v * 1.0 for k, v in d.items()})  # non-leaf views: the reference applies sigmoid_ in place
        n_pos = [7, 12]
        ind = np.zeros((B, M), np.int64)
        mask = np.zeros((B, M), np.uint8)
        cat = np.zeros((B, M), np.int64)
        anno = np.zeros((B, M, 10), np.float32)
        gtb = np.zeros((B, M, 7), np.float32)
        hm = rng.uniform(0, 0.3, (B, len(names), H, W)).astype(np.float32)
        for b in range(B):
            cells = rng.choice(H * W, n_pos[b], replace=False)
            ind[b, : n_pos[b]] = cells
            mask[b, : n_pos[b]] = 1
            cat[b, : n_pos[b]] = rng.integers(0, len(names), n_pos[b])
            hm[b, cat[b, : n_pos[b]], cells // W, cells % W] = 1.0
            anno[b, : n_pos[b]] = rng.standard_normal((n_pos[b], 10)).astype(np.float32) * 0.4
            anno[b, 0, 6] = np.nan  # the reference tolerates NaN velocity targets (centerloss.py:55-56)
            cx = (cells % W + 0.5) * 4 * voxel[0] + pc_range[0]
            cy = (cells // W + 0.5) * 4 * voxel[1] + pc_range[1]
            gtb[b, : n_pos[b]] = np.stack([cx + rng.normal(0, 0.3, n_pos[b]), cy + rng.normal(0, 0.3, n_pos[b]), rng.normal(-1, 0.3, n_pos[b]),
                                           rng.uniform(1.5, 4.5, n_pos[b]), rng.uniform(0.6, 2.0, n_pos[b]), rng.uniform(1.2, 2.0, n_pos[b]),
                                           rng.uniform(-3, 3, n_pos[b])], 1).astype(np.float32)
        for k, v in [("hm", hm), ("ind", ind), ("mask", mask), ("cat", cat), ("anno_box", anno), ("gt_boxes", gtb)]:
            example[k].append(torch.from_numpy(v.copy()))
            out[f"t{t}_label_{k}"] = v
    loss, rets = head.loss(example, feeds)
    loss.backward()
    out["total_loss"] = np.float32(loss.item())
    for t, r in enumerate(rets):
        for k in ("loss", "hm_loss", "loc_loss", "iou_loss", "iou_reg_loss"):
            out[f"t{t}_out_{k}"] = np.float32(float(r[k]))
        out[f"t{t}_out_loc_loss_elem"] = r["loc_loss_elem"].numpy()
        for k in ("reg", "height", "dim", "rot", "vel", "iou", "hm"):
            out[f"t{t}_grad_{k}"] = preds[t][k].grad.numpy()
    out["pc_range"] = np.asarray(pc_range, np.float64)
    out["voxel_size"] = np.asarray(voxel, np.float64)
    np.savez_compressed(os.path.join(OUT, "head_loss_2task.npz"), **out)
    print(f"[golden] head_loss_2task: total {loss.item():.5f}")


# --------------------------------------------------------------------------- voxel / multi-view readers
def install_spconv_import_stub():
    """det3d/models/readers/mvf_encoder.py imports spconv (third-party, absent from the image) at module level and derives SingleView's
    blocks from it.  These stand-ins only let the FILE import; nothing of them is ever executed: the fixtures below come from
    PillarVoxelNet, CylinderNet, PointNet and SingleView.bilinear_interpolate, which are torch + torch_scatter alone.  What needs spconv
    (SingleView.forward, MVFFeatureNet.forward end to end) has no fixture -- unpinned, like the backbone."""
    sp = types.ModuleType("spconv")
    spt = types.ModuleType("spconv.pytorch")
    core = types.ModuleType("spconv.core")

    class _Never(torch.nn.Module):
        def __init__(self, *a, **k):
            raise RuntimeError("spconv stand-in: import only")

    spt.SparseModule = torch.nn.Module
    spt.SparseSequential = torch.nn.Sequential
    spt.SubMConv2d = spt.SparseConv2d = spt.SubMConv3d = spt.SparseConv3d = spt.SparseConvTensor = _Never
    core.ConvAlgo = types.SimpleNamespace(Native=0)
    sp.pytorch, sp.core = spt, core
    sys.modules["spconv"], sys.modules["spconv.pytorch"], sys.modules["spconv.core"] = sp, spt, core


def gen_voxel():
    from det3d.models.readers.voxel_encoder import VoxelFeatureNet

    rng = np.random.default_rng(51)
    pr, vs = [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0], [0.2, 0.2, 0.4]
    a = synth.sweep_cloud(3000, pr, 1020, 0)
    b = synth.uniform_cloud(2000, pr, 1021, 2)          # sample 1 absent
    b[:, 3] = rng.uniform(-6.0, 4.0, len(b))             # some rows outside the z range: this reader drops them (voxel_encoder.py:50-55)
    e = edge_rows(pr, vs, 0)
    e[:, 3] = np.where(np.arange(len(e)) % 3 == 0, -5.0, np.where(np.arange(len(e)) % 3 == 1, np.nextafter(np.float32(3.0), np.float32(-9)), 3.0))
    pts = np.concatenate([a, b, e]).astype(np.float32)
    pts = pts[rng.permutation(len(pts))]
    net = VoxelFeatureNet(vs, pr)
    with torch.no_grad():
        f, c, g = net(torch.from_numpy(pts))
        _, _, inv, _ = net.voxelization(torch.from_numpy(pts))
    np.savez_compressed(os.path.join(OUT, "voxel_b3_gap.npz"), points=pts, pc_range=np.asarray(pr, np.float64), voxel_size=np.asarray(vs, np.float64),
                        features=f.numpy(), coords=c.numpy().astype(np.int32), unq_inv=inv.numpy().astype(np.int64), grid=np.asarray(g, np.int64))
    print(f"[golden] voxel_b3_gap: N={len(pts)} V={len(c)} grid={g}")


def gen_mvf_parts():
    install_spconv_import_stub()
    from det3d.models.readers.mvf_encoder import CylinderNet, PillarVoxelNet, PointNet, SingleView

    rng = np.random.default_rng(52)
    pr, vs = [-76.8, -76.8, -10.0, 76.8, 76.8, 10.0], [0.075, 0.075, 20]
    cr, cs = [-180, -10.0, 0, 180, 10.0, 107], [0.140625, 0.2, 107]
    a = synth.sweep_cloud(2500, pr, 1030, 0, beams=64, sweeps=3)
    b = synth.uniform_cloud(1500, pr, 1031, 1)
    pts = np.concatenate([a, b]).astype(np.float32)
    r = np.asarray(pr, np.float32)
    keep = ((pts[:, 1] >= r[0]) & (pts[:, 1] < r[3]) & (pts[:, 2] >= r[1]) & (pts[:, 2] < r[4]) & (pts[:, 3] >= r[2]) & (pts[:, 3] < r[5]))
    pts = pts[keep]                                      # MVFFeatureNet.forward masks first (mvf_encoder.py:292-299); the two nets then CLAMP
    # rows on the clamp edges: x at the last cell's upper edge region, rho beyond the cylinder range, phi = +-180
    extra = np.array([[0, 76.79999, 0.0, 0.0, 0.5, 0.0], [1, -76.8, -76.8, -10.0, 0.1, 0.0], [0, -50.0, 0.0, 1.0, 0.2, 0.0], [0, -50.0, -0.0, 1.0, 0.2, 0.0],
                      [1, 76.0, 76.0, 9.99, 0.3, 0.0], [0, 0.0, 0.0, 0.0, 0.4, 0.0]], np.float32)
    pts = np.concatenate([pts, extra])
    pts = pts[rng.permutation(len(pts))]
    tp = torch.from_numpy(pts)
    out = dict(points=pts, pc_range=np.asarray(pr, np.float64), voxel_size=np.asarray(vs, np.float64), cylinder_range=np.asarray(cr, np.float64),
               cylinder_size=np.asarray(cs, np.float64))
    with torch.no_grad():
        for tag, net in (("pillar", PillarVoxelNet(vs, pr)), ("cyl", CylinderNet(cs, cr))):
            f, c, inv, g = net(tp)
            out[f"{tag}_features"], out[f"{tag}_coords"], out[f"{tag}_unq_inv"], out[f"{tag}_grid"] = (f.numpy(), c.numpy().astype(np.int32),
                                                                                                     inv.numpy().astype(np.int64), np.asarray(g, np.int64))
            print(f"[golden] mvf {tag}: N={len(pts)} cells={len(c)} grid={g}")
        pn = PointNet(20, 32).eval()
        torch.manual_seed(5)
        for prm in pn.parameters():
            prm.copy_(torch.randn(prm.shape) * 0.3)
        pn.norm.running_mean.copy_(torch.randn(32) * 0.2)
        pn.norm.running_var.copy_(torch.rand(32) + 0.5)
        x = torch.randn(300, 20)
        out["pn_in"], out["pn_out"] = x.numpy(), pn(x).numpy()
        for k, v in pn.state_dict().items():
            out["pn_" + k] = v.numpy()
        img = torch.randn(2, 6, 9, 11)
        co = torch.stack([torch.randint(0, 2, (200,)).float(), torch.rand(200) * 13 - 1, torch.rand(200) * 11 - 1], 1)
        out["bil_image"], out["bil_coords"] = img.numpy(), co.numpy()
        out["bil_out"] = SingleView.bilinear_interpolate(None, img, co).numpy()
    np.savez_compressed(os.path.join(OUT, "mvf_parts.npz"), **out)


# --------------------------------------------------------------------------- multi-sweep merge (nusc.py:76-121, waymo.py:49-67)
def install_devkit_import_stubs():
    """The dataset modules import the nuScenes / Waymo devkits (and pyquaternion, fire) at module level for evaluation and label code
    that is never called here; read_file / read_sweep / remove_close / load_pointcloud are pure numpy.  Import-only stand-ins, like numba."""
    def mod(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    never = type("Never", (), {"__init__": lambda self, *a, **k: (_ for _ in ()).throw(RuntimeError("devkit stand-in: import only"))})
    mod("nuscenes", NuScenes=never)
    mod("nuscenes.utils", splits=types.SimpleNamespace())
    mod("nuscenes.utils.data_classes", Box=never)
    mod("nuscenes.eval")
    mod("nuscenes.eval.detection")
    mod("nuscenes.eval.detection.config", config_factory=lambda *a, **k: None)
    mod("nuscenes.eval.detection.evaluate", NuScenesEval=never)
    mod("pyquaternion", Quaternion=never)
    mod("fire", Fire=lambda *a, **k: None)
    mod("waymo_open_dataset", label_pb2=types.SimpleNamespace())
    mod("waymo_open_dataset.protos", metrics_pb2=types.SimpleNamespace())


def gen_merge():
    """Fixtures for the sweep merge: the reference's NuScenesDataset.load_pointcloud / WaymoDataset.load_pointcloud run on raw sweep
    files written to a temporary directory.  Inputs (raw sweeps, transforms / poses, time lags) and the merged (N, 5) result."""
    import tempfile

    install_devkit_import_stubs()
    from det3d.datasets.nuscenes.nusc import NuScenesDataset
    from det3d.datasets.waymo.waymo import WaymoDataset

    rng = np.random.default_rng(61)
    out = {}
    with tempfile.TemporaryDirectory() as root:
        # ---- nuScenes: key frame + 3 past sweeps, (n, 5) fp32 files [x y z intensity ring], 4x4 fp64 transforms, time lags
        ds = object.__new__(NuScenesDataset)
        ds._root_path = root
        sweeps = []
        for k in range(4):
            n = 900 + 37 * k
            p = np.empty((n, 5), np.float32)
            p[:, :2] = rng.uniform(-45, 45, (n, 2))
            p[: n // 12, :2] = rng.uniform(-1.6, 1.6, (n // 12, 2))            # near the ego vehicle: removed from PAST sweeps (remove_close)
            p[0, :2] = [1.0, 0.3]                                              # |x| == radius exactly: kept (strict <)
            p[1, :2] = [0.999999, -0.999999]
            p[:, 2] = rng.uniform(-3, 1, n)
            p[:, 3] = rng.uniform(0, 255, n)
            p[:, 4] = rng.integers(0, 32, n)
            p.tofile(os.path.join(root, f"s{k}.bin"))
            out[f"nusc_raw{k}"] = p
            if k > 0:
                a = 0.013 * k
                T = np.eye(4)
                T[:3, :3] = np.array([[np.cos(a), -np.sin(a), 0.002 * k], [np.sin(a), np.cos(a), -0.001 * k], [-0.002 * k, 0.001 * k, 1.0]])
                T[:3, 3] = [0.43 * k, -0.11 * k, 0.02 * k]
                out[f"nusc_T{k}"] = T
                sweeps.append({"lidar_path": f"s{k}.bin", "transform_matrix": T, "time_lag": 0.05 * k})
        out["nusc_time_lag"] = np.asarray([0.0, 0.05, 0.10, 0.15], np.float64)
        res = ds.load_pointcloud({}, {"lidar_path": "s0.bin", "sweeps": sweeps})
        out["nusc_points"] = res["points"]
        print(f"[golden] merge nusc: {sum(len(out[f'nusc_raw{k}']) for k in range(4))} raw rows -> {len(res['points'])} merged, dtype {res['points'].dtype}")
        # ---- Waymo: key frame + 2 past sweeps, (n, 6) fp32 files [x y z intensity elongation nlz], poses, timestamps; nlz != -1 rows are dropped
        os.makedirs(os.path.join(root, "lidar_point"))
        wd = object.__new__(WaymoDataset)
        wd._root_path, wd.nsweeps, wd.drop_frames = root, 3, 0
        poses, ts = [], [0.0, -0.1, -0.2]
        for k in range(3):
            n = 700 + 29 * k
            p = np.empty((n, 6), np.float32)
            p[:, :2] = rng.uniform(-70, 70, (n, 2))
            p[:, 2] = rng.uniform(-2, 4, n)
            p[:, 3] = rng.uniform(0, 1, n)
            p[:, 4] = rng.uniform(0, 1, n)
            p[:, 5] = np.where(rng.uniform(0, 1, n) < 0.9, -1.0, 1.0)
            p.tofile(os.path.join(root, "lidar_point", f"w{k}.bin"))
            out[f"waymo_raw{k}"] = p
            a = 0.02 * k
            P = np.eye(4)
            P[:3, :3] = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1.0]])
            P[:3, 3] = [1.3 * k, 0.2 * k, 0.01 * k]
            poses.append(P)
            out[f"waymo_pose{k}"] = P
        out["waymo_timestamp"] = np.asarray(ts, np.float64)
        info = {"token": "w0", "pose": poses[0], "sweeps": [{"token": f"w{k}", "timestamp": ts[k], "pose": poses[k]} for k in (1, 2)]}
        res = wd.load_pointcloud({}, info)
        out["waymo_points"] = res["points"]
        print(f"[golden] merge waymo: -> {len(res['points'])} merged rows")
    np.savez_compressed(os.path.join(OUT, "merge_sweeps.npz"), **out)


def main():
    assert os.path.isdir(REF), "gen_golden.py needs the reference tree"
    os.makedirs(OUT, exist_ok=True)
    O.build()
    assert O.have_ref(), "oracle/_ref/libref_iou3d.so missing: run `make -C oracle ref`"
    install_torch_scatter()
    install_numba_identity()
    install_iou3d_nms()
    sys.path.insert(0, REF)
    what = sys.argv[1:] or ["reader", "train", "iou", "decode", "decode_lazy", "loss", "voxel", "mvf", "merge"]
    if "reader" in what:
        gen_reader()
    if "train" in what:
        gen_reader_train()
    if "iou" in what:
        gen_iou_nms()
    if "decode" in what:
        gen_decode()
    if "decode_lazy" in what:
        gen_decode("decode_2task_bf16reg", bf16_reg=True)
    if "loss" in what:
        gen_loss()
    if "voxel" in what:
        gen_voxel()
    if "mvf" in what:
        gen_mvf_parts()
    if "merge" in what:
        gen_merge()


if __name__ == "__main__":
    main()

"""GPU: the BASELINE.json configurations round 1 left untested on hardware (VERDICT r1 "configs not exercised"):
  C5   Waymo 3-frame, 540 k points, 0.1 m pillars, 1504 x 1504, FP16 canvas -- full size vs the CPU oracle
  C3/C4 train-mode reader at C2 / C4 geometry, full size (batch statistics) vs a torch fp32 statement of the same op sequence
  C3   one training step of PillarNeXt-B at C2 geometry, 4 frames on one GPU: finite loss and gradients, peak memory recorded
  plus the plain (un-fused, fp32) eval detector without manual casts and a 2-rank RCCL DDP step when 2 GPUs exist."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _net(config, layers, train=False):
    from pillarnext_amd import synth
    from test_gpu_reader import make_net

    cfg = synth.CONFIGS[config]
    net = make_net(cfg["pc_range"], cfg["voxel_size"], layers)
    return (net.train() if train else net), cfg


def test_c5_waymo_3frame_fp16_canvas_full_size(oracle):
    from pillarnext_amd import synth

    layers = synth.pfn_params(5, (64, 64), 0)
    net, cfg = _net("C5", layers)
    pts = synth.make_batch("C5", 1, "sweep")
    assert pts.shape[0] == 540_000
    o = oracle.reader_forward(pts, cfg["pc_range"], cfg["voxel_size"], [64, 64], layers, B=1)
    tp = torch.from_numpy(pts).cuda()
    ny, nx = (int(v) for v in net.grid_size)
    assert (ny, nx) == (1504, 1504)
    occ = torch.empty((1, ny, nx), dtype=torch.uint8, device="cuda")
    canvas = net.forward_dense(tp, 1, dtype=torch.float16, occupancy=occ)
    assert canvas.dtype == torch.float16 and canvas.shape == (1, 64, ny, nx)
    c = torch.from_numpy(o["coords"]).long().cuda()
    assert int(occ.sum()) == o["P"] and bool((occ[c[:, 0], c[:, 1], c[:, 2]] == 1).all())        # indices: bit-exact
    got = canvas.permute(0, 2, 3, 1)[c[:, 0], c[:, 1], c[:, 2]].float()
    ref = torch.from_numpy(o["feat_max"]).cuda()
    # fp16 store of an fp32 value that is within 1e-4 + 1e-4|ref| of the reference: <= that + half an fp16 ulp
    assert bool(((got - ref).abs() <= 1e-4 + 1e-4 * ref.abs() + 2.0 ** -11 * ref.abs().clamp(min=2.0 ** -14)).all())
    assert int((canvas != 0).any(dim=1).sum()) <= o["P"]
    fm, coords, _ = net(tp, 1)
    assert np.array_equal(coords.cpu().numpy(), o["coords"])
    assert torch.equal(got, fm.to(torch.float16).float())                                          # exactly one RNE rounding of our fp32
    # size-independent properties at full size: permutation invariance and idempotence, bit for bit
    perm = torch.randperm(tp.shape[0], device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))
    assert torch.equal(canvas, net.forward_dense(tp[perm].contiguous(), 1, dtype=torch.float16))
    assert torch.equal(canvas, net.forward_dense(tp, 1, dtype=torch.float16))


def _torch_train_reference(feats, inv, P, layers_mods):
    """fp32 torch statement of PFNLayer x2 in train mode (pe:35-50): Linear, BatchNorm1d with batch statistics, ReLU, per-pillar max."""
    x = feats
    for i, pfn in enumerate(layers_mods):
        y = torch.nn.functional.linear(x, pfn.linear.weight)
        mu, var = y.mean(0), y.var(0, unbiased=False)
        y = torch.relu((y - mu) / torch.sqrt(var + pfn.norm.eps) * pfn.norm.weight + pfn.norm.bias)
        m = torch.zeros((P, y.shape[1]), device=y.device).scatter_reduce(0, inv[:, None].expand_as(y), y, "amax", include_self=True)
        x = m if i == len(layers_mods) - 1 else torch.cat([y, m[inv]], dim=1)
    return x


@pytest.mark.parametrize("config,batch", [("C2", 4), ("C4", 4)])
def test_train_mode_reader_full_size(config, batch):
    """C3 / C4 training geometry, 4 frames per GPU, train mode: forward == torch fp32 statement on the HIP voxelizer's own
    (oracle-verified) features; parameter gradients flow and are finite; the dense canvas carries the same rows."""
    from pillarnext_amd import synth

    layers = synth.pfn_params(5, (64, 64), 0)
    net, cfg = _net(config, layers, train=True)
    tp = torch.from_numpy(synth.make_batch(config, batch, "sweep")).cuda()
    feats, coords, inv, _ = net.voxelization(tp, batch)
    P = coords.shape[0]
    with torch.no_grad():
        ref = _torch_train_reference(feats, inv, P, net.pfn_layers)
    rm0 = net.pfn_layers[1].norm.running_mean.clone()
    fm, coords2, _ = net(tp, batch)
    assert torch.equal(coords, coords2)
    torch.testing.assert_close(fm.detach(), ref, rtol=1e-4, atol=1e-4)
    assert not torch.equal(net.pfn_layers[1].norm.running_mean, rm0)                               # running statistics moved
    fm.square().mean().backward()
    for pfn in net.pfn_layers:
        for p in (pfn.linear.weight, pfn.norm.weight, pfn.norm.bias):
            assert p.grad is not None and bool(torch.isfinite(p.grad).all()) and float(p.grad.abs().sum()) > 0
    canvas = net.forward_dense(tp, batch, dtype=torch.float32)
    c = coords.long()
    torch.testing.assert_close(canvas.permute(0, 2, 3, 1)[c[:, 0], c[:, 1], c[:, 2]].detach(), ref, rtol=1e-4, atol=1e-4)


def test_plain_eval_detector_needs_no_manual_casts():
    """build_pillarnext_b(...).cuda().eval()(example) with fp32 modules -- as instantiated from the YAML or loaded from a reference
    checkpoint -- must run as is (the canvas used to be bf16 against fp32 weights)."""
    from pillarnext_amd import synth
    from pillarnext_amd.models import build_pillarnext_b

    cfg = synth.CONFIGS["C1"]
    torch.manual_seed(0)
    det = build_pillarnext_b(cfg["pc_range"], cfg["voxel_size"], tasks=[["car"], ["pedestrian", "cyclist"]]).cuda().eval()
    pts = torch.from_numpy(synth.make_batch("C1", 2, "sweep", n=20_000)).cuda()
    out = det({"points": pts, "token": ["a", "b"], "batch_size": 2})
    assert set(out) == {"a", "b"} and out["a"]["box3d_lidar"].shape[1] == 9 and out["a"]["scores"].dtype == torch.float32
    det.backbone.to(torch.bfloat16), det.neck.to(torch.bfloat16), det.head.to(torch.bfloat16)      # and follows a later cast
    out2 = det({"points": pts, "token": ["a", "b"], "batch_size": 2})
    assert set(out2) == {"a", "b"}


def test_training_step_c2_geometry_4_frames():
    """tools/train_step.py (C3: PillarNeXt-B training at C2 geometry, 4 frames per GPU) on ONE GPU: two optimizer steps, finite loss,
    finite non-zero gradients; the peak memory goes to gpurun_out/ for the report."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.setdefault("MIOPEN_FIND_MODE", "2")  # FAST: the exhaustive search over fp32 backward convolutions at 1440^2 took 260 s of a 265 s test
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "train_step.py"), "--batch", "4", "--steps", "2", "--config", "C2", "--check"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("step ")]
    assert len(lines) == 2 and "peak" in p.stdout
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "train_step_c2_b4.log"), "w") as f:
        f.write(p.stdout)


def test_waymo_training_step_full_size():
    """BASELINE configs[3] as a TRAINING step: the Waymo detector of configs/pillarnext_b_waymo.yaml (2 tasks, iou head: IouLoss with the
    aligned rotated 3-D IoU target on the fused loss kernel, waymo_det_pp18_aspp_iou_car_sp_f1.yaml) at C4 geometry (180 k points, 0.1 m,
    1504 x 1504), 2 frames, bf16 autocast: two optimizer steps, finite loss, finite non-zero gradients."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.setdefault("MIOPEN_FIND_MODE", "2")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "train_step.py"), "--batch", "2", "--steps", "2", "--config", "C4", "--check", "--amp",
                        "--yaml", os.path.join(ROOT, "configs", "pillarnext_b_waymo.yaml")], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("step ")]
    assert len(lines) == 2 and "peak" in p.stdout
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "train_step_c4_b2_waymo.log"), "w") as f:
        f.write(p.stdout)


def test_two_rank_rccl_ddp_matches_single_process():
    """2 ranks over RCCL (backend "nccl"): SyncBN conversion + DDP through the real reader; averaged gradients == one process on
    the 2 x 2-frame batch.  Skipped on the 1-GPU box; the driver's 8-GPU node runs it."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "tools", "ddp_parity.py")], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "DDP PARITY OK" in p.stdout, p.stdout[-1500:] + p.stderr[-3000:]


def _waymo_fused(config, dtype):
    """Waymo detector from configs/pillarnext_b_waymo.yaml with BASELINE's synthetic geometry (C4 / C5: 0.1 m, +-75.2 m, 1504 x 1504)."""
    from pillarnext_amd import config as C
    from pillarnext_amd import synth
    from pillarnext_amd.models import FusedPillarNeXt

    cfg = C.load(os.path.join(ROOT, "configs", "pillarnext_b_waymo.yaml"))
    g = synth.CONFIGS[config]
    cfg["model"]["reader"]["voxel_size"] = list(g["voxel_size"])
    cfg["model"]["reader"]["pc_range"] = list(g["pc_range"])
    for blk in ("head", "post_processing"):
        cfg["model"][blk]["voxel_size"] = list(g["voxel_size"])
        cfg["model"][blk]["pc_range"] = list(g["pc_range"])
    torch.manual_seed(0)
    det = C.instantiate(cfg["model"]).cuda().eval()
    assert list(det.reader.grid_size) == [1504, 1504] and det.head.with_iou
    return det, FusedPillarNeXt(det, dtype=dtype).cuda().eval()


@pytest.mark.parametrize("config,dtype", [("C4", "bfloat16"), ("C5", "float16")])
def test_waymo_fused_detector_full_size(config, dtype):
    """BASELINE configs[3] / [4] end to end through the fused inference graph: C4 = 180 k points, bf16 (HIP conv kernels); C5 = 540 k
    points (3 sweeps), fp16 canvas and network.  2 tasks with the iou head (IoU-rectified scores), NMS pre 4096 / post 500."""
    from pillarnext_amd import synth

    det, fused = _waymo_fused(config, getattr(torch, dtype))
    B = 2
    pts = torch.from_numpy(synth.make_batch(config, B, "sweep")).cuda()
    ex = {"points": pts, "token": ["a", "b"], "batch_size": B}
    with torch.no_grad():
        d1 = fused(ex)
        d2 = fused(ex)
    assert set(d1) == {"a", "b"}
    for tok in ("a", "b"):
        r = d1[tok]
        n = len(r["scores"])
        assert r["box3d_lidar"].shape == (n, 9) and r["label_preds"].shape == (n,) and n > 0
        assert bool(torch.isfinite(r["box3d_lidar"]).all()) and bool(torch.isfinite(r["scores"]).all())
        assert int(r["label_preds"].min()) >= 0 and int(r["label_preds"].max()) <= 2
        assert float(r["scores"].min()) > 0.0 and float(r["scores"].max()) <= 1.0
        for c in range(3):
            sc = r["scores"][r["label_preds"] == c]
            assert len(sc) <= 500 and bool((sc[:-1] >= sc[1:]).all())          # post_max per class, score order inside a class
        lim = 80.0
        assert bool((r["box3d_lidar"][:, :2].abs() <= lim).all())
        assert torch.equal(r["box3d_lidar"], d2[tok]["box3d_lidar"]) and torch.equal(r["scores"], d2[tok]["scores"])   # deterministic
    # the packed HIP decoder against the module implementation of CenterHead.predict (centerhead.py:231-384) on the SAME head maps:
    # same number of detections per frame and class, same boxes in the same order
    with torch.no_grad():
        preds = fused.forward_preds(pts, B)
        ref = det.head.predict({"token": ["a", "b"]}, [{k: v.float() for k, v in p.items()} for p in preds], det.post_processing)
    for tok, rr in zip(("a", "b"), ref):
        got = d1[tok]
        assert len(rr["scores"]) == len(got["scores"])
        torch.testing.assert_close(got["scores"].cpu(), rr["scores"].cpu().float(), rtol=1e-4, atol=1e-5)
        # the lazy head evaluates the regression branches at the candidates with its fp32 sums in another order than the dense kernels: an
        # intermediate may round to the neighbouring bf16 / fp16 value (tests/test_gpu_lazy_head.py) -- all but a handful of elements agree to 1e-4
        gb, rb = got["box3d_lidar"].cpu(), rr["box3d_lidar"].cpu().float()
        # Row by row: the same box at the same rank -- except inside runs of candidates whose scores agree to the last bits.  On this random-init head whole
        # runs share ONE fp16 class score, the IoU-rectified score (hm^(1-a) * iou^a, fp32) then differs between the HIP decoder and the module by a few ulp,
        # and so does the order inside the run (seen on C5 / fp16 after round 6 changed the convolutions' summation order: 14-27 of 1000 rows, the SET of boxes
        # identical).  So: the same multiset of boxes, every row matched by a row of (nearly) the same score, and all but 5 % of the rows in place.
        rows_ok = ((gb - rb).abs() <= 2e-2 + 2e-2 * rb.abs()).all(1)
        assert float(rows_ok.float().mean()) >= 0.95, f"{int((~rows_ok).sum())} of {len(rows_ok)} boxes out of place"
        gs, rs = got["scores"].cpu(), rr["scores"].cpu().float()
        matched = set()
        for r in (~rows_ok).nonzero().flatten().tolist():          # a displaced row IS one of the reference's rows, among rows of its own score
            d = (rb[:, :7] - gb[r, :7]).abs().sum(1)
            j = int(d.argmin())
            assert float(d[j]) < 0.05, f"row {r}: no such box in the module's output"
            assert abs(float(rs[j]) - float(gs[r])) <= 1e-4 * float(gs[r]) + 1e-6 and j not in matched
            matched.add(j)
        assert float(((gb - rb).abs() <= 1e-4 + 1e-4 * rb.abs()).float().mean()) > 0.95
        assert torch.equal(got["label_preds"].cpu(), rr["label_preds"].cpu())

"""GPU: the lazy SepHead (regression branches evaluated only at the decoder's candidate cells, csrc/conv3x3.hip::k_sephead_lazy and
decode.PackedDecoder.launch_lazy) against the dense head: the kernel against torch convolutions over the whole map, the torch evaluator
against the kernel, and the detections of the lazy pipeline against the dense pipeline (PNX_HEAD_LAZY=0) on the same weights."""
import os

import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

OFF, K = [0, 2, 3, 6, 8], [2, 1, 3, 2, 2]


def _weights(seed):
    g = torch.Generator().manual_seed(seed)
    W1 = (torch.randn((320, 64, 3, 3), generator=g) * 0.06).to(torch.bfloat16).float()
    b1 = torch.randn((320,), generator=g) * 0.1
    W2 = torch.zeros((10, 320, 3, 3))
    for j in range(5):
        W2[OFF[j]:OFF[j] + K[j], 64 * j:64 * (j + 1)] = torch.randn((K[j], 64, 3, 3), generator=g) * 0.06
    W2 = W2.to(torch.bfloat16).float()
    b2 = torch.randn((10,), generator=g) * 0.1
    return W1.cuda(), b1.cuda(), W2.cuda(), b2.cuda()


def _w2m(W2):
    m = torch.zeros((9 * 320, 10), device=W2.device)
    for pos in range(9):
        m[pos * 320:(pos + 1) * 320] = W2[:, :, pos // 3, pos % 3].t()
    return m


@pytest.mark.parametrize("shape,pre_max,lens", [((2, 40, 36), 500, [500, 123]), ((1, 7, 5), 35, [35]), ((3, 64, 64), 40, [40, 17, 0]),
                                                ((2, 33, 47), 64, [64, 1])])
def test_lazy_kernel_equals_dense_convolutions_at_the_cells(shape, pre_max, lens):
    from pillarnext_amd import ops

    B, H, W = shape
    g = torch.Generator(device="cuda").manual_seed(5)
    tasks, dense = [], []
    for ti in range(2):                                                                               # two tasks of one class each
        W1, b1, W2, b2 = _weights(3 + ti)
        up = torch.randn((B, 64, H, W), device="cuda", generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        t = torch.relu(torch.nn.functional.conv2d(up.float(), W1, b1, padding=1)).to(torch.bfloat16).float()
        dense.append(torch.nn.functional.conv2d(t, W2, b2, padding=1).to(torch.bfloat16).float().permute(0, 2, 3, 1).reshape(-1, 10))
        tasks.append((up, ops.conv3x3_pack_weights(W1), b1, ops.sephead_lazy_pack_w2(_w2m(W2)), b2))
    S = 2 * B
    local = torch.randint(0, B * H * W, (S, pre_max), device="cuda", generator=g)
    local[:, :4] = torch.tensor([0, W - 1, (H - 1) * W, B * H * W - 1], device="cuda")                # the four kinds of corner
    seg_len = torch.tensor([lens[s // 2] for s in range(S)], dtype=torch.int32, device="cuda")       # lists s = sample * 2 + class
    got = ops.sephead_lazy(tasks, [0, 1], B, local, seg_len, pre_max)
    valid = torch.arange(pre_max, device="cuda")[None, :] < seg_len[:, None]
    ref = torch.stack([dense[s % 2][local[s]] for s in range(S)]) * valid[..., None]
    assert bool((got[~valid] == 0).all())
    # fp32 sums in a different order than the library convolution: an intermediate may round to the neighbouring bf16 value
    torch.testing.assert_close(got, ref, rtol=2e-2, atol=2e-2)
    assert float((got == ref).float().mean()) > 0.8
    assert torch.equal(got, ops.sephead_lazy(tasks, [0, 1], B, local, seg_len, pre_max))              # deterministic


def _fused(lazy):
    from test_gpu_configs import _waymo_fused

    os.environ["PNX_HEAD_LAZY"] = "1" if lazy else "0"
    try:
        return _waymo_fused("C4", torch.bfloat16)
    finally:
        os.environ.pop("PNX_HEAD_LAZY", None)


def test_lazy_pipeline_equals_dense_pipeline(monkeypatch):
    """Same weights (same seed), same frames: the detections of the lazy head equal the dense head's -- same counts per frame, same labels,
    scores bit-equal (they come from the same dense hm / iou kernels), boxes within the bf16 rounding of one intermediate."""
    from pillarnext_amd import synth

    torch.manual_seed(0)
    det, lazy = _fused(True)
    from pillarnext_amd.models import FusedPillarNeXt

    os.environ["PNX_HEAD_LAZY"] = "0"
    try:
        dense = FusedPillarNeXt(det, dtype=torch.bfloat16).cuda().eval()
    finally:
        os.environ.pop("PNX_HEAD_LAZY", None)
    assert lazy.lazy_head and not dense.lazy_head
    B = 2
    pts = torch.from_numpy(synth.make_batch("C4", B, "sweep")).cuda()
    ex = {"points": pts, "token": ["a", "b"], "batch_size": B}
    with torch.no_grad():
        dl, dd = lazy(ex), dense(ex)
        # the torch statement of the evaluator (models.FusedPillarNeXt.lazy_eval) against the kernel, through the whole pipeline
        monkeypatch.setenv("PNX_HEAD_LAZY_TORCH", "1")
        dt = lazy(ex)
        monkeypatch.delenv("PNX_HEAD_LAZY_TORCH")
    for tok in ("a", "b"):
        a, b, c = dl[tok], dd[tok], dt[tok]
        assert len(a["scores"]) == len(b["scores"]) > 0
        assert torch.equal(a["label_preds"], b["label_preds"]) and torch.equal(a["scores"], b["scores"])
        torch.testing.assert_close(a["box3d_lidar"], b["box3d_lidar"], rtol=2e-2, atol=2e-2)
        assert float((a["box3d_lidar"] == b["box3d_lidar"]).float().mean()) > 0.9
        assert len(c["scores"]) == len(a["scores"]) and torch.equal(c["scores"], a["scores"])
        torch.testing.assert_close(c["box3d_lidar"], a["box3d_lidar"], rtol=2e-2, atol=2e-2)


def test_fallback_to_the_dense_path_when_the_range_test_cuts_a_full_list(monkeypatch):
    """A post_center_limit_range tighter than the map makes selected candidates fail the range test while their lists are cut at pre_max:
    the lazy launch raises its flag and result() returns exactly what the dense path returns."""
    from pillarnext_amd import synth

    torch.manual_seed(0)
    det, lazy = _fused(True)
    if isinstance(det.post_processing, dict):
        det.post_processing["post_center_limit_range"] = [-20.0, -20.0, -10.0, 20.0, 20.0, 10.0]
    else:
        det.post_processing.post_center_limit_range = [-20.0, -20.0, -10.0, 20.0, 20.0, 10.0]
    from pillarnext_amd.models import FusedPillarNeXt

    lazy = FusedPillarNeXt(det, dtype=torch.bfloat16).cuda().eval()
    os.environ["PNX_HEAD_LAZY"] = "0"
    try:
        dense = FusedPillarNeXt(det, dtype=torch.bfloat16).cuda().eval()
    finally:
        os.environ.pop("PNX_HEAD_LAZY", None)
    pts = torch.from_numpy(synth.make_batch("C4", 1, "sweep")).cuda()
    ex = {"points": pts, "token": ["a"], "batch_size": 1}
    with torch.no_grad():
        pend = lazy.forward_async(ex)
        res = pend.result()
        assert int(pend.flag_h[0]) != 0, "random heads fill every list: candidates outside the range must raise the flag"
        dd = dense(ex)["a"]
    got = lazy.detections(res)["a"]
    assert torch.equal(got["scores"], dd["scores"]) and torch.equal(got["box3d_lidar"], dd["box3d_lidar"])


def test_lazy_launch_matches_the_reference_golden():
    """The bench's DEFAULT decode path -- PackedDecoder.launch_lazy: keys and exact top-k from the dense [iou] hm maps, k_sephead_lazy at the
    candidates, k_decode_boxes_lazy, batched NMS -- against the reference's CenterHead.predict (tests/golden/decode_2task_bf16reg.npz,
    oracle/gen_golden.py decode_lazy: the regression maps are bf16-representable, so identity convolutions reproduce them EXACTLY through
    the evaluator: `up` carries the positive and negative parts of the ten regression channels, conv 1 copies them (centre tap), conv 2
    subtracts).  Same labels and counts, scores to 1e-5, boxes to 1e-4: the tolerances of the dense decoder's golden test."""
    import numpy as np

    from conftest import load_golden
    from pillarnext_amd import ops
    from pillarnext_amd.decode import PackedDecoder

    g = load_golden("decode_2task_bf16reg")
    test_cfg = dict(post_center_limit_range=list(g["post_center_limit_range"]), score_threshold=float(g["score_threshold"]),
                    nms=dict(nms_pre_max_size=int(g["pre_max"]), nms_post_max_size=int(g["post_max"]), nms_iou_threshold=[[0.2], [0.2, 0.25]]),
                    out_size_factor=[int(v) for v in g["out_size_factor"]], voxel_size=list(g["voxel_size"]), pc_range=list(g["pc_range"]))
    W1 = torch.zeros((320, 64, 3, 3))
    W2 = torch.zeros((10, 320, 3, 3))
    for j in range(5):
        for i in range(K[j]):
            W1[64 * j + i, OFF[j] + i, 1, 1] = 1.0                  # positive part of regression channel OFF[j] + i
            W1[64 * j + K[j] + i, 10 + OFF[j] + i, 1, 1] = 1.0      # negative part
            W2[OFF[j] + i, 64 * j + i, 1, 1] = 1.0
            W2[OFF[j] + i, 64 * j + K[j] + i, 1, 1] = -1.0
    W1, W2 = W1.cuda(), W2.cuda()
    zeros320, zeros10 = torch.zeros(320, device="cuda"), torch.zeros(10, device="cuda")
    dense, tasks = [], []
    for t, ncls in enumerate((1, 2)):
        reg = np.concatenate([g[f"t{t}_{k}"] for k in ("reg", "height", "dim", "rot", "vel")], axis=1)       # (B, 10, H, W), bf16-representable
        B, _, H, Wd = reg.shape
        up = np.zeros((B, 64, H, Wd), np.float32)
        up[:, :10], up[:, 10:20] = np.maximum(reg, 0), np.maximum(-reg, 0)
        upt = torch.from_numpy(up).cuda().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        assert torch.equal(upt.float().cpu(), torch.from_numpy(up))                                             # nothing was rounded
        tasks.append((upt, ops.conv3x3_pack_weights(W1), zeros320, ops.sephead_lazy_pack_w2(_w2m(W2)), zeros10))
        hm = np.concatenate([g[f"t{t}_iou"], g[f"t{t}_hm"]], axis=1)                                          # [iou] hm, fp32
        hm = np.concatenate([hm, np.zeros((B, 16 - hm.shape[1], H, Wd), np.float32)], axis=1)
        dense.append(torch.from_numpy(hm).cuda().contiguous(memory_format=torch.channels_last))
    dec = PackedDecoder([1, 2], [[0.5], [0.68, 0.2]], test_cfg, True, [16, 16])

    def evaluator(local, seg_len, valid, segs):
        return ops.sephead_lazy(tasks, [0, 1, 1], 2, local, seg_len, local.shape[1])

    pend = dec.launch_lazy(dense, evaluator, ["a", "b"], None)
    res = pend.result()
    assert int(pend.flag_h[0]) == 0
    for i, r in enumerate(res):
        assert np.array_equal(r["label_preds"].numpy(), g[f"s{i}_labels"])
        np.testing.assert_allclose(r["scores"].numpy(), g[f"s{i}_scores"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(r["box3d_lidar"].numpy(), g[f"s{i}_boxes"], rtol=1e-4, atol=1e-4)

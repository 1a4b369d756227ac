"""GPU: the fused CenterHead losses (csrc/center_loss.hip behind pnx_center_loss_forward / _backward) against the module statement in
pillarnext_amd/losses.py (itself pinned to the reference's loss values and gradients by tests/golden/head_loss_2task.npz):
det3d/models/loss/centerloss.py:8-110,139-176 as combined by det3d/models/heads/centerhead.py:142-229."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _case(seed, B=3, H=40, W=48, M=500, ncls=2, n_pos=37, with_iou=True, nan_targets=True, collide=True):
    g = torch.Generator(device="cuda").manual_seed(seed)
    dev = "cuda"
    r = lambda *s: torch.randn(*s, device=dev, generator=g)  # noqa: E731
    pd = {"hm": r(B, ncls, H, W) - 2.0, "reg": torch.rand((B, 2, H, W), device=dev, generator=g), "height": r(B, 1, H, W) * 0.5,
          "dim": r(B, 3, H, W) * 0.4 + 0.5, "rot": r(B, 2, H, W), "vel": r(B, 2, H, W)}
    pd["dim"][0, 0, 0, :4] = 7.0          # outside the clamp of the size: no gradient through exp
    if with_iou:
        pd["iou"] = r(B, 1, H, W) * 0.5
    hm_t = torch.rand((B, ncls, H, W), device=dev, generator=g) ** 4
    ind = torch.randint(0, H * W, (B, M), device=dev, generator=g)
    if collide:
        ind[:, 1] = ind[:, 0]             # two objects in one cell: the gradients add up
        ind[0, 2] = 0                     # the cell with the clamped size
    mask = torch.zeros((B, M), dtype=torch.uint8, device=dev)
    mask[:, :n_pos] = 1
    mask[1, :] = 0                        # a sample without objects
    cat = torch.randint(0, ncls, (B, M), device=dev, generator=g)
    anno = r(B, M, 10) * 0.5
    if nan_targets:
        anno[:, ::5, 6:8] = float("nan")  # "no velocity label" (:55-56)
    # ground-truth boxes near the predicted cell so that the IoUs are not all zero
    ys, xs = (ind // W).float(), (ind % W).float()
    gtb = torch.stack([(xs + 0.5) * 0.8 - 20.0 + r(B, M) * 0.2, (ys + 0.5) * 0.8 - 16.0 + r(B, M) * 0.2, r(B, M) * 0.3,
                       1.5 + torch.rand((B, M), device=dev, generator=g), 1.2 + torch.rand((B, M), device=dev, generator=g),
                       1.0 + torch.rand((B, M), device=dev, generator=g), r(B, M)], dim=2)
    ex = {k: [v] for k, v in dict(hm=hm_t, ind=ind, mask=mask, cat=cat, anno_box=anno, gt_boxes=gtb).items()}
    return pd, ex


@pytest.mark.parametrize("with_iou,n_pos", [(True, 37), (False, 60), (True, 0)])
def test_fused_losses_match_the_module_losses(with_iou, n_pos, monkeypatch):
    from pillarnext_amd.models import CenterHead

    common = {"reg": (2, 2), "height": (1, 2), "dim": (3, 2), "rot": (2, 2), "vel": (2, 2)}
    if with_iou:
        common["iou"] = (1, 2)
    head = CenterHead(16, [["a", "b"]], 0.25, [1.0] * 6 + [0.2, 0.2, 1.0, 1.0], common, [2], share_conv_channel=16, with_reg_iou=True,
                      voxel_size=[0.2, 0.2, 8.0], pc_range=[-20.0, -16.0, -5.0, 18.4, 16.0, 3.0], out_size_factor=[4]).cuda()
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("PNX_FUSED_LOSS", mode)
        pd, ex = _case(5, with_iou=with_iou, n_pos=n_pos)
        pd = {k: v.clone().requires_grad_(True) for k, v in pd.items()}
        total, rets = head.loss(ex, [pd])
        total.backward()
        res[mode] = (float(total), {k: float(rets[0][k]) for k in rets[0] if k.endswith("loss")}, rets[0]["loc_loss_elem"].detach().cpu().float(),
                     {k: v.grad.clone() for k, v in pd.items()})
    t0, l0, e0, g0 = res["0"]
    t1, l1, e1, g1 = res["1"]
    assert abs(t0 - t1) <= 2e-5 * abs(t0) + 1e-6, (t0, t1)
    for k in l0:
        assert abs(l0[k] - l1[k]) <= 2e-5 * abs(l0[k]) + 1e-6, (k, l0[k], l1[k])
    torch.testing.assert_close(e1, e0, rtol=2e-5, atol=1e-7)
    for k in g0:
        # the positive cells carry sums of a few atomically added terms; the IoU-loss sign flips only where |pred - target| ~ 0
        torch.testing.assert_close(g1[k], g0[k], rtol=2e-4, atol=2e-7, msg=lambda s, k=k: f"grad {k}: {s}")
    assert float(g1["hm"].abs().sum()) > 0

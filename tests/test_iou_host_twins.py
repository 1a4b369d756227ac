"""The host twins of the IoU entry points (include/pnx.h: pnx_boxes_iou_bev_cpu, pnx_boxes_aligned_iou_bev_cpu; the reference's
det3d/core/iou3d_nms/src/iou3d_cpu.cpp:232-273 through pillarnext_amd/iou3d_nms_cuda.py) need no GPU: CPU tensors in and out.  They are the
device kernels' source (csrc/iou3d_geom.h) compiled for the host, so they must equal the det-math oracle bit for bit, the reference-generated
golden matrix to 1e-5 -- and, on the GPU box, the HIP kernel bit for bit."""
import numpy as np
import pytest

from conftest import load_golden

torch = pytest.importorskip("torch")


def _random_boxes(n, seed):
    rng = np.random.default_rng(seed)
    b = np.zeros((n, 7), np.float32)
    b[:, :2] = rng.uniform(-10, 10, (n, 2))
    b[:, 2] = rng.uniform(-1, 1, n)
    b[:, 3:6] = rng.uniform(0.5, 5.0, (n, 3))
    b[:, 6] = rng.uniform(-3.2, 3.2, n)
    b[: n // 8] = b[n // 8: 2 * (n // 8)]            # identical pairs
    return b


def test_cpu_entry_points_match_the_oracle_and_the_golden_matrix(oracle):
    from pillarnext_amd import iou3d_nms_cuda as ext

    g = load_golden("iou_bev_64")
    a, b = torch.from_numpy(g["boxes_a"]), torch.from_numpy(g["boxes_b"])
    out = torch.zeros((a.shape[0], b.shape[0]))
    assert ext.boxes_iou_bev_cpu(a, b, out) == 1
    assert np.array_equal(out.numpy().view(np.uint32), oracle.boxes_iou_bev(g["boxes_a"], g["boxes_b"], "det").view(np.uint32))
    np.testing.assert_allclose(out.numpy(), g["iou_ab"], rtol=0, atol=1e-5)                       # the reference's own values (libm)
    al = torch.zeros((40, 1))
    assert ext.boxes_aligned_iou_bev_cpu(a[:40], b[:40], al) == 1
    assert np.array_equal(al.numpy().ravel().view(np.uint32), oracle.boxes_aligned_iou_bev(g["boxes_a"][:40], g["boxes_b"][:40], "det").ravel().view(np.uint32))
    # a larger, threaded call; empty inputs; a CUDA-typed / wrong-dtype argument is refused
    x, y = _random_boxes(700, 1), _random_boxes(300, 2)
    big = torch.zeros((700, 300))
    ext.boxes_iou_bev_cpu(torch.from_numpy(x), torch.from_numpy(y), big)
    assert np.array_equal(big.numpy().view(np.uint32), oracle.boxes_iou_bev(x, y, "det").view(np.uint32))
    ext.boxes_iou_bev_cpu(torch.zeros((0, 7)), torch.from_numpy(y), torch.zeros((0, 300)))
    with pytest.raises(Exception):
        ext.boxes_iou_bev_cpu(torch.from_numpy(x).double(), torch.from_numpy(y), big)


@pytest.mark.gpu
def test_host_twin_equals_the_hip_kernel_bit_for_bit():
    from pillarnext_amd import iou3d_nms_cuda as ext

    x, y = _random_boxes(500, 3), _random_boxes(400, 4)
    host = torch.zeros((500, 400))
    ext.boxes_iou_bev_cpu(torch.from_numpy(x), torch.from_numpy(y), host)
    dev = torch.zeros((500, 400), device="cuda")
    ext.boxes_iou_bev_gpu(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda(), dev)
    assert torch.equal(host.view(torch.int32), dev.cpu().view(torch.int32))

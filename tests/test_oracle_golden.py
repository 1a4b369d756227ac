"""CPU: the oracle (oracle/pnx_oracle.c) against the golden vectors produced by the reference
(tests/golden/*.npz, oracle/gen_golden.py) and against the compiled reference (oracle/_ref)."""
import numpy as np
import pytest

from conftest import READER_CASES, golden_layers, load_golden


@pytest.mark.parametrize("case", READER_CASES)
def test_voxelize_indices_bit_exact(oracle, case):
    g = load_golden(case)
    v = oracle.voxelize(g["points"], g["pc_range"], g["voxel_size"])
    assert v["P"] == len(g["coords"])
    assert np.array_equal(v["coords"], g["coords"])            # [b, y, x] int32, torch.unique order
    assert np.array_equal(v["inv"], g["unq_inv"])              # unq_inv over the compacted points
    assert np.array_equal(v["grid"], g["grid"])                # [ny, nx]


@pytest.mark.parametrize("case", READER_CASES)
def test_decorated_features(oracle, case):
    g = load_golden(case)
    v = oracle.voxelize(g["points"], g["pc_range"], g["voxel_size"])
    f = oracle.decorate(g["points"], v, g["pc_range"], g["voxel_size"])
    # raw columns and pillar-centre offsets are exact; cluster offsets depend on the sum order
    F = g["points"].shape[1] - 1
    assert np.array_equal(f[:, :F], g["features"][:, :F], equal_nan=True)
    assert np.array_equal(f[:, F + 3:], g["features"][:, F + 3:], equal_nan=True)
    np.testing.assert_allclose(f[:, F:F + 3], g["features"][:, F:F + 3], rtol=0, atol=2e-5)


@pytest.mark.parametrize("case", READER_CASES)
def test_reader_eval_feat_max(oracle, case):
    g = load_golden(case)
    r = oracle.reader_forward(g["points"], g["pc_range"], g["voxel_size"], list(g["num_filters"]), golden_layers(g),
                              eps=float(g["eps"]), B=int(g["coords"][:, 0].max()) + 1, want_canvas=True)
    assert np.array_equal(r["coords"], g["coords"])
    # tolerance stated by north_star / SURVEY H9: |d| <= 1e-4 + 1e-4 |ref| on fp32 features
    np.testing.assert_allclose(r["feat_max"], g["feat_max"], rtol=1e-4, atol=1e-4)
    c = r["canvas"]
    co = g["coords"]
    assert np.array_equal(c[co[:, 0], :, co[:, 1], co[:, 2]], r["feat_max"])
    assert np.count_nonzero(c) == np.count_nonzero(r["feat_max"])


def test_grid_size_rounding(oracle):
    assert list(oracle.grid_size([-50.4, -50.4, -5, 50.4, 50.4, 3], [0.075, 0.075, 8])) == [1344, 1344, 1]
    assert list(oracle.grid_size([-54, -54, -5, 54, 54, 3], [0.075, 0.075, 8])) == [1440, 1440, 1]
    assert list(oracle.grid_size([-76.8, -76.8, -2, 76.8, 76.8, 4], [0.075, 0.075, 6])) == [2048, 2048, 1]
    assert list(oracle.grid_size([-75.2, -75.2, -2, 75.2, 75.2, 4], [0.1, 0.1, 6])) == [1504, 1504, 1]


def test_iou_golden_bit_exact(oracle):
    g = load_golden("iou_bev_64")
    for a, b, key in [(g["boxes_a"], g["boxes_a"], "iou_aa"), (g["boxes_a"], g["boxes_b"], "iou_ab")]:
        o = oracle.boxes_iou_bev(a, b, "libm")
        assert np.array_equal(o.view(np.uint32), g[key].view(np.uint32))
        d = oracle.boxes_iou_bev(a, b, "det")
        np.testing.assert_allclose(d, g[key], rtol=0, atol=1e-5)
    al = oracle.boxes_aligned_iou_bev(g["boxes_a"][:40], g["boxes_b"], "libm")
    assert np.array_equal(al.view(np.uint32), g["iou_aligned"].view(np.uint32))


@pytest.mark.parametrize("name", ["n256_t020", "n256_t070", "n1000_t020", "n1000_t025", "n130_t020"])
def test_nms_golden_keep_indices(oracle, name):
    g = load_golden("nms_rotated")
    boxes, thr, keep = g[name + "_boxes"], float(g[name + "_thr"]), g[name + "_keep"]
    for math in ("libm", "det"):
        k = oracle.nms_rotated(boxes, thr, math)
        assert np.array_equal(k, keep), math
    sel = oracle.rotate_nms_pcdet(boxes, g[name + "_scores"], thr, pre_maxsize=1000, post_max_size=83, math="det")
    assert np.array_equal(sel, keep[:83])


def test_oracle_vs_compiled_reference_random(oracle):
    """oracle(libm) == reference iou3d_cpu.cpp bit-for-bit on fresh random boxes (needs oracle/_ref)."""
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built (reference tree absent and no prebuilt library)")
    from pillarnext_amd import synth

    a, _ = synth.clustered_boxes(300, 101, spread=10.0)
    b, _ = synth.clustered_boxes(200, 102, spread=10.0)
    r = oracle.ref_boxes_iou_bev(a, b)
    o = oracle.boxes_iou_bev(a, b, "libm")
    assert np.array_equal(r.view(np.uint32), o.view(np.uint32))
    assert (r > 0).mean() > 0.02
    np.testing.assert_allclose(oracle.boxes_iou_bev(a, b, "det"), r, rtol=0, atol=1e-5)
    assert np.array_equal(oracle.ref_nms_rotated(a, 0.2), oracle.nms_rotated(a, 0.2, "libm"))


def test_nms_edge_cases(oracle):
    assert len(oracle.nms_rotated(np.zeros((0, 7), np.float32), 0.2)) == 0
    one = np.array([[0, 0, 0, 2, 1, 1, 0.3]], np.float32)
    assert list(oracle.nms_rotated(one, 0.2)) == [0]
    same = np.repeat(one, 70, 0)
    assert list(oracle.nms_rotated(same, 0.2, "det")) == [0]
    assert list(oracle.nms_normal(same, 0.2)) == [0]

"""oracle.py -- TEST INFRASTRUCTURE ONLY: ctypes front-end to the CPU oracle (oracle/liborc.so).

Allowed importers: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.  The product
package (pillarnext_amd/) must never import this module -- tests/test_layout.py enforces that.

`ref_*` functions call the reference's own iou3d_cpu.cpp compiled in place (oracle/_ref/, built
by oracle/Makefile when /root/reference is mounted; the prebuilt .so travels to the GPU box).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None

_f32p = ctypes.POINTER(ctypes.c_float)
_i64p = ctypes.POINTER(ctypes.c_int64)
_i32p = ctypes.POINTER(ctypes.c_int32)
_u64p = ctypes.POINTER(ctypes.c_uint64)
_f64p = ctypes.POINTER(ctypes.c_double)


def _p(a, t):
    return a.ctypes.data_as(t)


def build(force=False):
    """Compile liborc.so (and oracle/_ref when the reference tree is present)."""
    so = os.path.join(_HERE, "liborc.so")
    srcs = [os.path.join(_HERE, f) for f in ("pnx_oracle.c", "orc_iou_impl.h")]
    srcs.append(os.path.join(_HERE, "..", "pillarnext_amd", "csrc", "pnx_detmath.h"))
    stale = force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "oracle"], stdout=subprocess.DEVNULL)
    ref_so = os.path.join(_HERE, "_ref", "libref_iou3d.so")
    if not os.path.exists(ref_so) and os.path.exists("/root/reference/det3d/core/iou3d_nms/src/iou3d_cpu.cpp"):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


def lib():
    global _LIB
    if _LIB is None:
        build()
        L = ctypes.CDLL(os.path.join(_HERE, "liborc.so"))
        for sfx in ("_libm", "_det"):
            getattr(L, "orc_nms_rotated" + sfx).restype = ctypes.c_int64
        L.orc_nms_normal.restype = ctypes.c_int64
        L.orc_voxelize.restype = ctypes.c_int64
        L.orc_reader_forward.restype = ctypes.c_int64
        _LIB = L
    return _LIB


def have_ref():
    return os.path.exists(os.path.join(_HERE, "_ref", "libref_iou3d.so"))


def ref():
    """The compiled reference (needs libtorch, loaded through torch)."""
    global _REF
    if _REF is None:
        import torch  # noqa: F401  (puts libtorch/libc10 in the process before dlopen)

        _REF = ctypes.CDLL(os.path.join(_HERE, "_ref", "libref_iou3d.so"))
    return _REF


def _boxes(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    assert a.ndim == 2 and a.shape[1] == 7
    return a


# ----------------------------------------------------------------------------- IoU / NMS
def boxes_iou_bev(a, b, math="libm"):
    a, b = _boxes(a), _boxes(b)
    out = np.zeros((a.shape[0], b.shape[0]), np.float32)
    getattr(lib(), "orc_boxes_iou_bev_" + math)(
        _p(a, _f32p), ctypes.c_int64(a.shape[0]), _p(b, _f32p), ctypes.c_int64(b.shape[0]), _p(out, _f32p))
    return out


def boxes_overlap_bev(a, b, math="libm"):
    a, b = _boxes(a), _boxes(b)
    out = np.zeros((a.shape[0], b.shape[0]), np.float32)
    getattr(lib(), "orc_boxes_overlap_bev_" + math)(
        _p(a, _f32p), ctypes.c_int64(a.shape[0]), _p(b, _f32p), ctypes.c_int64(b.shape[0]), _p(out, _f32p))
    return out


def boxes_aligned_overlap_bev(a, b, math="libm"):
    a, b = _boxes(a), _boxes(b)
    out = np.zeros((a.shape[0],), np.float32)
    getattr(lib(), "orc_boxes_aligned_overlap_bev_" + math)(_p(a, _f32p), _p(b, _f32p), ctypes.c_int64(a.shape[0]), _p(out, _f32p))
    return out


def boxes_aligned_iou_bev(a, b, math="libm"):
    a, b = _boxes(a), _boxes(b)
    out = np.zeros((a.shape[0],), np.float32)
    getattr(lib(), "orc_boxes_aligned_iou_bev_" + math)(_p(a, _f32p), _p(b, _f32p), ctypes.c_int64(a.shape[0]), _p(out, _f32p))
    return out


def boxes_aligned_iou3d(a, b, math="libm"):
    a, b = _boxes(a), _boxes(b)
    out = np.zeros((a.shape[0],), np.float32)
    lib().orc_boxes_aligned_iou3d(_p(a, _f32p), _p(b, _f32p), ctypes.c_int64(a.shape[0]), ctypes.c_int(math == "det"), _p(out, _f32p))
    return out


def nms_rotated(boxes, thresh, math="libm", return_mask=False):
    """boxes must already be score-sorted (descending). Returns kept indices (int64)."""
    boxes = _boxes(boxes)
    n = boxes.shape[0]
    keep = np.zeros((max(n, 1),), np.int64)
    mask = np.zeros((max(n, 1), max((n + 63) // 64, 1)), np.uint64)
    nk = getattr(lib(), "orc_nms_rotated_" + math)(
        _p(boxes, _f32p), ctypes.c_int64(n), ctypes.c_float(thresh), _p(keep, _i64p), _p(mask, _u64p))
    if return_mask:
        return keep[:nk].copy(), mask[:n, : (n + 63) // 64]
    return keep[:nk].copy()


def nms_normal(boxes, thresh):
    boxes = _boxes(boxes)
    n = boxes.shape[0]
    keep = np.zeros((max(n, 1),), np.int64)
    nk = lib().orc_nms_normal(_p(boxes, _f32p), ctypes.c_int64(n), ctypes.c_float(thresh), _p(keep, _i64p))
    return keep[:nk].copy()


def rotate_nms_pcdet(boxes, scores, thresh, pre_maxsize=None, post_max_size=None, math="libm"):
    """det3d/core/bbox/box_torch_ops.py:5-31 restated on numpy (stable argsort; fixtures avoid ties)."""
    boxes = _boxes(boxes)
    order = np.argsort(-np.asarray(scores, np.float32), kind="stable")
    if pre_maxsize is not None:
        order = order[:pre_maxsize]
    if len(order) == 0:
        return np.zeros((0,), np.int64)
    keep = nms_rotated(boxes[order], thresh, math=math)
    sel = order[keep]
    if post_max_size is not None:
        sel = sel[:post_max_size]
    return sel.astype(np.int64)


def ref_boxes_iou_bev(a, b):
    a, b = _boxes(a), _boxes(b)
    out = np.zeros((a.shape[0], b.shape[0]), np.float32)
    ref().ref_boxes_iou_bev_cpu(_p(a, _f32p), ctypes.c_int64(a.shape[0]), _p(b, _f32p), ctypes.c_int64(b.shape[0]), _p(out, _f32p))
    return out


def ref_boxes_aligned_iou_bev(a, b):
    a, b = _boxes(a), _boxes(b)
    out = np.zeros((a.shape[0], 1), np.float32)
    ref().ref_boxes_aligned_iou_bev_cpu(_p(a, _f32p), _p(b, _f32p), ctypes.c_int64(a.shape[0]), _p(out, _f32p))
    return out[:, 0]


def ref_nms_rotated(boxes, thresh):
    """Reference CPU IoU matrix + the greedy rule of iou3d_nms.cpp:144-155 (the reference has no
    CPU NMS; this is SURVEY.md section 8c's oracle for it)."""
    boxes = _boxes(boxes)
    n = boxes.shape[0]
    iou = ref_boxes_iou_bev(boxes, boxes)
    removed = np.zeros(n, bool)
    keep = []
    for i in range(n):
        if removed[i]:
            continue
        keep.append(i)
        removed[i + 1:] |= iou[i, i + 1:] > np.float32(thresh)
    return np.asarray(keep, np.int64)


# ----------------------------------------------------------------------------- reader
def grid_size(pc_range, voxel_size):
    pr = np.ascontiguousarray(pc_range, np.float64)
    vs = np.ascontiguousarray(voxel_size, np.float64)
    g = np.zeros(3, np.int64)
    lib().orc_grid_size(_p(pr, _f64p), _p(vs, _f64p), _p(g, _i64p))
    return g


def geom32(pc_range, voxel_size):
    """fp32 casts of the fp64 config values, as torch.from_numpy(...).type_as(points) gives (pe:91-93)."""
    return (np.asarray(pc_range[:3], np.float64).astype(np.float32), np.asarray(voxel_size, np.float64).astype(np.float32))


def voxelize(points, pc_range, voxel_size):
    pts = np.ascontiguousarray(points, np.float32)
    n, stride = pts.shape
    g = grid_size(pc_range, voxel_size)
    pc_min, vs = geom32(pc_range, voxel_size)
    kept = np.zeros(n + 1, np.int64)
    inv = np.zeros(n + 1, np.int64)
    coords = np.zeros((n + 1, 3), np.int32)
    P = ctypes.c_int64(0)
    m = lib().orc_voxelize(_p(pts, _f32p), ctypes.c_int64(n), ctypes.c_int(stride), _p(pc_min, _f32p), _p(vs, _f32p),
                           ctypes.c_int64(g[0]), ctypes.c_int64(g[1]), _p(kept, _i64p), _p(inv, _i64p), _p(coords, _i32p),
                           ctypes.byref(P))
    return dict(kept=kept[:m].copy(), inv=inv[:m].copy(), coords=coords[: P.value].copy(), P=P.value,
                grid=np.array([g[1], g[0]], np.int64))


def decorate(points, vox, pc_range, voxel_size):
    pts = np.ascontiguousarray(points, np.float32)
    stride = pts.shape[1]
    pc_min, vs = geom32(pc_range, voxel_size)
    m = len(vox["kept"])
    feat = np.zeros((m, stride + 4), np.float32)
    lib().orc_decorate(_p(pts, _f32p), ctypes.c_int(stride), _p(vox["kept"], _i64p), _p(vox["inv"], _i64p), ctypes.c_int64(m),
                       ctypes.c_int64(vox["P"]), _p(pc_min, _f32p), _p(vs, _f32p), _p(feat, _f32p))
    return feat


def pack_pfn_params(layers):
    """layers: list of dicts W (units,cin), gamma, beta, mean, var -> flat fp32 vector."""
    out = []
    for L in layers:
        for k in ("W", "gamma", "beta", "mean", "var"):
            out.append(np.asarray(L[k], np.float32).reshape(-1))
    return np.ascontiguousarray(np.concatenate(out))


def pfn_eval(feat, inv, P, num_filters, layers, eps=1e-3):
    feat = np.ascontiguousarray(feat, np.float32)
    inv = np.ascontiguousarray(inv, np.int64)
    nf = np.ascontiguousarray([feat.shape[1]] + list(num_filters), np.int32)
    params = pack_pfn_params(layers)
    out = np.zeros((P, nf[-1]), np.float32)
    lib().orc_pfn_eval(_p(feat, _f32p), ctypes.c_int64(feat.shape[0]), _p(inv, _i64p), ctypes.c_int64(P),
                       ctypes.c_int(len(num_filters)), nf.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), _p(params, _f32p),
                       ctypes.c_float(eps), _p(out, _f32p))
    return out


def scatter_canvas(feat_max, coords, B, gy, gx):
    fm = np.ascontiguousarray(feat_max, np.float32)
    co = np.ascontiguousarray(coords, np.int32)
    C = fm.shape[1]
    canvas = np.zeros((B, C, gy, gx), np.float32)
    lib().orc_scatter_canvas(_p(fm, _f32p), _p(co, _i32p), ctypes.c_int64(fm.shape[0]), ctypes.c_int(C), ctypes.c_int64(B),
                             ctypes.c_int64(gy), ctypes.c_int64(gx), _p(canvas, _f32p))
    return canvas


def reader_forward(points, pc_range, voxel_size, num_filters, layers, eps=1e-3, B=None, want_canvas=False):
    """PillarFeatureNet.forward (eval): returns feat_max (P,C), coords (P,3) int32 [b,y,x], grid [ny,nx]."""
    pts = np.ascontiguousarray(points, np.float32)
    n, stride = pts.shape
    g = grid_size(pc_range, voxel_size)
    pc_min, vs = geom32(pc_range, voxel_size)
    nf = np.ascontiguousarray([stride + 4] + list(num_filters), np.int32)
    params = pack_pfn_params(layers)
    coords = np.zeros((n + 1, 3), np.int32)
    fm = np.zeros((n + 1, nf[-1]), np.float32)
    canvas = None
    cptr = None
    if want_canvas:
        canvas = np.zeros((B, int(nf[-1]), int(g[1]), int(g[0])), np.float32)
        cptr = _p(canvas, _f32p)
    P = lib().orc_reader_forward(_p(pts, _f32p), ctypes.c_int64(n), ctypes.c_int(stride), _p(pc_min, _f32p), _p(vs, _f32p),
                                 ctypes.c_int64(g[0]), ctypes.c_int64(g[1]), ctypes.c_int(len(num_filters)),
                                 nf.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), _p(params, _f32p), ctypes.c_float(eps),
                                 _p(coords, _i32p), _p(fm, _f32p), ctypes.c_int64(B or 0), cptr)
    res = dict(feat_max=fm[:P].copy(), coords=coords[:P].copy(), grid=np.array([g[1], g[0]], np.int64), P=int(P))
    if want_canvas:
        res["canvas"] = canvas
    return res


# ----------------------------------------------------------------------------- multi-sweep merge (numpy restatement)
def merge_sweeps(sweeps, n_copy=4):
    """Restates det3d/datasets/nuscenes/nusc.py:76-121 (read_sweep: fp64 `transform.dot(vstack(xyz, 1))[:3]` stored back into the
    float32 array, remove_close :91-99 on past sweeps only, time-lag column), det3d/datasets/waymo/waymo.py:49-67 (same with
    `xyz1 @ rel_pose.T`, timestamp column, no close-point filter) and collate.py:15-22 (batch index column).  PINNED: tests/golden/
    merge_sweeps.npz holds the output of the reference's own NuScenesDataset.load_pointcloud / WaymoDataset.load_pointcloud on raw sweep
    files (oracle/gen_golden.py merge: the devkit imports of those modules are stood in for at import only, their numpy code runs
    unmodified); tests/test_merge_golden.py checks this function and the device merge against it.
    sweeps: list of dicts {points (n, C) fp32, batch, time, radius, transform (4x4 fp64 or None)} in concatenation order."""
    rows = []
    for s in sweeps:
        p = np.array(s["points"][:, :n_copy], dtype=np.float32).T.copy()          # (C, n) as read_sweep holds it
        T = s.get("transform")
        if T is not None:
            p[:3, :] = np.asarray(T, np.float64).dot(np.vstack((p[:3, :], np.ones(p.shape[1]))))[:3, :]
        r = float(s.get("radius", 0.0))
        if r > 0:
            close = np.logical_and(np.abs(p[0, :]) < r, np.abs(p[1, :]) < r)
            p = p[:, np.logical_not(close)]
        t = (float(s["time"]) * np.ones((1, p.shape[1]))).astype(np.float32)
        b = np.full((1, p.shape[1]), float(s["batch"]), np.float32)
        rows.append(np.vstack((b, p, t)).T)
    return np.concatenate(rows, axis=0).astype(np.float32)

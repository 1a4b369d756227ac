"""CPU: the C-ABI library builds, loads, and exports exactly the symbols include/pnx.h declares."""
import ctypes
import os
import re

from conftest import ROOT


def header_functions():
    src = open(os.path.join(ROOT, "include", "pnx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pnx_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from pillarnext_amd import _lib, build

    build.build()
    L = ctypes.CDLL(_lib.LIB_PATH)
    names = header_functions()
    assert len(names) >= 18
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/pnx.h but not exported"
    assert sorted(_lib.PROTOTYPES) == names, "pillarnext_amd/_lib.py PROTOTYPES out of sync with include/pnx.h"


def test_host_helpers_without_gpu():
    from pillarnext_amd import _lib

    L = _lib.lib()
    assert b"gfx950" in L.pnx_version()
    g = _lib.make_geom([-50.4, -50.4, -5.0, 50.4, 50.4, 3.0], [0.075, 0.075, 8])
    assert (g.gx, g.gy) == (1344, 1344)  # 100.8/0.075 = 1344.0000000000002 -> np.round -> 1344
    g = _lib.make_geom([-76.8, -76.8, -2, 76.8, 76.8, 4], [0.075, 0.075, 6])
    assert (g.gx, g.gy) == (2048, 2048)
    assert abs(g.voxel[0] - 0.075) < 1e-8 and g.voxel[0] != 0.075  # fp32 cast of the fp64 config value
    n = L.pnx_reader_workspace_bytes(300000, 1, ctypes.byref(g))
    assert 0 < n < 1 << 30
    # argument validation needs no GPU: null geometry, bad stride
    rc = L.pnx_reader_forward(None, 10, 6, 1, None, None, None, 0, 0, None, None, None, 0, None, None, None, None, 0, None)
    assert rc == -1 and b"geom" in L.pnx_last_error()
    try:
        _lib.make_geom([0, 0, 0, -1, 1, 1], [0.1, 0.1, 1])
        assert False
    except _lib.PnxError as e:
        assert "bad range" in str(e)


def test_product_never_touches_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may use oracle/."""
    pkg = os.path.join(ROOT, "pillarnext_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "oracle" not in txt.replace("oracle/pnx_oracle.c's", "").lower() or f in ("iou3d.hip",), f


def test_conv_entry_points_validate_before_launching():
    """Argument checks of the convolution entry points need no GPU: they come before any HIP call."""
    from pillarnext_amd import _lib

    L = _lib.lib()
    buf = (ctypes.c_char * 64)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    rc = L.pnx_conv3x3_bf16(None, None, None, None, None, None, 1, 8, 8, 64, 64, 1, 1, None, None, None, None)
    assert rc < 0 and b"bad arguments" in L.pnx_last_error()
    rc = L.pnx_conv3x3_bf16(p, p, p, None, None, p, 1, 8, 8, 64, 64, 3, 1, None, None, None, None)
    assert rc < 0 and b"stride" in L.pnx_last_error()
    rc = L.pnx_conv3x3_bf16(p, p, p, None, None, p, 1, 8, 8, 64, 64, 1, 1, p, None, None, None)       # row_dirty without a mask
    assert rc < 0 and b"row_dirty" in L.pnx_last_error()
    rc = L.pnx_conv3x3_bf16(p, p, p, None, p, p, 1, 8, 8, 48, 64, 1, 1, None, None, None, None)       # no kernel for 48 input channels
    assert rc < 0 and b"48" in L.pnx_last_error()
    rc = L.pnx_conv3x3_bf16(p, p, p, None, p, p, 1, 8, 8, 64, 128, 2, 1, None, p, p, None)       # the strided kernels take no tile list
    assert rc < 0 and b"tile list" in L.pnx_last_error()
    rc = L.pnx_conv3x3_bf16(p, p, p, None, p, p, 1, 8, 8, 64, 64, 1, 1, None, p, None, None)         # list without its count
    assert rc < 0 and b"tile_count" in L.pnx_last_error()
    assert L.pnx_conv3x3_tile_rows(64, 64, 1) == 16 and L.pnx_conv3x3_tile_rows(256, 256, 1) == 8 and L.pnx_conv3x3_tile_rows(64, 128, 2) == 0
    rc = L.pnx_sephead_out_bf16(p, p, p, p, 1, 8, 8, 3, None)
    assert rc < 0 and b"branches" in L.pnx_last_error()
    rc = L.pnx_sephead_out_bf16(None, p, p, p, 1, 8, 8, 6, None)
    assert rc < 0 and b"bad arguments" in L.pnx_last_error()


def test_round2_entry_points_validate_before_launching():
    """The entry points added in round 2 reject bad arguments before any HIP call (runs without a GPU)."""
    from pillarnext_amd import _lib

    L = _lib.lib()
    buf = (ctypes.c_char * 64)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    arr = (ctypes.c_void_p * 9)(*([p.value] * 9))
    assert L.pnx_sum_bias_act(arr, 9, p, p, 10, 64, 1, 1, None) < 0 and b"summands" in L.pnx_last_error()
    assert L.pnx_sum_bias_act(arr, 2, p, p, 10, 60, 1, 1, None) < 0 and b"multiple of 8" in L.pnx_last_error()
    assert L.pnx_sum_bias_act(None, 2, p, p, 10, 64, 1, 1, None) < 0 and b"bad arguments" in L.pnx_last_error()
    assert L.pnx_deconv2x2_bf16(p, p, p, p, 1, 8, 8, 128, 64, 1, None) < 0 and b"128" in L.pnx_last_error()
    assert L.pnx_deconv2x2_bf16(None, p, p, p, 1, 8, 8, 64, 64, 1, None) < 0 and b"bad arguments" in L.pnx_last_error()
    assert L.pnx_conv_tile_list(None, arr, 1, 1, 8, 8, 16, p, p, None) < 0 and b"bad arguments" in L.pnx_last_error()
    assert L.pnx_conv_tile_list(p, arr, 5, 1, 8, 8, 16, p, p, None) < 0 and b"row_dirty" in L.pnx_last_error()
    assert L.pnx_sort_keys(p, 0, 4, p, p, p, 1 << 20, None) < 0 and b"bad arguments" in L.pnx_last_error()
    assert L.pnx_sort_keys(p, 1000, 4, p, p, p, 8, None) < 0 and b"workspace" in L.pnx_last_error()
    assert L.pnx_sort_keys_workspace_bytes(0) >= 256


def test_decode_descriptor_layout_matches_the_library():
    from pillarnext_amd import _lib
    from pillarnext_amd.decode import pack_task

    blob = pack_task(16, True, 2, 3, 360, 360, 4, (0.075, 0.075), (-54.0, -54.0), 0.1, [-61.2, -61.2, -10, 61.2, 61.2, 10], [0.5, 0.5])
    assert len(blob) == _lib.lib().pnx_decode_task_desc_bytes() == 104  # 92 + o_hm, o_iou, lazy

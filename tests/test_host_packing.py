"""CPU: host-side packing helpers (plain torch, no GPU): the second-convolution weight layout of the lazy SepHead evaluator and the lazy
flavour of the decoder's task descriptor."""
import struct

import numpy as np
import torch

from pillarnext_amd import ops
from pillarnext_amd.decode import pack_task


def test_lazy_second_conv_weight_layout():
    off, k = [0, 2, 3, 6, 8], [2, 1, 3, 2, 2]
    g = torch.Generator().manual_seed(1)
    w2m = torch.zeros((9 * 320, 10))
    for pos in range(9):
        for j in range(5):
            w2m[pos * 320 + 64 * j: pos * 320 + 64 * (j + 1), off[j]:off[j] + k[j]] = torch.randn((64, k[j]), generator=g)
    p = ops.sephead_lazy_pack_w2(w2m)
    assert p.shape == (10, 9, 32, 3)
    for mt in range(10):
        j = mt // 2
        for pos in (0, 4, 8):
            for cl in (0, 17, 31):
                row = w2m[pos * 320 + mt * 32 + cl]
                assert torch.equal(p[mt, pos, cl, :k[j]], row[off[j]:off[j] + k[j]])
                assert bool((p[mt, pos, cl, k[j]:] == 0).all())
                assert float(row.abs().sum()) == float(p[mt, pos, cl].abs().sum())     # nothing outside the branch's own outputs


def test_lazy_task_descriptor_points_at_the_class_map_only():
    args = (16, True, 2, 3, 360, 360, 4, (0.075, 0.075), (-54.0, -54.0), 0.1, [-61.2, -61.2, -10, 61.2, 61.2, 10], [0.5, 0.5])
    dense, lazy = pack_task(*args), pack_task(*args, lazy=True)
    assert len(dense) == len(lazy) == 104
    d, z = struct.unpack("6i6f6fi4f3i", dense), struct.unpack("6i6f6fi4f3i", lazy)
    assert d[:23] == z[:23]                                  # geometry, thresholds, range, rectifier: the same
    assert d[-3:] == (11, 10, 0) and z[-3:] == (1, 0, 1)     # o_hm, o_iou, lazy: [reg .. vel | iou | hm] vs [iou | hm]


def test_launch_table_structures_mirror_the_c_ones():
    """plan.PnxOp / decode._PnxLazyDecode are ctypes mirrors of include/pnx.h's pnx_op / pnx_lazy_decode: same size as the compiled ones,
    pointers 8-byte aligned behind the integer block."""
    import ctypes

    from pillarnext_amd import _lib, decode, plan

    L = _lib.lib()
    assert ctypes.sizeof(plan.PnxOp) == L.pnx_op_bytes() == 128
    assert plan.PnxOp.p.offset == 40 and plan.PnxOp.i.offset == 4
    assert ctypes.sizeof(decode._PnxLazyDecode) == L.pnx_lazy_decode_bytes()
    assert decode._PnxLazyDecode.dense_host.offset == 24 and decode._PnxLazyDecode.flag_host.offset == L.pnx_lazy_decode_bytes() - 8

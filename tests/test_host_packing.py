"""CPU: host-side packing helpers added in round 3 (plain torch, no GPU): the occupancy words + rank prefix of the sparse tensor view, the
second-convolution weight layout of the lazy SepHead evaluator, and the lazy flavour of the decoder's task descriptor."""
import struct

import numpy as np
import torch

from pillarnext_amd import ops
from pillarnext_amd.decode import pack_task


def test_sparse_index_from_mask_matches_a_numpy_statement():
    rng = np.random.default_rng(3)
    B, gx, gy = 2, 7, 70                                     # gy not a multiple of 32: the last word of a row is partly padding
    m = rng.random((B, gx, gy)) < 0.3
    w, wpr = ops.sparse_index_from_mask(torch.from_numpy(m))
    assert wpr == 3 and w.shape == (B * gx * wpr, 2) and w.dtype == torch.int32
    bits = w[:, 0].numpy().astype(np.int64) & 0xFFFFFFFF
    pre = w[:, 1].numpy()
    rank = 0
    for b in range(B):
        for xi in range(gx):
            for k in range(wpr):
                word = 0
                for j in range(32):
                    yi = 32 * k + j
                    if yi < gy and m[b, xi, yi]:
                        word |= 1 << j
                i = (b * gx + xi) * wpr + k
                assert bits[i] == word and pre[i] == rank    # rank of the word's first active cell = active cells before it in (b, xi, yi) order
                rank += bin(word).count("1")
    assert rank == int(m.sum())


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

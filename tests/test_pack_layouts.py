"""Host-side weight packers of the HIP convolutions: the fragment orders documented in csrc/conv3x3.hip, checked element by
element against an explicit index formula (CPU only; the kernels themselves are covered by tests/test_gpu_dense_ops.py)."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")


def test_conv3x3_pack_weights_fragment_order():
    from pillarnext_amd.ops import conv3x3_pack_weights

    co, ci = 64, 32
    w = torch.arange(co * ci * 9, dtype=torch.float32).reshape(co, ci, 3, 3) % 251  # exactly representable in bf16
    f = conv3x3_pack_weights(w).float().numpy().reshape(9, ci // 16, co // 32, 64, 8)
    wn = w.numpy()
    rng = np.random.default_rng(0)
    for _ in range(500):
        tap, cb, mt, lane, e = (int(rng.integers(n)) for n in (9, ci // 16, co // 32, 64, 8))
        kb, n = lane >> 5, lane & 31
        assert f[tap, cb, mt, lane, e] == wn[mt * 32 + n, cb * 16 + 8 * kb + e, tap // 3, tap % 3]


def test_sephead_pack_weights_fragment_order():
    from pillarnext_amd.ops import sephead_pack_weights

    nb = 6
    w = torch.arange(16 * nb * 64 * 9, dtype=torch.float32).reshape(16, nb * 64, 3, 3) % 241
    f = sephead_pack_weights(w).float().numpy().reshape(nb, 9, 2, 64, 8)
    wn = w.numpy()
    rng = np.random.default_rng(1)
    for _ in range(500):
        j, tap, kc, lane, e = (int(rng.integers(n)) for n in (nb, 9, 2, 64, 8))
        q, o = lane >> 4, lane & 15
        assert f[j, tap, kc, lane, e] == wn[o, j * 64 + kc * 32 + q * 8 + e, tap // 3, tap % 3]


def test_sephead_pack_rejects_other_shapes():
    from pillarnext_amd._lib import PnxError
    from pillarnext_amd.ops import sephead_pack_weights

    with pytest.raises(PnxError):
        sephead_pack_weights(torch.zeros((8, 384, 3, 3)))
    with pytest.raises(PnxError):
        sephead_pack_weights(torch.zeros((16, 100, 3, 3)))


def test_deconv2x2_pack_weights_fragment_order():
    """ConvTranspose2d weight (Cin, Cout, 2, 2) -> [parity ky*2+kx][cin/16][cout/32][lane = kb*32 + n][8] (csrc/conv3x3.hip::k_deconv2x2_64)."""
    from pillarnext_amd.ops import deconv2x2_pack_weights

    ci, co = 64, 64
    w = torch.arange(ci * co * 4, dtype=torch.float32).reshape(ci, co, 2, 2) % 251
    f = deconv2x2_pack_weights(w).float().numpy().reshape(4, ci // 16, co // 32, 64, 8)
    wn = w.numpy()
    rng = np.random.default_rng(2)
    for _ in range(500):
        p, cb, mt, lane, e = (int(rng.integers(n)) for n in (4, ci // 16, co // 32, 64, 8))
        kb, n = lane >> 5, lane & 31
        assert f[p, cb, mt, lane, e] == wn[cb * 16 + 8 * kb + e, mt * 32 + n, p >> 1, p & 1]


def test_kernel_shape_tables_are_consistent_with_the_library():
    """ops.CONV3X3_SHAPES_S1: the stride-1 shapes the library has kernels for; those served by the LDS-slab kernels also take tile lists."""
    from pillarnext_amd import ops

    for ci, co in ops.CONV3X3_SHAPES_S1:
        rows = ops.conv_tile_rows(ci, co, 1)
        assert rows in ((16,) if ci == 64 and co != 128 else (8, 0)), (ci, co, rows)
    for ci, co in ops.CONV3X3_SHAPES_S2:
        assert ops.conv_tile_rows(ci, co, 2) == 0        # the strided kernels walk all tiles

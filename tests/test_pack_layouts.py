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

"""CPU: the host side of the launch tables (pillarnext_amd/plan.py -> include/pnx.h: pnx_op / pnx_enqueue).  Tables are built from tensors'
shapes and addresses, so everything but run() works without a GPU; run() itself refuses CPU tensors, and pnx_enqueue reports the index
of an entry it cannot forward before anything is launched."""
import ctypes

import pytest
import torch

from pillarnext_amd import _lib, plan
from pillarnext_amd._lib import PnxError


def _nhwc(b, c, h, w):
    return torch.zeros((b, c, h, w), dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)


def test_table_entries_mirror_the_calls_they_name():
    B, H, W = 2, 10, 40
    x, y = _nhwc(B, 64, H, W), _nhwc(B, 64, H, W)
    mask, pooled = torch.zeros((B, H, W), dtype=torch.uint8), torch.zeros((B, H, W), dtype=torch.uint8)
    dirty = [torch.zeros((B, H, 2), dtype=torch.uint8) for _ in range(3)]
    tiles = (torch.zeros((B * 2,), dtype=torch.int32), torch.zeros((1,), dtype=torch.int32))
    wfrag, bias = torch.zeros(9 * 64 * 64, dtype=torch.bfloat16), torch.zeros(64)
    p = plan.LaunchPlan()
    p.mask_pool3(mask, pooled, 1)
    p.tile_list(pooled, dirty, 16, tiles)
    p.conv3x3(plan.Dyn("x", x), wfrag, bias, 64, 1, pooled, None, True, out=(y, dirty[0]), tiles=tiles)
    up = _nhwc(B, 64, 2 * H, 2 * W)
    p.deconv2x2(y, wfrag, bias, 64, plan.Dyn("up", up))
    p.freeze()
    a = p._arr
    assert len(p) == 4 and [a[k].kind for k in range(4)] == [plan.OP_MASK_POOL3, plan.OP_TILE_LIST, plan.OP_CONV3X3, plan.OP_DECONV2X2]
    assert list(a[0].i[:4]) == [B, H, W, 1] and a[0].p[0] == mask.data_ptr() and a[0].p[1] == pooled.data_ptr()
    assert list(a[1].i[:5]) == [3, B, H, W, 16] and [a[1].p[k] for k in range(6)] == [pooled.data_ptr(), tiles[0].data_ptr(), tiles[1].data_ptr()] + [d.data_ptr() for d in dirty]
    assert list(a[2].i[:7]) == [B, H, W, 64, 64, 1, 1]
    assert [a[2].p[k] for k in range(9)] == [x.data_ptr(), wfrag.data_ptr(), bias.data_ptr(), None, pooled.data_ptr(), y.data_ptr(), dirty[0].data_ptr(),
                                            tiles[0].data_ptr(), tiles[1].data_ptr()]
    assert list(a[3].i[:6]) == [B, H, W, 64, 64, 1] and a[3].p[3] == up.data_ptr()
    # re-binding patches exactly the entries that named the tensor; a tensor of another shape is refused; a frozen plan takes no entries
    x2, up2 = _nhwc(B, 64, H, W), _nhwc(B, 64, 2 * H, 2 * W)
    p.bind("x", x2)
    p.bind("up", up2)
    assert a[2].p[0] == x2.data_ptr() and a[3].p[3] == up2.data_ptr() and a[2].p[5] == y.data_ptr()
    with pytest.raises(PnxError):
        p.bind("x", _nhwc(B, 64, H + 1, W))
    with pytest.raises(PnxError):
        p.mask_pool3(mask, pooled, 1)
    with pytest.raises(PnxError):
        p.run()                                  # CPU tensors: there is no CPU path


def test_builders_check_shapes_like_the_wrappers_do():
    p = plan.LaunchPlan()
    x = _nhwc(1, 64, 8, 32)
    with pytest.raises(PnxError):
        p.conv3x3(x, torch.zeros(1), torch.zeros(64), 64, 1, None, None, True, out=(_nhwc(1, 64, 8, 31), None))
    with pytest.raises(PnxError):
        p.deconv2x2(x, torch.zeros(1), torch.zeros(64), 64, _nhwc(1, 64, 8, 32))
    with pytest.raises(PnxError):
        p.sephead_out(x, torch.zeros(1), torch.zeros(16), _nhwc(1, 64, 8, 32))
    with pytest.raises(PnxError):
        p.conv3x3(x.float(), torch.zeros(1), torch.zeros(64), 64, 1, None, None, True, out=(_nhwc(1, 64, 8, 32), None))


def test_enqueue_names_the_entry_it_cannot_forward():
    L = _lib.lib()
    ops = (plan.PnxOp * 2)()
    ops[0].kind = 99                              # not a call of the library: refused before anything is launched
    rc = L.pnx_enqueue(ops, 1, None)
    assert rc == _lib.PNX_ERR_INVALID if hasattr(_lib, "PNX_ERR_INVALID") else rc < 0
    msg = L.pnx_last_error().decode()
    assert "entry 0" in msg and "99" in msg
    assert L.pnx_enqueue(ops, 0, None) == 0       # an empty table is a no-op
    assert L.pnx_enqueue(None, 1, None) < 0
    assert L.pnx_decode_lazy_enqueue(None, None) < 0


def test_training_entry_points_validate_before_launching():
    """pnx_conv3x3_wgrad_* / pnx_conv3x3_pack_weights: sizes and argument checks (no GPU work)."""
    L = _lib.lib()
    assert L.pnx_conv3x3_wgrad_workspace_bytes(64, 64) == 512 * 9 * 4096 * 4 + 256          # 512 workgroups x one 64 x 64 x 9 fp32 partial
    assert L.pnx_conv3x3_wgrad_workspace_bytes(256, 256) == 16 * 32 * 9 * 4096 * 4 + 256      # 16 block pairs x 32 workgroups
    assert L.pnx_conv3x3_wgrad_workspace_bytes(48, 64) == 0 and L.pnx_conv3x3_wgrad_workspace_bytes(64, 100) == 0
    assert L.pnx_conv3x3_wgrad_bf16(None, None, None, None, 1, 8, 8, 64, 64, 1, None, 0, None) < 0
    assert L.pnx_conv3x3_pack_weights(None, 0, 64, 64, 0, None, None) < 0

"""Host side of the fused training-mode PFN (csrc/pfn_train.hip behind pnx_pfn_forward_train / pnx_pfn_backward).

What the reference does per training step (det3d/models/readers/pillar_encoder.py:35-50 x2, :174-182) as ~20 autograd nodes over
(N',32/64) tensors is here ONE autograd.Function whose forward and backward are passes over the pillar-sorted point records; between
the passes this module reduces the kernels' partial sums in fp64, exchanges BatchNorm statistics when the reader was put into
synchronised mode (models.convert_sync_batchnorm -- the same 65/129-float all-reduces torch's SyncBatchNorm makes, tools/train.py:56),
updates the running statistics (momentum 0.01, unbiased variance) and does the BatchNorm/Linear weight-gradient algebra:

  forward   F1 = sum f, F2 = sum f f^T  ->  mean/var of x0 = W0 f;   U1 = sum u, U2 = sum u u^T  ->  mean/var of x1 = W1 u;   out
  backward  D1 = sum dz1, D2 = sum dz1*xhat1, A = sum dz1^T u        ->  dgamma1, dbeta1, dW1 = g1*is1*(A - D1/N U1^T - D2/N Xhat1U)
            E1 = sum dz0, E2 = sum dz0*xhat0, B0 = sum dz0^T f       ->  dgamma0, dbeta0, dW0 likewise, with
            Xhat1U = is1 * (W1 U2 - mu1 U1^T) and Xhat0F = is0 * (W0 F2 - mu0 F1^T)  (x is linear in u / f, so no extra pass)
Parameter gradients are the LOCAL sums (DistributedDataParallel averages them), statistics are global -- as in SyncBatchNorm."""
import ctypes

import torch

from . import _lib
from ._lib import check, lib, ptr, stream_ptr


def _all_reduce(t, group):
    if group is not False:
        from .dist_utils import all_reduce_sum

        all_reduce_sum(t, group)
    return t


def _params(F, w0, w1, mu0, is0, g0, b0, mu1, is1, g1, b1, m1=None, m2=None):
    z = torch.zeros(64, dtype=torch.float32, device=w0.device)
    parts = [w1.reshape(-1), mu1, is1, g1, b1, z if m1 is None else m1, z if m2 is None else m2, mu0, is0, g0, b0, w0.reshape(-1)]
    out = torch.cat([p.float() for p in parts]).contiguous()
    assert out.numel() == lib().pnx_pfn_train_param_floats(F)
    return out


class FusedPFNTrain(torch.autograd.Function):
    """feat_max (P,64), coords (P,3) = f(points; W0, gamma0, beta0, W1, gamma1, beta1) with batch statistics."""

    @staticmethod
    def forward(ctx, points, w0, g0, b0, w1, g1, b1, mod, batch):
        L = lib()
        n, stride = points.shape
        F, C0 = stride - 1, stride + 4
        dev = points.device
        geom = mod._geom
        group = mod.sync_group if mod.sync else False
        eps = float(mod.pfn_layers[0].norm.eps)
        ws = torch.empty(int(L.pnx_reader_workspace_bytes(n, batch, ctypes.byref(geom))) + 256, dtype=torch.uint8, device=dev)
        cap = max(min(n, batch * int(geom.gx) * int(geom.gy)), 1)
        coords = torch.empty((cap, 3), dtype=torch.int32, device=dev)
        counts = torch.zeros(2, dtype=torch.int32, device=dev)
        out = torch.empty((cap, 64), dtype=torch.float32, device=dev)
        nb = int(L.pnx_pfn_train_partial_floats(F, 0)) // (C0 + C0 * C0)

        def fwd(pass_, prm, part):
            check(L.pnx_pfn_forward_train(pass_, ptr(points), n, stride, batch, ctypes.byref(geom), ptr(prm), ptr(part), ptr(out), cap, ptr(coords),
                                          None, ptr(counts), ptr(ws), ws.numel(), stream_ptr()), f"pnx_pfn_forward_train({pass_})")

        w0d, w1d = w0.detach().double(), w1.detach().double()
        pa = torch.empty(int(L.pnx_pfn_train_partial_floats(F, 0)), dtype=torch.float32, device=dev)
        fwd(0, None, pa)
        P, n_kept = (int(v) for v in counts.tolist())  # the output shape is data dependent (as in the reference): one sync
        sa = pa.view(nb, C0 + C0 * C0).sum(0, dtype=torch.float64)
        F1, F2 = sa[:C0], sa[C0:].view(C0, C0)
        st0 = torch.cat([w0d @ F1, ((w0d @ F2) * w0d).sum(1), torch.tensor([float(n_kept)], dtype=torch.float64, device=dev)])
        st0 = _all_reduce(st0, group)
        N = max(float(st0[-1]), 1.0)
        mu0 = st0[:32] / N
        var0 = (st0[32:64] / N - mu0 * mu0).clamp(min=0.0)
        is0 = torch.rsqrt(var0 + eps)
        prm = _params(F, w0.detach(), w1.detach(), mu0, is0, g0.detach(), b0.detach(), torch.zeros(64, device=dev), torch.ones(64, device=dev), g1.detach(), b1.detach())
        nw = nb * 4
        pb = torch.empty(int(L.pnx_pfn_train_partial_floats(F, 1)), dtype=torch.float32, device=dev)
        fwd(1, prm, pb)
        sb = pb.view(nw, 64, 65).sum(0, dtype=torch.float64)
        U2, U1 = sb[:, :64], sb[:, 64]
        st1 = _all_reduce(torch.cat([w1d @ U1, ((w1d @ U2) * w1d).sum(1)]), group)
        mu1 = st1[:64] / N
        var1 = (st1[64:] / N - mu1 * mu1).clamp(min=0.0)
        is1 = torch.rsqrt(var1 + eps)
        prm = _params(F, w0.detach(), w1.detach(), mu0, is0, g0.detach(), b0.detach(), mu1, is1, g1.detach(), b1.detach())
        fwd(2, prm, None)
        with torch.no_grad():  # running statistics: momentum, unbiased variance (BatchNorm1d semantics)
            unb = N / max(N - 1.0, 1.0)
            for norm, mu, var in ((mod.pfn_layers[0].norm, mu0, var0), (mod.pfn_layers[1].norm, mu1, var1)):
                m = norm.momentum
                norm.running_mean.mul_(1 - m).add_(mu.to(norm.running_mean.dtype), alpha=m)
                norm.running_var.mul_(1 - m).add_((var * unb).to(norm.running_var.dtype), alpha=m)
                norm.num_batches_tracked += 1
        fm = out[:P]
        ctx.save_for_backward(points, w0, g0, b0, w1, g1, b1, fm)
        ctx.misc = (ws, geom, batch, group, N, mu0, is0, mu1, is1, F1, F2, U1, U2, nb)
        ctx.mark_non_differentiable(coords)
        return fm, coords[:P]

    @staticmethod
    def backward(ctx, grad_fm, _grad_coords):
        L = lib()
        points, w0, g0, b0, w1, g1, b1, fm = ctx.saved_tensors
        ws, geom, batch, group, N, mu0, is0, mu1, is1, F1, F2, U1, U2, nb = ctx.misc
        n, stride = points.shape
        F, C0 = stride - 1, stride + 4
        dev = points.device
        nw = nb * 4
        G = grad_fm.contiguous().float()
        w0d, w1d = w0.detach().double(), w1.detach().double()

        def bwd(pass_, prm, part):
            check(L.pnx_pfn_backward(pass_, n, stride, batch, ctypes.byref(geom), ptr(prm), ptr(G), ptr(fm), ptr(part), ptr(ws), ws.numel(), stream_ptr()),
                  f"pnx_pfn_backward({pass_})")

        prm = _params(F, w0.detach(), w1.detach(), mu0, is0, g0.detach(), b0.detach(), mu1, is1, g1.detach(), b1.detach())
        pd = torch.empty(int(L.pnx_pfn_train_partial_floats(F, 3)), dtype=torch.float32, device=dev)
        bwd(0, prm, pd)
        sd = pd.view(nw, 64, 66).sum(0, dtype=torch.float64)
        A, D1, D2 = sd[:, :64], sd[:, 64], sd[:, 65]
        d12 = _all_reduce(torch.cat([D1, D2]), group) / N
        m1, m2 = d12[:64], d12[64:]
        a1 = g1.detach().double() * is1
        xh1u = is1[:, None] * (w1d @ U2 - mu1[:, None] * U1[None, :])
        dW1 = a1[:, None] * (A - m1[:, None] * U1[None, :] - m2[:, None] * xh1u)
        prm = _params(F, w0.detach(), w1.detach(), mu0, is0, g0.detach(), b0.detach(), mu1, is1, g1.detach(), b1.detach(), m1, m2)
        pe = torch.empty(int(L.pnx_pfn_train_partial_floats(F, 4)), dtype=torch.float32, device=dev)
        bwd(1, prm, pe)
        se = pe.view(nw, 32, C0 + 2).sum(0, dtype=torch.float64)
        B0, E1, E2 = se[:, :C0], se[:, C0], se[:, C0 + 1]
        e12 = _all_reduce(torch.cat([E1, E2]), group) / N
        e1, e2 = e12[:32], e12[32:]
        a0 = g0.detach().double() * is0
        xh0f = is0[:, None] * (w0d @ F2 - mu0[:, None] * F1[None, :])
        dW0 = a0[:, None] * (B0 - e1[:, None] * F1[None, :] - e2[:, None] * xh0f)
        ctx.misc = None  # release the workspace
        return (None, dW0.to(w0.dtype), E2.to(g0.dtype), E1.to(b0.dtype), dW1.to(w1.dtype), D2.to(g1.dtype), D1.to(b1.dtype), None, None)


def fused_pfn_train(mod, points, batch):
    l0, l1 = mod.pfn_layers
    return FusedPFNTrain.apply(points, l0.linear.weight, l0.norm.weight, l0.norm.bias, l1.linear.weight, l1.norm.weight, l1.norm.bias, mod, batch)

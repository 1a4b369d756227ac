"""TEST INFRASTRUCTURE -- a rulebook restatement of the two spconv layers the reference's backbone is built from.

The reference's 2-D sparse ResNet (det3d/models/backbones/sparse_resnet.py:9-68, blocks in det3d/models/utils/sparse_conv.py:16-63) calls
`spconv.pytorch.SubMConv2d` and `spconv.pytorch.SparseConv2d`.  spconv is a third-party CUDA package that is absent from this image and
from /root/reference (`pip install spconv-cu116`, UNPINNED, docker/Dockerfile:18), so it cannot be run or compiled here: parity of the
backbone is **unpinned**.  What can be checked is that the package's masked-dense formulation (models.SparseConvBlock / SparseBasicBlock /
SparseResNet: dense convolution, zeros at inactive sites, active set carried as a mask) computes exactly what spconv's published
gather -> GEMM -> scatter semantics say, site by site:

  SubMConv2d(k, stride 1)           output sites = input sites;  out[p] = sum over kernel offsets o with p + o - k//2 ACTIVE of W[o] . in[p + o - k//2]
  SparseConv2d(k, stride s, pad)    output sites = every q in the output grid whose window q*s - pad + [0,k)^2 holds an active input site;
                                    out[q] = sum over those active inputs of W[o] . in[q*s - pad + o];  grid = floor((H + 2 pad - k) / s) + 1
  BatchNorm1d over out.features     i.e. over the ACTIVE sites only (eval: running-statistics affine)

Only tests/ imports this file.  Plain numpy loops: small cases only."""
import numpy as np


def subm_conv2d(idx, feats, weight):
    """idx (N,3) int [b,y,x] unique; feats (N,Cin); weight (Cout,Cin,k,k) -> (N,Cout) at the same sites."""
    k = weight.shape[2]
    table = {tuple(p): i for i, p in enumerate(idx.tolist())}
    out = np.zeros((len(idx), weight.shape[0]), feats.dtype)
    for i, (b, y, x) in enumerate(idx.tolist()):
        for ky in range(k):
            for kx in range(k):
                j = table.get((b, y + ky - k // 2, x + kx - k // 2))
                if j is not None:
                    out[i] += weight[:, :, ky, kx] @ feats[j]
    return out


def sparse_conv2d(idx, feats, hw, weight, stride, pad):
    """-> (out_idx (M,3) sorted by (b,y,x), out_feats (M,Cout), (Ho,Wo))."""
    k = weight.shape[2]
    H, W = hw
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    acc = {}
    for (b, y, x), f in zip(idx.tolist(), feats):
        for ky in range(k):
            for kx in range(k):
                qy, ry = divmod(y + pad - ky, stride)
                qx, rx = divmod(x + pad - kx, stride)
                if ry == 0 and rx == 0 and 0 <= qy < Ho and 0 <= qx < Wo:
                    key = (b, qy, qx)
                    acc[key] = acc.get(key, 0) + weight[:, :, ky, kx] @ f
    keys = sorted(acc)
    return np.asarray(keys, np.int64).reshape(-1, 3), np.stack([acc[q] for q in keys]) if keys else np.zeros((0, weight.shape[0]), feats.dtype), (Ho, Wo)


def bn_eval(feats, mean, var, gamma, beta, eps):
    return (feats - mean) / np.sqrt(var + eps) * gamma + beta


def conv_block(idx, feats, hw, p, stride, subm):
    """sparse_conv.py:16-39: conv -> BatchNorm1d(features) -> ReLU.  p: dict(weight, mean, var, gamma, beta, eps)."""
    if stride == 1 and subm:
        out_idx, out, hw_out = idx, subm_conv2d(idx, feats, p["weight"]), hw
    else:
        out_idx, out, hw_out = sparse_conv2d(idx, feats, hw, p["weight"], stride, p["weight"].shape[2] // 2)
    return out_idx, np.maximum(bn_eval(out, p["mean"], p["var"], p["gamma"], p["beta"], p["eps"]), 0), hw_out


def basic_block(idx, feats, hw, p1, p2):
    """sparse_conv.py:42-63: block1 (SubM conv + BN + ReLU), SubM conv2 + BN, + identity, ReLU."""
    _, out, _ = conv_block(idx, feats, hw, p1, 1, True)
    out = bn_eval(subm_conv2d(idx, out, p2["weight"]), p2["mean"], p2["var"], p2["gamma"], p2["beta"], p2["eps"])
    return np.maximum(out + feats, 0)

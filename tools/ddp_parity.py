#!/usr/bin/env python3
"""2+ ranks: SyncBN conversion + data-parallel step through the real HIP reader (fused training passes, csrc/pfn_train.hip) and a
small masked-dense backbone; outputs, running statistics and the averaged gradients (all six PFN parameters included) must equal a
single-process run on the concatenated batch.  Mirrors tools/train.py:53-60 + trainer/trainer/trainer.py:94-108 of the reference.

  torchrun --nproc-per-node 2 tools/ddp_parity.py                  one rank per GPU, RCCL, DistributedDataParallel
  ... tools/ddp_parity.py --backend gloo --one-gpu                 every rank on cuda:0, gloo (device tensors staged through the host by
                                                                   dist_utils.all_reduce_sum), gradients averaged by hand: exercises
                                                                   the reader's statistic exchange with world > 1 on a 1-GPU box
Prints "DDP PARITY OK" on rank 0."""
import argparse
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pillarnext_amd import dist_utils, synth  # noqa: E402
from pillarnext_amd.models import SparseResNet, convert_sync_batchnorm  # noqa: E402
from pillarnext_amd.reader import PillarFeatureNet  # noqa: E402


class Net(torch.nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.reader = PillarFeatureNet(5, [64, 64], list(cfg["voxel_size"]), list(cfg["pc_range"]))
        self.backbone = SparseResNet([1, 1], [2, 2], [32, 32], 64, kernel_size=(3, 3), out_channels=32)

    def forward(self, pts, batch):
        ny, nx = (int(v) for v in self.reader.grid_size)
        occ = torch.empty((batch, ny, nx), dtype=torch.uint8, device=pts.device)
        canvas = self.reader.forward_dense(pts, batch, dtype=torch.float32, occupancy=occ)
        return self.backbone.forward_dense(canvas, occ.unsqueeze(1).float())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--one-gpu", action="store_true", help="all ranks share cuda:0; no DistributedDataParallel, gradients averaged by hand")
    a = ap.parse_args()
    rank, world, local = dist_utils.init(a.backend)
    assert world > 1, "run under torchrun / torch.distributed.run with >= 2 ranks"
    local = 0 if a.one_gpu else local
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cfg = synth.CONFIGS["C1"]
    B = 2
    torch.manual_seed(0)
    net = Net(cfg).to(dev).train()
    state = {k: v.clone() for k, v in net.state_dict().items()}
    if a.one_gpu:
        model = convert_sync_batchnorm(net)
    else:
        model = dist_utils.wrap_ddp(net, device_ids=[local], sync_batchnorm=True)
    assert net.reader.sync and net.reader._fused_supported(), "the reader must run its fused, synchronised training passes"
    clouds = [synth.make_batch("C1", B, "sweep", n=12_000 + 900 * r, frame0=r * B) for r in range(world)]  # unequal point counts per rank
    y = model(torch.from_numpy(clouds[rank]).to(dev), B)
    # per-rank loss = sum over the local batch / global element count: the global loss is the plain SUM of the rank losses; DDP
    # averages gradients (so scale by world there), the hand-rolled path sums them
    numel_global = y.numel() * world
    loss = y.square().sum() / numel_global
    (loss * (1 if a.one_gpu else world)).backward()
    if a.one_gpu:
        for p in net.parameters():
            dist_utils.all_reduce_sum(p.grad)
    grads = {k: p.grad.clone() for k, p in net.named_parameters()}
    stats = {k: v.clone() for k, v in net.state_dict().items() if "running" in k}
    ok = True
    if rank == 0:
        ref = Net(cfg).to(dev).train()
        ref.load_state_dict(state)
        allpts = []
        for r, c in enumerate(clouds):
            c = c.copy()
            c[:, 0] += r * B
            allpts.append(c)
        y2 = ref(torch.from_numpy(np.concatenate(allpts)).to(dev), B * world)
        (y2.square().sum() / y2.numel()).backward()

        def cmp(name, got, want, rtol, atol):
            nonlocal ok
            if not torch.allclose(got, want, rtol=rtol, atol=atol):
                ok = False
                print("MISMATCH", name, float((got - want).abs().max()), float(want.abs().max()))

        cmp("output", y, y2[:B], 1e-3, 1e-5)
        pfn = 0
        for k, p in ref.named_parameters():
            # norm-wise: the joint-batch reference runs its convolutions at another batch size, i.e. on other MIOpen solvers (Winograd / implicit
            # GEMM / direct differ at the 1e-3 level of a gradient's scale in fp32); a missing or wrong all-reduce is an O(1) error
            cmp("grad " + k, grads[k], p.grad, 5e-3, 1e-2 * float(p.grad.abs().max()) + 1e-7)
            pfn += k.startswith("reader.pfn_layers")
        assert pfn == 6, pfn  # W0, gamma0, beta0, W1, gamma1, beta1
        for k, v in ref.state_dict().items():
            if "running" in k:
                cmp(k, stats[k], v, 1e-4, 1e-6)
        print("DDP PARITY OK" if ok else "DDP PARITY FAILED")
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()

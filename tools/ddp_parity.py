#!/usr/bin/env python3
"""2+ ranks (torchrun, RCCL): SyncBN conversion + DDP through the real HIP reader and a small dense detector; the averaged gradients
must equal a single-process run on the concatenated batch.  Mirrors tools/train.py:53-60 + trainer/trainer/trainer.py:94-108 of the
reference.  Prints "DDP PARITY OK" on rank 0."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pillarnext_amd import dist_utils, synth  # noqa: E402
from pillarnext_amd.models import SparseResNet  # noqa: E402
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
    rank, world, local = dist_utils.init("nccl")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cfg = synth.CONFIGS["C1"]
    B = 2
    torch.manual_seed(0)
    net = Net(cfg).to(dev).train()
    state = {k: v.clone() for k, v in net.state_dict().items()}
    ddp = dist_utils.wrap_ddp(net, device_ids=[local], sync_batchnorm=True)
    clouds = [synth.make_batch("C1", B, "sweep", n=12_000, frame0=r * B) for r in range(world)]
    y = ddp(torch.from_numpy(clouds[rank]).to(dev), B)
    (y.square().mean()).backward()
    grads = {k: p.grad.clone() for k, p in ddp.module.named_parameters()}
    ok = True
    if rank == 0:
        ref = Net(cfg).to(dev).train()
        ref.load_state_dict(state)
        import numpy as np

        allpts = []
        for r, c in enumerate(clouds):
            c = c.copy()
            c[:, 0] += r * B
            allpts.append(c)
        y2 = ref(torch.from_numpy(np.concatenate(allpts)).to(dev), B * world)
        y2.square().mean().backward()                         # mean over the global batch == average of the per-rank means
        for k, p in ref.named_parameters():
            if not torch.allclose(grads[k], p.grad, rtol=5e-3, atol=1e-5):
                ok = False
                print("MISMATCH", k, float((grads[k] - p.grad).abs().max()), float(p.grad.abs().max()))
        print("DDP PARITY OK" if ok else "DDP PARITY FAILED")
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()

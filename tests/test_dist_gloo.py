"""CPU, world_size 2, gloo: the N>1 code path (frame sharding, max-over-ranks timing, DDP gradient all-reduce over the
dense modules, detection gather).  The HIP reader itself needs a GPU and is covered by the -m gpu tests."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from pillarnext_amd import dist_utils
    from pillarnext_amd.models import ASPPNeck, SepHead

    r, w, _ = dist_utils.init("gloo")
    assert (r, w) == (rank, world)
    # 1. frame sharding: disjoint cover
    mine = dist_utils.shard_frames(7, rank, world)
    allf = [None] * world
    dist.all_gather_object(allf, mine)
    assert sorted(set(sum(allf, []))) == list(range(7)) and len(set(map(len, allf))) == 1
    # 2. timing aggregation
    assert dist_utils.max_over_ranks(1.0 + rank) == float(world)
    # 3. DDP over dense modules: averaged grads == single-process grads on the concatenated batch
    torch.manual_seed(0)
    net = torch.nn.Sequential(ASPPNeck(8), SepHead(8, {"hm": (2, 2), "reg": (2, 2)}, stride=2, head_conv=8, final_kernel=3))
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.eval()  # frozen statistics: keeps the comparison exact without SyncBN on CPU
    ref = [p.detach().clone() for p in net.parameters()]
    ddp = dist_utils.wrap_ddp(net, sync_batchnorm=False)
    g = torch.Generator().manual_seed(123)
    x_all = torch.randn((2 * world, 8, 12, 12), generator=g)
    x = x_all[rank * 2: rank * 2 + 2].clone().requires_grad_(True)
    out = ddp(x)
    (out["hm"].square().mean() + out["reg"].abs().mean()).backward()
    grads = [p.grad.clone() for p in net.parameters()]
    if rank == 0:
        net2 = torch.nn.Sequential(ASPPNeck(8), SepHead(8, {"hm": (2, 2), "reg": (2, 2)}, stride=2, head_conv=8, final_kernel=3))
        with torch.no_grad():
            for p2, p in zip(net2.parameters(), ref):
                p2.copy_(p)
        for m in net2.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.eval()
        loss = 0
        for k in range(world):
            o = net2(x_all[k * 2: k * 2 + 2].clone().requires_grad_(True))
            loss = loss + (o["hm"].square().mean() + o["reg"].abs().mean()) / world
        loss.backward()
        for a, p2 in zip(grads, net2.parameters()):
            torch.testing.assert_close(a, p2.grad, rtol=1e-4, atol=1e-6)
    # 4. detection gather
    det = dist_utils.gather_detections({f"r{rank}": {"scores": torch.ones(rank + 1)}})
    assert sorted(det) == [f"r{k}" for k in range(world)]
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, "ok"))


def test_world_size_2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    got = sorted(q.get(timeout=5) for _ in range(2))
    assert got == [(0, "ok"), (1, "ok")]


def _worker_syncbn(rank, world, port, q):
    """Masked SyncBN through the backbone: wrap_ddp(sync_batchnorm=True) must keep the (x, mask) signature of
    MaskedBatchNorm and take the statistics over the ACTIVE sites of the GLOBAL batch: outputs, running statistics and parameter
    gradients of 2 ranks x 1 sample == one process on the 2-sample batch."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from pillarnext_amd import dist_utils
    from pillarnext_amd.models import MaskedBatchNorm, SparseResNet

    dist_utils.init("gloo")
    torch.manual_seed(0)
    net = SparseResNet([1, 1], [1, 2], [8, 16], 8, kernel_size=(3, 3), out_channels=16).train()
    ref_state = {k: v.clone() for k, v in net.state_dict().items()}
    g = torch.Generator().manual_seed(5)
    x_all = torch.randn((world, 8, 16, 16), generator=g)
    m_all = (torch.rand((world, 1, 16, 16), generator=g) < (0.3 + 0.3 * torch.arange(world).view(-1, 1, 1, 1))).float()  # unequal site counts
    x_all = x_all * m_all
    from pillarnext_amd.models import convert_sync_batchnorm

    convert_sync_batchnorm(net)  # what wrap_ddp(sync_batchnorm=True) does before the DDP wrap
    assert all(m.sync for m in net.modules() if isinstance(m, MaskedBatchNorm))
    # DDP's forward hooks need the module's forward(): call forward_dense through a thin wrapper module
    class Wrap(torch.nn.Module):
        def __init__(self, bb):
            super().__init__()
            self.bb = bb

        def forward(self, x, m):
            return self.bb.forward_dense(x, m)

    w = torch.nn.parallel.DistributedDataParallel(Wrap(net))
    y = w(x_all[rank: rank + 1], m_all[rank: rank + 1])
    # DDP averages gradients over ranks: scale so that the sum of the two local losses equals the single-process loss
    (y.square().sum() * world).backward()
    grads = {k: p.grad.clone() for k, p in net.named_parameters()}
    stats = {k: v.clone() for k, v in net.state_dict().items() if "running" in k}
    if rank == 0:
        net2 = SparseResNet([1, 1], [1, 2], [8, 16], 8, kernel_size=(3, 3), out_channels=16).train()
        net2.load_state_dict(ref_state)
        y2 = net2.forward_dense(x_all, m_all)
        y2.square().sum().backward()
        torch.testing.assert_close(y, y2[:1], rtol=1e-4, atol=1e-5)
        for k, p in net2.named_parameters():
            torch.testing.assert_close(grads[k], p.grad, rtol=2e-3, atol=2e-4, msg=lambda s, k=k: f"{k}: {s}")  # fp32 summation order (local sums + all-reduce vs one pass)
        for k, v in net2.state_dict().items():
            if "running" in k:
                torch.testing.assert_close(stats[k], v, rtol=1e-5, atol=1e-6)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, "ok"))


def test_masked_syncbn_world_size_2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_syncbn, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert sorted(q.get(timeout=5) for _ in range(2)) == [(0, "ok"), (1, "ok")]

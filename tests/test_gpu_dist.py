"""GPU, world_size 2 on ONE device: the fused training reader's SyncBatchNorm statistic exchange (pfn_train.py, forward and backward)
and the masked SyncBN of the backbone, with gloo as the transport (device tensors staged through the host) -- reference
tools/train.py:55-60, pillar_encoder.py:33,38.  The RCCL form of the same script needs >= 2 GPUs (tests/test_gpu_configs.py)."""
import os
import socket
import subprocess
import sys

import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(args, timeout=420):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tools", "ddp_parity.py")] + args
    return subprocess.run(cmd, env=env, stdin=subprocess.DEVNULL, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout, text=True)


def test_fused_reader_syncbn_world2_on_one_gpu():
    torch.cuda.empty_cache()  # the two ranks share this process's GPU: hand back what earlier tests left in the caching allocator
    r = _run(["--backend", "gloo", "--one-gpu"])
    assert r.returncode == 0 and "DDP PARITY OK" in r.stdout, r.stdout[-4000:]

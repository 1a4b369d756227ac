#!/bin/bash
# Is the convolution power / clock bound?  The same launches on random-normal, post-ReLU and all-zero activations, and a library GEMM as the practical ceiling.
OUT=gpurun_out/${1:-r06d}; mkdir -p $OUT
export TMPDIR=/tmp PNX_CONV_PC=${2:-0}
{
for d in randn relu zero; do
  echo "## data=$d"
  timeout 300 python tools/bench_conv.py --cin 256 --cout 256 --hw 360 --batch 8 --data $d 2>&1 | grep -v amdgpu.ids | tail -1
  timeout 300 python tools/bench_conv.py --cin 64 --cout 64 --batch 2 --data $d 2>&1 | grep -v amdgpu.ids | tail -1
  timeout 300 python tools/bench_conv.py --batch 12 --tiles --cin 64 --cout 64 --lidar 0 --dilate --res --data $d 2>&1 | grep -v amdgpu.ids | tail -1
  timeout 300 python tools/bench_conv.py --batch 12 --tiles --cin 256 --cout 256 --lidar 2 --dilate --res --data $d 2>&1 | grep -v amdgpu.ids | tail -1
done
python - <<'PY'
import torch, time
for n in (4096, 8192):
    for kind in ("randn", "zero"):
        a = (torch.randn(n, n, device="cuda") if kind == "randn" else torch.zeros(n, n, device="cuda")).bfloat16()
        b = (torch.randn(n, n, device="cuda") if kind == "randn" else torch.zeros(n, n, device="cuda")).bfloat16()
        for _ in range(3): a @ b
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): a @ b
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f"torch.matmul bf16 {n}^3 {kind}: {ms*1e3:.0f} us  {2*n**3/ms/1e9:.0f} TFLOP/s")
PY
} > $OUT/power.txt 2>&1
cat $OUT/power.txt

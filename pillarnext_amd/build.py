"""Builds libpnx_hip.so (the C-ABI library of include/pnx.h) in-tree with hipcc for gfx950."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.environ.get("PNX_LIB_OUT") or os.path.join(HERE, "libpnx_hip.so")  # PNX_LIB_OUT: where an instrumented build goes
SOURCES = ["reader.hip", "pfn_v3.hip", "chunk_sort.hip", "pfn_spans.hip", "pfn_train.hip", "scatter.hip", "group.hip", "merge.hip", "iou3d.hip", "dense_ops.hip", "masked_bn.hip", "conv3x3.hip", "conv_wgrad.hip", "head_train.hip", "decode.hip", "center_loss.hip", "enqueue.hip", "iou3d_host.cpp", "capi.cpp"]
HEADERS = ["pnx_common.h", "pnx_scan.h", "pnx_detmath.h", "pnx_fill.h", "pnx_dppscan.h", "pfn_common.h", "reader_bins.h", "spans.h", "iou3d_geom.h", "conv_pc.h", "conv_dgrad_s2.h", os.path.join("..", "..", "include", "pnx.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wno-unused-function",
         "-Wno-unused-result"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force=False, verbose=False):
    if not force and not stale():
        return LIB
    objs = []
    procs = []
    # conv3x3.hip is compiled twice: bf16 (as is) and IEEE half (-DPNX_CONV_F16 -> the pnx_*_f16 entry points)
    for src, variant in [(s, "") for s in SOURCES] + [("conv3x3.hip", "_f16")]:
        obj = os.path.join(CSRC, os.path.splitext(src)[0] + variant + (os.environ.get("PNX_OBJ_SUFFIX") or "") + ".o")
        extra = []
        if src in ("pfn_v3.hip", "pfn_spans.hip"):  # fmaxf without canonicalising v_max pairs (-inf still honoured); MFMA accumulators in VGPRs (no v_accvgpr traffic)
            extra = ["-fno-honor-nans", "-mllvm", "-amdgpu-mfma-vgpr-form=1"] + (["-DPNX_PFN_TIMERS"] if os.environ.get("PNX_PFN_TIMERS") else []) + (["-DPNX_BINS_TIMERS"] if os.environ.get("PNX_BINS_TIMERS") else [])
            if src == "pfn_spans.hip" and os.environ.get("PNX_SPAN_NUM_VGPR"):
                extra.append("-DPNX_SPAN_NUM_VGPR=" + os.environ["PNX_SPAN_NUM_VGPR"])
        elif src == "chunk_sort.hip":
            extra = ["-DPNX_BINS_TIMERS"] if os.environ.get("PNX_BINS_TIMERS") else []
        elif src == "conv3x3.hip":
            extra = ["-fno-honor-nans"] + (os.environ.get("PNX_CONV_FLAGS") or "").split() + (["-DPNX_CONV_TIMERS"] if os.environ.get("PNX_CONV_TIMERS") else []) + (["-DPNX_CONV_F16"] if variant else []) + (os.environ.get("PNX_CONV_DEFS") or "").split()
        cmd = [_hipcc()] + FLAGS + extra + (["-x", "hip"] if src.endswith(".hip") else []) + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src + variant, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError(f"hipcc failed on {src}")
        if verbose and out:
            print(out.decode())
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

"""In-tree build of libairv2x_hip.so with hipcc for gfx950 (no JIT cache: the .so travels
with the repo snapshot to the GPU box)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libairv2x_hip.so")
SOURCES = ["capi.hip", "conv_igemm.hip", "conv_wino_x3.hip", "conv_wino4_x3.hip", "conv_x3p.hip", "pillar.hip", "where2comm.hip", "where2comm_attn.hip", "postproc.hip", "voxelize.hip", "transformer.hip", "v2xvit.hip", "linear_bf16.hip", "when2com.hip", "v2vnet.hip", "lss.hip", "camera.hip", "labels.hip", "conv_backward.hip", "loss.hip", "train.hip", "train_fusion.hip", "train_v2xvit.hip", "train_when2com.hip", "train_camera.hip"]


# per-source flags.  The split-3 Winograd kernels keep their channel-pair arithmetic scalar on purpose (a packed fp32 instruction beside
# MFMAs costs more than the two scalar ones it replaces): neither the SLP vectoriser nor VectorCombine may re-pack it.
_SCALAR_F32 = ["-fno-slp-vectorize", "-mllvm", "-disable-vector-combine"]
EXTRA_FLAGS = {"conv_wino_x3.hip": _SCALAR_F32, "conv_wino4_x3.hip": _SCALAR_F32}


class HipccMissing(RuntimeError):
    pass


def hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise HipccMissing("hipcc not found: libairv2x_hip.so cannot be built (no CPU fallback exists)")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(ROOT, "include", "airv2x_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    cc = hipcc()
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=on",
             "-I", os.path.join(ROOT, "include"), "-I", CSRC]
    objs = []
    procs = []
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s) + ".o")
        objs.append(o)
        # headers / .inc files are included by several sources: any of them newer than the object -> recompile
        hdr_t = max([os.path.getmtime(os.path.join(CSRC, f)) for f in os.listdir(CSRC) if not f.endswith(".hip")]
                    + [os.path.getmtime(os.path.join(ROOT, "include", "airv2x_hip.h"))])
        if not force and os.path.exists(o) and os.path.getmtime(o) > max(os.path.getmtime(s), hdr_t):
            continue
        cmd = [cc, *flags, *EXTRA_FLAGS.get(os.path.basename(s), []), "-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd))
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {s}:\n{out}")
        if verbose and out.strip():
            print(out)
    cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

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
SOURCES = ["capi.hip", "conv_igemm.hip", "conv_wino_x3.hip", "conv_wino4_x3.hip", "conv_x3p.hip", "pillar.hip", "where2comm.hip", "where2comm_attn.hip", "postproc.hip", "voxelize.hip", "transformer.hip", "v2xvit.hip", "linear_bf16.hip", "when2com.hip", "v2vnet.hip", "lss.hip", "camera.hip", "labels.hip", "conv_backward.hip", "loss.hip", "train.hip", "train_fusion.hip", "train_v2xvit.hip", "train_when2com.hip", "train_camera.hip", "roiaware.hip", "sparse_conv.hip"]


# per-source flags.  The split-3 Winograd kernels keep their channel-pair arithmetic scalar on purpose (a packed fp32 instruction beside
# MFMAs costs more than the two scalar ones it replaces): neither the SLP vectoriser nor VectorCombine may re-pack it.
_SCALAR_F32 = ["-fno-slp-vectorize", "-mllvm", "-disable-vector-combine"]
# gfx950 hazard found in round 4 (tools/micro/coreside.py + guard.hip): a packed-fp32 VOP3P instruction whose OP_SEL bit of
# the SECOND (or third) source is set -- v_pk_add_f32 / v_pk_mul_f32 ... op_sel:[0,1], v_pk_fma_f32 ... op_sel:[0,1,0] -- returns wrong
# lanes while ANOTHER wave on the same SIMD issues v_mfma_f32_32x32x16_bf16 (the split-3 and bf16 kernels of another in-flight frame).
# op_sel_hi-only (broadcast), neg_* and first-source op_sel forms are not affected.  The compiler forms the bad pattern when it folds a
# lane swap into a packed op; the files where it did are compiled without packed-fp32 instructions (they are HBM-bound kernels), and
# lint_isa() below rejects the pattern in EVERY kernel of the library at build time.
_NO_PACKED_F32 = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
# (Timing-only ablation macros -- AV2X_X3P_NOSPLIT / NOLDS / NOEPI, AV2X_WX3_ABLATE, AV2X_W4X3_ABLATE: wrong results by design -- are never part
# of this build: the harnesses under tools/micro/ compile their OWN binaries / libraries from the same sources.)
EXTRA_FLAGS = {"conv_wino_x3.hip": _SCALAR_F32, "conv_wino4_x3.hip": _SCALAR_F32,
               **{f: _NO_PACKED_F32 for f in ("transformer.hip", "postproc.hip", "train.hip", "loss.hip", "pillar.hip", "lss.hip", "camera.hip", "train_fusion.hip", "train_v2xvit.hip",
                                              "voxelize.hip")}}


class HipccMissing(RuntimeError):
    pass


def hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise HipccMissing("hipcc not found: libairv2x_hip.so cannot be built (no CPU fallback exists)")


BASE_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=on"]


def _flag_sig(src):
    import hashlib
    return hashlib.sha256(" ".join([*BASE_FLAGS, *EXTRA_FLAGS.get(os.path.basename(src), [])]).encode()).hexdigest()


def _object_current(src, objdir):
    """The object of ``src`` was made with today's flags and its device-ISA listing (what lint_isa scans) is there."""
    o = os.path.join(objdir, os.path.basename(src) + ".o")
    isa = os.path.join(objdir, os.path.splitext(os.path.basename(src))[0] + "-hip-amdgcn-amd-amdhsa-gfx950.s")
    sig_path = o + ".flags"
    return os.path.exists(o) and os.path.exists(isa) and os.path.exists(sig_path) and open(sig_path).read().strip() == _flag_sig(src)


def needs_build():
    if not os.path.exists(LIB):
        return True
    objdir = os.path.join(HERE, "build")
    if os.path.isdir(objdir) and not all(_object_current(s, objdir) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))):
        return True     # (no object directory at all = a shipped .so on a box that only runs it: nothing to compare)
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
    flags = [*BASE_FLAGS, "-I", os.path.join(ROOT, "include"), "-I", CSRC]
    objs = []
    procs = []
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s) + ".o")
        objs.append(o)
        # an object is only as good as the command line that made it: the flags (global + per file) are hashed into <object>.flags and a
        # mismatch -- or a missing device-ISA listing, which lint_isa() needs -- recompiles, whatever the time stamps say
        if not _object_current(s, objdir) and os.path.exists(o):
            os.remove(o)
        open(o + ".flags", "w").write(_flag_sig(s))
        # headers / .inc files are included by several sources: any of them newer than the object -> recompile
        hdr_t = max([os.path.getmtime(os.path.join(CSRC, f)) for f in os.listdir(CSRC) if not f.endswith(".hip")]
                    + [os.path.getmtime(os.path.join(ROOT, "include", "airv2x_hip.h"))])
        if not force and os.path.exists(o) and os.path.getmtime(o) > max(os.path.getmtime(s), hdr_t):
            continue
        cmd = [cc, *flags, *EXTRA_FLAGS.get(os.path.basename(s), []), "-save-temps=obj", "-c", s, "-o", o]    # keeps the device ISA for lint_isa
        if verbose:
            print(" ".join(cmd))
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {s}:\n{out}")
        if verbose and out.strip():
            print(out)
    for f in os.listdir(objdir):        # -save-temps leaves ~10 files per source; only the device ISA is kept
        if f.endswith((".bc", ".hipi", ".out", ".hipfb", ".resolution.txt")) or f.endswith("-host-x86_64-unknown-linux-gnu.s") \
                or f.endswith("-hip-amdgcn-amd-amdhsa-gfx950.o"):
            os.remove(os.path.join(objdir, f))
    missing = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))
               and not os.path.exists(os.path.join(objdir, os.path.splitext(s)[0] + "-hip-amdgcn-amd-amdhsa-gfx950.s"))]
    if missing:
        raise RuntimeError("no device ISA listing (-save-temps) for " + ", ".join(missing) + ": the packed-fp32 OP_SEL lint cannot vouch for them")
    bad = lint_isa(objdir)
    if bad:
        raise RuntimeError("packed-fp32 instructions with a second / third source OP_SEL bit (wrong results next to bf16-MFMA waves on gfx950, "
                           "see build.py) in:\n" + "\n".join(f"  {f}: {k}: {ins}" for f, k, ins in bad[:20]))
    cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    return LIB


def lint_isa(objdir=None):
    """[(file, kernel, instruction)] for every packed-fp32 instruction of the device ISA (the *-gfx950.s files -save-temps leaves beside the
    objects) whose op_sel sets the bit of the second or third source."""
    import glob
    import re
    objdir = objdir or os.path.join(HERE, "build")
    pat = re.compile(r"^\s+(v_pk_(?:add|mul|fma)_f32)\b.*\bop_sel:\[(\d(?:,\d)+)\]")
    bad = []
    for f in sorted(glob.glob(os.path.join(objdir, "*-hip-amdgcn-amd-amdhsa-gfx950.s"))):
        if not any(os.path.basename(f).startswith(os.path.splitext(s)[0] + "-hip-") for s in SOURCES):
            continue            # a stale file of a source that left the build
        kernel = None
        for line in open(f):
            m = re.match(r"^(_Z\w+):", line)
            if m:
                kernel = m.group(1)
            m = pat.match(line)
            if m and "1" in m.group(2).split(",")[1:]:
                bad.append((os.path.basename(f), kernel, line.strip()))
    return bad


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

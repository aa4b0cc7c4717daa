"""Build recipe for `oracle/_ref/` (TEST INFRASTRUCTURE, build container only): compiles the reference's OWN C++ rotated-IoU
(/root/reference/opencood/pcdet_utils/iou3d_nms/src/iou3d_cpu.cpp: box_overlap :128-229, iou_bev :231-238,
boxes_iou_bev_cpu :241-262) from the sources where they lie, with g++ against the torch headers of this image and the
CUDA toolkit headers that ship inside the triton wheel (the file includes <cuda.h> / <cuda_runtime_api.h> only for
the `__device__` annotation macros; nothing of CUDA is linked).  Nothing is copied into the repo; the only output is
oracle/_ref/iou3d_cpu_ref.so (git-ignored).  Used by tools/gen_golden.py (group `iou_pin`) to store the reference's
IoUs for random rotated box pairs under tests/golden/, and by nobody else.
"""
from __future__ import annotations

import os
import subprocess
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SRC = "/root/reference/opencood/pcdet_utils/iou3d_nms/src"
OUT = os.path.join(HERE, "_ref", "iou3d_cpu_ref.so")


def available():
    return os.path.exists(os.path.join(REF_SRC, "iou3d_cpu.cpp"))


def build(force=False):
    """-> path of the built library, or None when the reference tree is absent (the GPU box)."""
    if not available():
        return OUT if os.path.exists(OUT) else None
    src = os.path.join(REF_SRC, "iou3d_cpu.cpp")
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) > os.path.getmtime(src):
        return OUT
    import torch
    import triton
    tinc = os.path.join(os.path.dirname(torch.__file__), "include")
    cuda_inc = os.path.join(os.path.dirname(triton.__file__), "backends", "nvidia", "include")
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    cmd = ["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-w", src, "-o", OUT,
           "-I", REF_SRC, "-I", tinc, "-I", os.path.join(tinc, "torch", "csrc", "api", "include"), "-I", cuda_inc,
           "-I", sysconfig.get_paths()["include"], "-D_GLIBCXX_USE_CXX11_ABI=" + str(int(torch._C._GLIBCXX_USE_CXX11_ABI)),
           "-L", tlib, "-Wl,-rpath," + tlib, "-ltorch", "-ltorch_cpu", "-lc10", "-ltorch_python"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("reference iou3d_cpu.cpp did not build:\n" + r.stdout[-4000:])
    return OUT


def boxes_iou_bev(boxes_a, boxes_b):
    """The reference's boxes_iou_bev_cpu(boxes_a (N,7), boxes_b (M,7)) -> (N,M) float32 [x,y,z,dx,dy,dz,heading]."""
    import ctypes

    import torch
    lib = ctypes.CDLL(build())
    # C++ symbol: int boxes_iou_bev_cpu(at::Tensor, at::Tensor, at::Tensor) -- called through its mangled name with
    # at::Tensor passed by value = a pointer to an intrusive_ptr holder on this ABI; go through torch.ops-free ctypes by
    # handing over the TensorImpl pointers torch exposes
    fn = lib._Z17boxes_iou_bev_cpuN2at6TensorES0_S0_
    fn.restype = ctypes.c_int
    a = boxes_a.contiguous().float()
    b = boxes_b.contiguous().float()
    out = torch.zeros(a.shape[0], b.shape[0])
    # at::Tensor is a single pointer (c10::intrusive_ptr<TensorImpl>); non-trivially-copyable classes are passed by
    # invisible reference on the Itanium ABI: the callee receives the ADDRESS of a temporary holding that pointer
    hold = [ctypes.c_void_p(t._cdata) for t in (a, b, out)]
    fn(ctypes.byref(hold[0]), ctypes.byref(hold[1]), ctypes.byref(hold[2]))
    return out


PYX_SRC = "/root/reference/opencood/utils/box_overlaps.pyx"
PYX_OUT_DIR = os.path.join(HERE, "_ref")


def build_box_overlaps(force=False):
    """The reference's own utils/box_overlaps.pyx (bbox_overlaps :17-57: the IoU behind generate_label_airv2x) compiled
    where it lies with Cython + gcc; generated C and the module go to oracle/_ref/ only.  -> importable module path or None."""
    import glob
    import sys
    have = glob.glob(os.path.join(PYX_OUT_DIR, "box_overlaps*.so"))
    if not os.path.exists(PYX_SRC):
        return have[0] if have else None
    if have and not force and os.path.getmtime(have[0]) > os.path.getmtime(PYX_SRC):
        return have[0]
    import numpy as np
    os.makedirs(PYX_OUT_DIR, exist_ok=True)
    c_file = os.path.join(PYX_OUT_DIR, "box_overlaps.c")
    r = subprocess.run([sys.executable, "-m", "cython", "-3", PYX_SRC, "-o", c_file], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("cython failed on the reference's box_overlaps.pyx:\n" + r.stdout[-3000:])
    out = os.path.join(PYX_OUT_DIR, "box_overlaps" + sysconfig.get_config_var("EXT_SUFFIX"))
    r = subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-w", c_file, "-o", out, "-I", sysconfig.get_paths()["include"], "-I", np.get_include()],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("gcc failed on the generated box_overlaps.c:\n" + r.stdout[-3000:])
    os.unlink(c_file)      # the generated C embeds the .pyx text as comments: only the compiled module is kept, in this container only (oracle/_ref is git- and gpurun-ignored)
    return out


def import_box_overlaps():
    """The compiled reference module (for tools/gen_golden.py only)."""
    import importlib.util
    path = build_box_overlaps()
    spec = importlib.util.spec_from_file_location("box_overlaps", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    print(build(force=True))
    print(build_box_overlaps(force=True))

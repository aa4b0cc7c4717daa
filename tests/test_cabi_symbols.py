"""CPU: libairv2x_hip.so builds/loads without a GPU and exports every symbol that
include/airv2x_hip.h declares; the ctypes table covers exactly the same set.  No compute calls."""
import ctypes
import os
import re

from airv2x_perception_amd import _lib, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "airv2x_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(av2x_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_are_exported_and_bound():
    path = build.build()  # hipcc cross-compiles gfx950 without a GPU; no-op when up to date
    assert os.path.exists(path)
    names = _declared()
    assert len(names) >= 10
    lib = ctypes.CDLL(path)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in the header but not exported"
    assert sorted(_lib.SIGNATURES) == names, "ctypes table and header disagree"


def test_version_and_error_string_without_gpu():
    lib = _lib.load()
    assert lib.av2x_version() == 1
    # argument validation happens before any HIP call: a bad descriptor fails cleanly on a CPU-only host
    d = _lib.ConvDesc()
    d.cin = 30
    rc = lib.av2x_conv2d(ctypes.byref(d), 1, 1, 0, 1, 1, None)
    assert rc != 0
    assert b"cin" in lib.av2x_last_error()
    rc = lib.av2x_pixel_attn_fuse((ctypes.c_void_p * 1)(1), 99, 10, 64, 1, None)
    assert rc != 0 and b"n_agents" in lib.av2x_last_error()


def test_conv_desc_layout_matches_header():
    txt = open(os.path.join(ROOT, "include", "airv2x_hip.h")).read()
    body = re.search(r"typedef struct av2x_conv_desc \{(.*?)\} av2x_conv_desc;", txt, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if decl.startswith("int32_t"):
            fields += [f.strip() for f in decl[len("int32_t"):].split(",")]
    assert fields == [f for f, _ in _lib.ConvDesc._fields_]
    assert ctypes.sizeof(_lib.ConvDesc) == 4 * len(fields)

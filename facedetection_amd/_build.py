"""Builds the HIP library in-tree: facedetection_amd/libfacedet.so (gfx950 only).

    python -m facedetection_amd._build            # incremental
    python -m facedetection_amd._build --force

hipcc cross-compiles without a GPU; the resulting .so travels to the GPU box.
"""
import hashlib
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
OBJ = os.path.join(PKG, "csrc", "build")
LIB = os.path.join(PKG, "libfacedet.so")

# (source, extra flags).  The box kernels must round like NumPy: no fused multiply-add.
SOURCES = [
    ("fd_core.hip", []),
    ("fd_boxes.hip", ["-ffp-contract=off"]),
    ("fd_image.hip", ["-ffp-contract=off"]),
    ("fd_conv.hip", []),
    ("fd_conv_split.hip", []),
    ("fd_conv_fused.hip", []),
    ("fd_conv_panel.hip", []),
    ("fd_conv_tailw.hip", []),
    ("fd_winograd.hip", []),
    ("fd_detector.hip", []),
]
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def source_id():
    """sha256 (first 16 hex digits) over the library's sources: csrc/*.hip, csrc/*.h and
    include/facedet.h, in name order.  Compiled into the library (fd_build_id) so that a
    measurement file can name the kernel set it was taken on."""
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h")))
    files.append(os.path.join(os.path.dirname(PKG), "include", "facedet.h"))
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def _newer(a, b):
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(PKG), "include", "facedet.h"))
    objs, rebuilt = [], False
    # the id of this source set, as a generated header that only fd_core.hip includes
    bid = source_id()
    id_header = os.path.join(OBJ, "fd_build_id.h")
    id_text = '#define FD_BUILD_ID "%s"\n' % bid
    if not os.path.exists(id_header) or open(id_header).read() != id_text:
        with open(id_header, "w") as f:
            f.write(id_text)
    for src, extra in SOURCES:
        s = os.path.join(CSRC, src)
        if not os.path.exists(s):
            continue
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        stale_id = src == "fd_core.hip" and _newer(id_header, o)
        if force or stale_id or _newer(s, o) or any(_newer(h, o) for h in headers):
            cmd = [hipcc] + COMMON + extra + ["-I", OBJ, "-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            rebuilt = True
        objs.append(o)
    if rebuilt or not os.path.exists(LIB):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        # the offload bundler leaves its temporaries next to the output
        for f in os.listdir(PKG):
            if f.startswith("libfacedet.so."):
                os.remove(os.path.join(PKG, f))
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)

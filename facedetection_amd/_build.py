"""Builds the HIP library in-tree: facedetection_amd/libfacedet.so (gfx950 only).

    python -m facedetection_amd._build            # incremental
    python -m facedetection_amd._build --force

hipcc cross-compiles without a GPU; the resulting .so travels to the GPU box.
"""
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
    ("fd_detector.hip", []),
]
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _newer(a, b):
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(PKG), "include", "facedet.h"))
    objs, rebuilt = [], False
    for src, extra in SOURCES:
        s = os.path.join(CSRC, src)
        if not os.path.exists(s):
            continue
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        if force or _newer(s, o) or any(_newer(h, o) for h in headers):
            cmd = [hipcc] + COMMON + extra + ["-c", s, "-o", o]
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
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)

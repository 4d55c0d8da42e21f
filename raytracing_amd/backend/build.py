"""Builds ``raytracing_amd/_rt_hip.so`` with hipcc for gfx950 (cross-compiles without a GPU).

    python -m raytracing_amd.backend.build [--force]
"""
from __future__ import annotations

import os
import subprocess
import sys

PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(PKG, "csrc")
OUT = os.path.join(PKG, "_rt_hip.so")
SOURCES = ["rt_api.hip"]
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith(".h")) + [os.path.join("..", "..", "include", "raytracing_amd.h")]
# -ffp-contract=off: the double-precision exact code must round like upstream's Python floats (no
# fused multiply-add); the float32 kernels spell out every fma explicitly.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-shared", "-pthread", "-Wno-pass-failed"]


def needs_build() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, *FLAGS, "-o", OUT, *[os.path.join(CSRC, s) for s in SOURCES]]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=CSRC)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)

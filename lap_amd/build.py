"""Builds liblap_hip.so (the C-ABI kernel library, include/lap_hip.h) for gfx950 with hipcc.

In-tree build: the .so lands next to this file so that it travels to the GPU box with
the repository snapshot.  hipcc cross-compiles without a GPU.
"""
from __future__ import annotations

import hashlib
import os
import pathlib
import subprocess
import sys

ROOT = pathlib.Path(__file__).resolve().parent
CSRC = ROOT / "csrc"
LIB = ROOT / "liblap_hip.so"
SOURCES = ["gemm.hip", "gemm_fp8.hip", "norm.hip", "elementwise.hip", "attention.hip", "loss_optim.hip", "serve_fused.hip", "serve_skinny.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]  # no fast-math: parity with the f32 reference ops
if os.environ.get("LAP_GEMM_EXPERIMENTAL") == "1":   # also compile the non-production GEMM tile probes (1, 3, 4, 7, 8, 9)
    FLAGS.append("-DLAP_GEMM_EXPERIMENTAL")


def _digest() -> str:
    h = hashlib.sha256()
    for f in sorted(list(CSRC.glob("*.hip")) + list(CSRC.glob("*.hpp")) + [ROOT.parent / "include" / "lap_hip.h"]):
        h.update(f.name.encode())
        h.update(f.read_bytes())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True) -> pathlib.Path:
    stamp = ROOT / ".liblap_hip.digest"
    digest = _digest()
    if not force and LIB.exists() and stamp.exists() and stamp.read_text() == digest:
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = ROOT / "build"
    objdir.mkdir(exist_ok=True)
    procs = []
    objs = []
    for src in SOURCES:
        obj = objdir / (src + ".o")
        objs.append(str(obj))
        cmd = [hipcc, *FLAGS, "-c", str(CSRC / src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{out}")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", str(LIB)]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    stamp.write_text(digest)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)

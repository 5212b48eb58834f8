"""In-tree build of libisbfsar_hip.so (hipcc, gfx950 only).

    python -m isbfsar_amd.build [--force]

hipcc cross-compiles without a GPU; the resulting .so is git-ignored but travels with the
working tree to the GPU box. Objects are rebuilt when a source or header is newer.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
LIB = os.path.join(CSRC, "libisbfsar_hip.so")
OBJ = os.path.join(CSRC, "build")

SOURCES = [
    "isb_common.cpp",
    "gemm_f32.hip",
    "ar_kernels.hip",
    "ar_api.cpp",
]
# optional units appear as they are written
for _extra in ("hpe_kernels.hip", "conv_kernels.hip", "conv_ws.hip", "hpe_api.cpp", "det_kernels.hip", "det_api.cpp"):
    if os.path.exists(os.path.join(CSRC, _extra)):
        SOURCES.append(_extra)

FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wall", "-Wno-unused-function",
         "-Wno-unused-result", "-ffp-contract=on"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (needed to build libisbfsar_hip.so)")


def _deps_mtime() -> float:
    m = 0.0
    for d in (CSRC, os.path.join(ROOT, "include")):
        for f in os.listdir(d):
            if f.endswith((".h", ".hpp")):
                m = max(m, os.path.getmtime(os.path.join(d, f)))
    return m


def build(force: bool = False, verbose: bool = True) -> str:
    hipcc = _hipcc()
    os.makedirs(OBJ, exist_ok=True)
    hdr_m = _deps_mtime()
    jobs = []
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".", "_") + ".o")
        objs.append(o)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), hdr_m):
            cmd = [hipcc, *FLAGS, "-x", "hip", "-c", s, "-o", o]
            jobs.append(cmd)

    def run(cmd):
        if verbose:
            print("[build]", " ".join(cmd[-4:]), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        if verbose and r.stderr.strip():
            print(r.stderr.strip())

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if jobs or force or not os.path.exists(LIB):
        run([hipcc, "-shared", "-fPIC", "--offload-arch=gfx950", *objs, "-o", LIB])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))

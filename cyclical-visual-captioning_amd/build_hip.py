#!/usr/bin/env python3
"""Build libcvc_hip.so (gfx950) in-tree: cyclical-visual-captioning_amd/cvc/lib/libcvc_hip.so.

hipcc cross-compiles without a GPU; the .so is git-ignored but travels with the gpurun snapshot.
"""
from __future__ import annotations

import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "cvc", "lib")
OUT = os.path.join(OUT_DIR, "libcvc_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


# forms that were measured and do not pay: only in a library built with CVC_EXPERIMENTAL=1
EXPERIMENTAL_SOURCES = ("gemm_gsk.hip", "gemm_packed_ks.hip")


def experimental() -> bool:
    return os.environ.get("CVC_EXPERIMENTAL", "0") not in ("", "0")


def sources():
    return [f for f in sorted(glob.glob(os.path.join(CSRC, "*.hip")))
            if experimental() or os.path.basename(f) not in EXPERIMENTAL_SOURCES]


def deps():
    inc = os.path.join(os.path.dirname(HERE), "include")
    return sources() + sorted(glob.glob(os.path.join(CSRC, "*.h"))) + \
        [os.path.join(inc, h) for h in ("cvc_hip.h", "cvc_hip_blocks.h", "cvc_hip_experimental.h")]


def source_hash() -> str:
    """sha256 over the kernel sources + the C-ABI header (sorted by name): identifies the build that a PMC collection
    (profiles/traffic.json) belongs to; bench.py refuses byte counts collected on other sources."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(deps(), key=os.path.basename):
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


STAMP = os.path.join(OUT_DIR, "libcvc_hip.flags")     # the extra hipcc flags the in-tree .so was built with


def extra_flags() -> str:
    return " ".join(os.environ.get("CVC_EXTRA_HIPCC_FLAGS", "").split() + (["-DCVC_EXPERIMENTAL"] if experimental() else []))


def up_to_date() -> bool:
    """sources older than the .so AND the .so built with the flags asked for now: an ablation build (CVC_EXTRA_HIPCC_FLAGS=-D...)
    left behind by an interrupted experiment script is never taken for the product library"""
    if not os.path.exists(OUT):
        return False
    built_with = open(STAMP).read().strip() if os.path.exists(STAMP) else ""
    if built_with != extra_flags():
        return False
    t = os.path.getmtime(OUT)
    return all(os.path.getmtime(f) <= t for f in deps())


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and up_to_date():
        return OUT
    os.makedirs(OUT_DIR, exist_ok=True)
    objs = []
    obj_dir = os.path.join(HERE, "build")
    os.makedirs(obj_dir, exist_ok=True)
    procs = []
    for src in sources():
        obj = os.path.join(obj_dir, os.path.basename(src) + ".o")
        objs.append(obj)
        # hidden visibility: the .so exports the CVC_API declarations of include/cvc_hip.h and nothing else
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wno-comment", "-c", src, "-o", obj]
        cmd += extra_flags().split()      # e.g. -DCVC_ABL=1 for ablation builds, -DCVC_EXPERIMENTAL
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    objs = [o for o in objs if os.path.exists(o)]
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs + ["-ldl"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(STAMP, "w") as f:
        f.write(extra_flags() + "\n")
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))

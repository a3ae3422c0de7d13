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


# Kernels whose correctness leans on hand-written inline assembly (loads issued and waited for by statements the compiler cannot
# see through): their registers must never travel through scratch, where a value still in flight would be stored stale.  The
# build FAILS if the compiler reports scratch or spills for one of them (hipcc -Rpass-analysis=kernel-resource-usage), so that a
# compiler bump cannot regress this silently; the 128-row-vs-64-row bit-equality test in the default GPU suite is the other guard.
NO_SCRATCH_KERNELS = {"gemm_nn.hip": ("skinny_gemm_nn_split128_kernel",)}


def check_no_scratch(src: str, remarks: str):
    import re
    want = NO_SCRATCH_KERNELS.get(os.path.basename(src), ())
    seen = set()
    for block in remarks.split("remark: Function Name: ")[1:]:
        name = block.split()[0]
        hit = next((k for k in want if k in name), None)
        if hit is None:
            continue
        seen.add(hit)
        vals = {k: int(v) for k, v in re.findall(r"remark:\s+(ScratchSize \[bytes/lane\]|SGPRs Spill|VGPRs Spill): (\d+)", block)}
        bad = {k: v for k, v in vals.items() if v != 0}
        if bad or len(vals) < 3:
            raise RuntimeError(f"{os.path.basename(src)}: kernel {name} must not use scratch or spill (inline-assembly loads in "
                               f"flight): {vals}")
    missing = set(want) - seen
    if missing:
        raise RuntimeError(f"{os.path.basename(src)}: no resource-usage remark for {sorted(missing)} (renamed kernel?)")


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
        checked = os.path.basename(src) in NO_SCRATCH_KERNELS
        if checked:
            cmd.append("-Rpass-analysis=kernel-resource-usage")
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((cmd, src, checked, subprocess.Popen(cmd, stderr=subprocess.PIPE if checked else None, text=checked or None)))
    for cmd, src, checked, p in procs:
        err = p.communicate()[1] if checked else None
        if p.wait() != 0:
            if err:
                sys.stderr.write("\n".join(l for l in err.splitlines() if "kernel-resource-usage" not in l)[-8000:] + "\n")
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
        if checked:
            check_no_scratch(src, err)
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

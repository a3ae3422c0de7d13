"""What bench.py and bench_train.py share: the roofs, the host-core count, the PMC traffic look-up."""
from __future__ import annotations

import json
import os

ROOT = os.path.dirname(os.path.abspath(__file__))

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec); ~6.3 TB/s measured achievable
MFMA_F32_PEAK_TFLOPS = 157.3
MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense bf16 32x32x16 (same guide); the packed GEMMs issue 6 bf16 MFMAs per fp32 product


def usable_cores() -> int:
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota
    (os.cpu_count() reports the whole host and oversubscribes a containerised run)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    return max(1, n)


def pmc_traffic(kernel, args, over, mode="decode", beam=None, config_name=None):
    """HBM bytes per launch of `kernel` from the tracked PMC collection of THIS workload (profiles/traffic/<config>_beam<b>_<mode>.json,
    written by tools/collect_traffic.py from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this very command).
    The file records the hash of the kernel sources it was collected on; numbers from any other build are REFUSED
    (traffic = null plus the reason) instead of printed as if they were current."""
    cfg = config_name or args.config
    b = args.beam if beam is None else beam
    from cvc import synth as _synth
    # cfg2 and cfg3 are the same sizes: a decode collection of one serves the other
    names = [cfg] + [c for c in _synth.CONFIGS if c != cfg and cfg in _synth.CONFIGS and _synth.CONFIGS[c] == _synth.CONFIGS[cfg]]
    path = None
    for c in names:
        cand = os.path.join(ROOT, "profiles", "traffic", f"{c}_beam{b}_{mode}.json")
        if os.path.exists(cand):
            path = cand
            break
    if path is None:
        return None, f"profiles/traffic/{cfg}_beam{b}_{mode}.json absent"
    if over:
        return None, "dimension overrides on the command line: the tracked collection is for the named config"
    try:
        tf = json.load(open(path))
    except Exception as e:
        return None, f"{os.path.relpath(path, ROOT)} unreadable: {e}"
    import build_hip
    have = build_hip.source_hash()
    if tf.get("source_hash") != have:
        return None, f"stale: collected on kernel sources {tf.get('source_hash')}, this build is {have}"
    ent = tf.get("kernels", {}).get(kernel)
    if not ent:
        return None, f"no PMC entry for {kernel} in {os.path.relpath(path, ROOT)}"
    return int(ent["hbm_bytes"]), f"{ent['symbol'][:110]} ({ent['dispatches']} dispatches; 2 x FETCH_SIZE + WRITE_SIZE, KiB units; {os.path.basename(path)})"



"""Host arithmetic of the grouped stream-K launches (cvc_gsk_plan, include/cvc_hip.h): no GPU needed, the library only does
integer arithmetic here.  A plain-Python restatement of csrc/gsk.h checks that every unit of every tile is covered by exactly
one segment and that the slabs the plan asks for hold every segment."""
import ctypes as C

import pytest

pytestmark = pytest.mark.experimental      # cvc_gsk_plan lives in include/cvc_hip_experimental.h


def plan_py(ntile, nchunk, nwg):
    unit0, tot = [], 0
    for nt, nc in zip(ntile, nchunk):
        unit0.append(tot)
        tot += nt * nc
    U = -(-tot // nwg)
    maxseg = []
    for u0, nt, nc in zip(unit0, ntile, nchunk):
        maxseg.append(max((u0 + (t + 1) * nc - 1) // U - (u0 + t * nc) // U + 1 for t in range(nt)))
    return U, unit0, maxseg, tot


@pytest.mark.parametrize("ntile,nchunk,nwg", [
    ([32, 20], [128, 64], 256),          # cfg2: att-early || logits
    ([32, 4], [128, 64], 256),           # cfg2: lang-early || h2attn
    ([20], [64], 256),                   # logits alone
    ([64, 20], [256, 128], 256),         # cfg5
    ([16, 20], [64, 32], 256),           # cfg1
    ([1, 1, 1], [4, 2, 7], 3), ([2], [1], 256), ([4, 1], [4, 2], 304)])
def test_plan_matches_the_restatement_and_covers_every_unit(ntile, nchunk, nwg):
    from cvc import hip
    p = hip.gsk_plan(ntile, nchunk, nwg)
    U, unit0, maxseg, tot = plan_py(ntile, nchunk, nwg)
    assert (p["U"], p["unit0"], p["maxseg"]) == (U, unit0, maxseg)
    nwg_used = -(-tot // U)
    assert nwg_used <= nwg
    # walk the workgroups the way the kernel does: (group, tile, first chunk, length, segment index) of every run
    seen = {}
    for w in range(nwg_used):
        u, uend = w * U, min((w + 1) * U, tot)
        while u < uend:
            g = max(i for i in range(len(ntile)) if u >= unit0[i])
            rel = u - unit0[g]
            tile, c_lo = divmod(rel, nchunk[g])
            n = min(uend - u, nchunk[g] - c_lo)
            seg = w - (unit0[g] + tile * nchunk[g]) // U
            assert 0 <= seg < maxseg[g]
            assert (g, tile, seg) not in seen                     # one run per (tile, segment)
            seen[(g, tile, seg)] = (c_lo, n)
            u += n
    for g in range(len(ntile)):
        for t in range(ntile[g]):
            nseg = (unit0[g] + (t + 1) * nchunk[g] - 1) // U - (unit0[g] + t * nchunk[g]) // U + 1
            runs = [seen[(g, t, s)] for s in range(nseg)]
            assert (g, t, nseg) not in seen
            assert runs[0][0] == 0 and sum(n for _, n in runs) == nchunk[g]           # the segments tile the chunk range in order
            for (c0, n0), (c1, _) in zip(runs, runs[1:]):
                assert c0 + n0 == c1


def test_plan_rejects_bad_arguments():
    from cvc import hip
    L = hip.lib()
    U, u0, ms = C.c_int(), (C.c_int * 1)(), (C.c_int * 1)()
    one = lambda v: (C.c_int * 1)(v)
    assert L.cvc_gsk_plan(one(0), one(4), 1, 8, C.byref(U), u0, ms) == -1
    assert L.cvc_gsk_plan(one(2), one(0), 1, 8, C.byref(U), u0, ms) == -1
    assert L.cvc_gsk_plan(one(2), one(4), 4, 8, C.byref(U), u0, ms) == -1

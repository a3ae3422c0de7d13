#!/usr/bin/env python3
"""Micro-benchmark of the grouped stream-K launches and their consumers at the decode shapes (GPU box only).

Times, with HIP events over batches of launches that rotate over several copies of the weights (so that nothing is served
from the Infinity Cache that would not be in a decode step, where ~0.95 GB are streamed between two uses of a weight):
  ks      cvc_packed_lstm_ks_fwd over K = 2R (the round-2 K-split kernel + its finishing launch), the reference point
  early   cvc_gsk_gemm, one LSTM group over K = 2R alone (aligned: U divides the tile)
  lang    cvc_gsk_gemm {lang-early, h2attn}
  att     cvc_gsk_gemm {att-early (embedding segment skipped), logits}
  late_l / late_a   cvc_packed_lstm_late_fwd over K = R / K = E with the partial tiles
  full_l / full_a   cvc_packed_lstm_fwd over the whole K (the one-launch kernels)
  select  cvc_top2_slab
Prints us per launch and the weight-stream rate."""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cyclical-visual-captioning_amd"))
import torch  # noqa: E402
from cvc import hip  # noqa: E402
from cvc.decode import pack_weights, to_quad  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--R", type=int, default=2048)
ap.add_argument("--E", type=int, default=1024)
ap.add_argument("--A", type=int, default=1024)
ap.add_argument("--V", type=int, default=5000)
ap.add_argument("--M", type=int, default=64)
ap.add_argument("--nwg", type=int, default=256)
ap.add_argument("--copies", type=int, default=3)
ap.add_argument("--iters", type=int, default=200)
ap.add_argument("--warm", type=float, default=0.3)
ap.add_argument("--only", default="")
args = ap.parse_args()
R, E, A, V, M = args.R, args.E, args.A, args.V, args.M
dev = torch.device("cuda:0")
L = hip.lib()
st = torch.cuda.current_stream().cuda_stream
g = torch.Generator().manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=g)
KA, KL = 2 * R + E, 3 * R
w_att = [pack_weights((rnd(4 * R, KA) * 0.02).to(dev), R) for _ in range(args.copies)]
w_lang = [pack_weights((rnd(4 * R, KL) * 0.02).to(dev), R) for _ in range(args.copies)]
w_o = [pack_weights((rnd(V, R) * 0.02).to(dev)) for _ in range(args.copies)]
w_h = [pack_weights((rnd(A, R) * 0.02).to(dev)) for _ in range(args.copies)]
xa, xl = to_quad(rnd(M, KA).to(dev)), to_quad(rnd(M, KL).to(dev))
cq = to_quad(rnd(M, R).to(dev))
h1, h2, c2 = (torch.zeros(R // 4, 64, 4, device=dev) for _ in range(3))
gb = (rnd(M, 4 * R) * 0.1).to(dev)
bo = torch.zeros(V, device=dev)
table = rnd(V, E).to(dev)
nblk_v = (V + 31) // 32
nt_r, nt_v, nt_a = R // 64, (nblk_v + 7) // 8, (A // 32 + 7) // 8
pa = hip.gsk_plan([nt_r, nt_v], [2 * R // 32, R // 32], args.nwg)
pl = hip.gsk_plan([nt_r, nt_a], [2 * R // 32, R // 32], args.nwg)
pe = hip.gsk_plan([nt_r], [2 * R // 32], args.nwg)
print("plans: att+logits", pa, "| lang+h2attn", pl, "| early alone", pe)
slab = lambda nt, ms: torch.zeros(nt * ms * 16384, device=dev)
s_att, s_o = slab(nt_r, max(pa["maxseg"][0], pe["maxseg"][0])), slab(nt_v, pa["maxseg"][1])
s_lang, s_q = slab(nt_r, pl["maxseg"][0]), slab(nt_a, pl["maxseg"][1])
S = L.cvc_packed_lstm_ks_slices(2 * R, R)
ks_slab = torch.empty(max(S, 1) * (R // 8) * 2048, device=dev)
words = torch.zeros(M, dtype=torch.int64, device=dev)
lp = torch.zeros(M, device=dev)
embq = torch.zeros(E // 4, 64, 4, device=dev)
ws_att, ws_lang, ws_r = KA // 4 * 128, KL // 4 * 128, R // 4 * 128
qoff = lambda t, k0: t.data_ptr() + (k0 // 4) * 64 * 16
keep = []


def groups(*gs):
    arr = (hip.GskGroup * len(gs))()
    for i, x in enumerate(gs):
        arr[i] = hip.GskGroup(*x)
    keep.append(arr)
    return arr


seg_att = hip.GskSegs(s_att.data_ptr(), pa["unit0"][0], 2 * R // 32, pa["U"], pa["maxseg"][0])
seg_lang = hip.GskSegs(s_lang.data_ptr(), pl["unit0"][0], 2 * R // 32, pl["U"], pl["maxseg"][0])
seg_o = hip.GskSegs(s_o.data_ptr(), pa["unit0"][1], R // 32, pa["U"], pa["maxseg"][1])


def f_ks(i):      # K = 2R: the lang matrix's [R, 3R) columns
    return L.cvc_packed_lstm_ks_fwd(w_lang[i].data_ptr() + (R // 4) * 128 * 4, qoff(xl, R), 2 * R, None, None, None, cq.data_ptr(), M, R,
                                    h1.data_ptr(), h2.data_ptr(), c2.data_ptr(), ks_slab.data_ptr(), ws_lang, st)


# group descriptors are built once per weight copy: the timed loops must not be bound by ctypes object construction on the host
G_early = [groups((w_lang[i].data_ptr(), ws_lang, xl.data_ptr(), R // 8, 2 * R // 32, 0, R // 32, s_att.data_ptr(), pe["maxseg"][0]))
           for i in range(args.copies)]
G_lang = [groups((w_lang[i].data_ptr(), ws_lang, xl.data_ptr(), R // 8, 2 * R // 32, 0, R // 32, s_lang.data_ptr(), pl["maxseg"][0]),
                 (w_h[i].data_ptr(), ws_r, qoff(xl, R), A // 32, R // 32, 0, 0, s_q.data_ptr(), pl["maxseg"][1])) for i in range(args.copies)]
G_att = [groups((w_att[i].data_ptr(), ws_att, xa.data_ptr(), R // 8, 2 * R // 32, R // 32, E // 32, s_att.data_ptr(), pa["maxseg"][0]),
                (w_o[i].data_ptr(), ws_r, xa.data_ptr(), nblk_v, R // 32, 0, 0, s_o.data_ptr(), pa["maxseg"][1])) for i in range(args.copies)]


def f_early(i):
    return L.cvc_gsk_gemm(G_early[i], 1, pe["U"], st)


def f_lang(i):
    return L.cvc_gsk_gemm(G_lang[i], 2, pl["U"], st)


def f_att(i):
    return L.cvc_gsk_gemm(G_att[i], 2, pa["U"], st)


def f_late_l(i):
    return L.cvc_packed_lstm_late_fwd(w_lang[i].data_ptr(), ws_lang, xl.data_ptr(), R, None, None, gb.data_ptr(), cq.data_ptr(), M, R,
                                      h1.data_ptr(), h2.data_ptr(), c2.data_ptr(), C.byref(seg_lang), st)


def f_late_a(i):
    return L.cvc_packed_lstm_late_fwd(w_att[i].data_ptr() + (R // 4) * 128 * 4, ws_att, qoff(xa, R), E, None, None, gb.data_ptr(), cq.data_ptr(), M, R,
                                      h1.data_ptr(), h2.data_ptr(), c2.data_ptr(), C.byref(seg_att), st)


def f_late_l0(i):
    return L.cvc_packed_lstm_late_fwd(w_lang[i].data_ptr(), ws_lang, xl.data_ptr(), R, None, None, gb.data_ptr(), cq.data_ptr(), M, R,
                                      h1.data_ptr(), h2.data_ptr(), c2.data_ptr(), None, st)


def f_full_l(i):
    return L.cvc_packed_lstm_fwd(w_lang[i].data_ptr(), xl.data_ptr(), KL, None, None, gb.data_ptr(), cq.data_ptr(), M, R, h1.data_ptr(),
                                 h2.data_ptr(), c2.data_ptr(), st)


def f_full_a(i):
    return L.cvc_packed_lstm_fwd(w_att[i].data_ptr(), xa.data_ptr(), KA, None, None, gb.data_ptr(), cq.data_ptr(), M, R, h1.data_ptr(),
                                 h2.data_ptr(), c2.data_ptr(), st)


def f_select(i):
    return L.cvc_top2_slab(C.byref(seg_o), bo.data_ptr(), V, M, 1, words.data_ptr(), 1, lp.data_ptr(), table.data_ptr(), E, embq.data_ptr(), 0, st)


MB = 1e6
cases = [("ks", f_ks, 4 * 4 * R * 2 * R / MB), ("early", f_early, 4 * 4 * R * 2 * R / MB),
         ("lang", f_lang, 4 * (4 * R * 2 * R + A * R) / MB), ("att", f_att, 4 * (4 * R * 2 * R + V * R) / MB),
         ("late_l", f_late_l, 4 * 4 * R * R / MB), ("late_l0", f_late_l0, 4 * 4 * R * R / MB), ("late_a", f_late_a, 4 * 4 * R * E / MB),
         ("full_l", f_full_l, 4 * 4 * R * KL / MB), ("full_a", f_full_a, 4 * 4 * R * KA / MB), ("select", f_select, 0.0)]
# T(n): one LSTM group, aligned (U = nchunk / 8 at 256 workgroups), over growing K ranges of the lang matrix
for nch in (32, 64, 128, 192):
    psw = hip.gsk_plan([nt_r], [nch], args.nwg)
    Gs = [groups((w_lang[i].data_ptr(), ws_lang, xl.data_ptr(), R // 8, nch, 0, 0, s_att.data_ptr(), psw["maxseg"][0])) for i in range(args.copies)]
    cases.append((f"sweep{nch}(U={psw['U']})", (lambda i, Gs=Gs, psw=psw: L.cvc_gsk_gemm(Gs[i], 1, psw["U"], st)), 4 * 4 * R * nch * 32 / MB))
only = set(args.only.split(",")) if args.only else None
for name, fn, mb in cases:
    if only and name.split("(")[0].rstrip("0123456789") not in only and name not in only:
        continue
    for i in range(args.copies):
        assert fn(i) == 0, name
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    k = 0
    while time.perf_counter() - t0 < args.warm:          # settled clocks: a cold GPU runs the first milliseconds slower
        for _ in range(20):
            fn(k % args.copies); k += 1
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(args.iters):
        fn(k % args.copies)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / args.iters * 1e3
    print(f"{name:14s} {us:8.2f} us" + (f"   {mb:7.1f} MB of weights -> {mb / us / 1e6 * 1e6:6.2f} TB/s" if mb else ""))

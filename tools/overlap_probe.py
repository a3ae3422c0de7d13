#!/usr/bin/env python3
"""Probe: does the word-independent 80 % of the next att-LSTM gate GEMM overlap with logits + word selection
when they run on two streams?  (GPU box only.)  Prints serial vs concurrent time per pair."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cyclical-visual-captioning_amd"))
import torch
from cvc import hip, synth
from cvc.decode import pack_weights

dev = torch.device("cuda:0")
L = hip.lib()
R, E, V, M = 2048, 1024, 5000, 64
g = torch.Generator(device="cpu").manual_seed(0)
w1 = pack_weights((torch.randn(4 * R, 2 * R, generator=g) * 0.02).to(dev))          # P1: natural row order, K = 2R
wo = pack_weights((torch.randn(V, R, generator=g) * 0.02).to(dev))
w2 = pack_weights((torch.randn(4 * R, E, generator=g) * 0.02).to(dev), R)           # P2: LSTM order, K = E
xa = torch.randn((2 * R + E) // 4, 64, 4, generator=g).to(dev)
g1 = torch.zeros(M, 4 * R, device=dev)
part = torch.zeros((V + 31) // 32, 64, 6, device=dev)
words = torch.zeros(M, dtype=torch.int64, device=dev)
lp = torch.zeros(M, device=dev)
emb = torch.randn(V, E, generator=g).to(dev)
cq = torch.zeros(R // 4, 64, 4, device=dev)
c2 = torch.zeros_like(cq); h1 = torch.zeros_like(cq)
bo = torch.zeros(V, device=dev)
s_side = torch.cuda.Stream()


def p1(stream):
    hip._check(L.cvc_packed_linear_fwd(w1.data_ptr(), xa.data_ptr(), 2 * R, None, M, 4 * R, 1, g1.data_ptr(), 4 * R, None, stream), "p1")


def head(stream):
    hip._check(L.cvc_packed_linear_fwd(wo.data_ptr(), xa.data_ptr(), R, bo.data_ptr(), M, V, 1, None, V, part.data_ptr(), stream), "logits")
    hip._check(L.cvc_top2_final(part.data_ptr(), (V + 31) // 32, M, 1, words.data_ptr(), 1, lp.data_ptr(), emb.data_ptr(), E,
                                xa.data_ptr() + (2 * R // 4) * 64 * 16, 0, stream), "top2")


def p2(stream):
    hip._check(L.cvc_packed_lstm_fwd(w2.data_ptr(), xa.data_ptr() + (2 * R // 4) * 64 * 16, E, None, None, g1.data_ptr(), cq.data_ptr(),
                                     M, R, h1.data_ptr(), None, c2.data_ptr(), stream), "p2")


def timed(fn, n=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


main = torch.cuda.current_stream()

def serial():
    s = main.cuda_stream
    p1(s); head(s); p2(s)

def concurrent():
    s = main.cuda_stream
    s_side.wait_stream(main)
    p1(s_side.cuda_stream)
    head(s)
    main.wait_stream(s_side)
    p2(s)

print("P1 alone %.1f us | head alone %.1f us | P2 alone %.1f us" % (timed(lambda: p1(main.cuda_stream)), timed(lambda: head(main.cuda_stream)),
                                                                   timed(lambda: p2(main.cuda_stream))))
print("serial P1+head+P2 %.1f us | two-stream %.1f us" % (timed(serial), timed(concurrent)))


def graphed(fn, reps=20):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    g_ = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g_):
        for _ in range(reps):
            fn()
    return g_


def serial_cur():
    s = torch.cuda.current_stream().cuda_stream
    p1(s); head(s); p2(s)


def concurrent_cur():
    cur = torch.cuda.current_stream()
    s_side.wait_stream(cur)
    p1(s_side.cuda_stream)
    head(cur.cuda_stream)
    cur.wait_stream(s_side)
    p2(cur.cuda_stream)


for name, fn in (("serial", serial_cur), ("two-stream", concurrent_cur)):
    gr = graphed(fn)
    for _ in range(3):
        gr.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        gr.replay()
    torch.cuda.synchronize()
    print("graph %s: %.1f us per P1+head+P2" % (name, (time.perf_counter() - t0) / 20 / 20 * 1e6))

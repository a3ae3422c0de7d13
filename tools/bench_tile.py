#!/usr/bin/env python3
"""Micro-benchmark of the tile GEMM (csrc/gemm_tile.hip) at the beam shapes: time per launch, bf16 MFMA rate."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cyclical-visual-captioning_amd"))
import torch
from cvc import hip
from cvc.decode import pack_weights_tile, to_frag

dev = torch.device("cuda:0")
shapes = [("att_lstm", 320, 5120, 8192, 4), ("lang_lstm", 320, 6144, 8192, 4), ("h2attn", 320, 2048, 1024, 16),
          ("logits", 320, 2048, 5000, 6), ("lang cfg5", 320, 12288, 16384, 2), ("lang b=1 M=150", 150, 6144, 8192, 4)]
big_shapes = [("dW lstm block", 8192, 2560, 2048, 1), ("dW emb block", 8192, 2560, 1024, 1), ("GRU input proj", 30720, 2048, 6144, 1),
              ("GRU dW_ih", 6144, 30720, 2048, 1), ("frame embed", 30720, 2048, 1024, 1), ("head dX", 1280, 5008, 2048, 1),
              ("hoisted gates", 1280, 1024, 8192, 1)]
if os.environ.get("BIG"):
    shapes = big_shapes
if len(sys.argv) > 1:
    shapes = [s for s in shapes if s[0].startswith(sys.argv[1])]
modes = [int(a) for a in sys.argv[2:]] or [1]
for mode in modes:
  hip.lib().cvc_tile_gemm_loaders(mode if mode < 10 else 3)
  hip.lib().cvc_tile_gemm_big(1 if mode >= 10 else 0, 1)          # mode 10: the 256 x 256 form where it applies
  print("loader waves:", mode)
  for name, M, K, N, ks in shapes:
      x = torch.randn(M, K, device=dev)
      w = torch.randn(N, K, device=dev) / K ** 0.5
      wb = pack_weights_tile(w)
      xb = to_frag(x, hip.tile_rows_alloc(M))
      parts = torch.empty(ks, M, N, device=dev)
      for _ in range(3):
          hip.tile_gemm(wb, xb, 0, K, M, N, ks, parts)
      torch.cuda.synchronize()
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      n = 20
      e0.record()
      for _ in range(n):
          hip.tile_gemm(wb, xb, 0, K, M, N, ks, parts)
      e1.record()
      torch.cuda.synchronize()
      us = e0.elapsed_time(e1) / n * 1e3
      fl = 2.0 * M * K * N
      print(f"{name:16s} M={M} K={K} N={N} ks={ks}: {us:8.1f} us  {fl / us / 1e6:7.1f} TF fp32-equivalent  {6 * fl / us / 1e6:7.1f} TF bf16 executed "
            f"({6 * fl / us / 1e6 / 2500 * 100:.0f}% of 2.5 PF)", flush=True)

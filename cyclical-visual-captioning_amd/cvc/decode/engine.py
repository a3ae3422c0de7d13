"""The decode engine: binds a checkpoint (DecodeWeights) and one batch of clip features to a flat per-step launch list, runs it
through the C drivers (cvc_decode_greedy / cvc_decode_beam: one host call per decode) or eagerly, optionally as a HIP-graph replay.
The per-path buffers and launch lists live in path_packed / path_tile / path_ring (/ path_experimental)."""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Tuple


import torch

from .. import hip
from .weights import *          # noqa: F401,F403  (packers, layouts, cache plan, switches)
from .weights import _segs
from .path_packed import PackedPath
from .path_tile import TilePath
from .path_ring import RingPath
from .path_experimental import ExperimentalPaths

def _capture_mode() -> str:
    """"global" (torch's default) unless a c10d "nccl" process group is up in this process: its watchdog thread polls the events of
    earlier collectives with hipEventQuery, which fails with hipErrorStreamCaptureUnsupported while ANOTHER thread captures in
    global mode -- the watchdog then dies with that exception and takes the rank down.  "thread_local" confines the capture's
    restrictions to the capturing thread; the decode graph contains no collective, so no event of that group is ever recorded in
    the capturing stream.  (The package's own runs -- bench.py, cvc.main -- keep torch.distributed on gloo and the exchange on
    cvc.comm.RcclComm: no such thread exists there.)"""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_backend() == "nccl":
        return "thread_local"
    return "global"


class DecodeEngine(PackedPath, TilePath, RingPath, ExperimentalPaths):
    """Binds weights + one batch of clip features to preallocated state and a launch list."""
    _warm = set()

    def __init__(self, weights: DecodeWeights, feats: Dict[str, torch.Tensor], T: int, unk_idx: int, beam: int = 1,
                 inv_temp: float = 1.0, own_features: bool = False, path: str = "auto", gate_ksplit: Optional[bool] = None,
                 driver: bool = True, gsk: Optional[bool] = None, embgate: Optional[bool] = None, lang_ksx: Optional[bool] = None):
        """driver: enqueue the decode through the C-ABI drivers cvc_decode_greedy / cvc_decode_beam (one host call per decode);
        False walks the launch list in Python (one ctypes call per kernel; tests compare the two).
        embgate: packed path only -- the embedding-gate schedule (the embedded word's share of the att-LSTM gates is a row of
        a per-checkpoint table: 34 MB less to stream per step at config 2, and the gate GEMM no longer waits for the word).  None = on when the table fits EMBGATE_MAX_BYTES; tests compare on / off.
        lang_ksx: packed path, R = 2048 -- the language cell on the K-split gate GEMM with the exchange finish
        (cvc_packed_lstm_ksx_fwd: activations read once per 256 gate rows instead of once per 32; the tile's 8 K slices
        exchange their partial tiles inside the launch).  None = off (measured no faster inside the decode graph; CVC_LANG_KSX=1:
        on); the exchange's error word is checked after the first decode and the engine re-binds without it if it is set.
        gsk: packed path only -- True selects the grouped stream-K schedule (csrc/gemm_gsk.hip; measured slower than the
        embedding-gate schedule, kept selectable and tested; needs R % 64 == 0, split-product arithmetic).
        path: "auto" picks packed (greedy, <= 64 rows) / tile (> 64 rows or beams) / ring (odd widths); "ring" forces the
        row-major fallback kernels (tests compare the paths).
        own_features: keep private copies of the clip features, so that the bound launch list (and a captured HIP
        graph) can be reused for the next batch of the same shape through load_features()."""
        W = self.W = weights
        self.T, self.unk, self.beam = int(T), int(unk_idx), int(beam)
        fc, conv, pconv = feats["fc_feats"], feats["conv_feats"], feats["p_conv_feats"]
        pool, ppool = feats["pool_feats"], feats["p_pool_feats"]
        mask = feats["pnt_mask"][:, 1:] if feats["pnt_mask"].shape[1] == pool.shape[1] + 1 else feats["pnt_mask"]
        self.B, self.N, self.F = pool.shape[0], pool.shape[1], conv.shape[1]
        B, N, Fr, R, A, V = self.B, self.N, self.F, W.R, W.A, W.V
        dev = pool.device
        for name, t, shape in (("fc_feats", fc, (B, R)), ("conv_feats", conv, (B, Fr, R)), ("p_conv_feats", pconv, (B, Fr, A)),
                               ("pool_feats", pool, (B, N, R)), ("p_pool_feats", ppool, (B, N, A))):
            if tuple(t.shape) != shape:
                raise RuntimeError(f"DecodeEngine: {name} has shape {tuple(t.shape)}, expected {shape}")
            hip._dev(t, name=name)
        self.mask = hip._mask(mask)
        if own_features:
            fc, conv, pconv, pool, ppool = (t.clone() for t in (fc, conv, pconv, pool, ppool))
            self.mask = self.mask.clone()
        self.own_features = own_features
        self.feats = (fc, conv, pconv, pool, ppool)
        nb = lambda t: t.numel() * t.element_size()
        rows = self.rows = B * self.beam
        f32 = dict(device=dev, dtype=torch.float32)
        z = lambda *s: torch.zeros(*s, **f32)
        # ping-pong recurrent state: index t & 1 is read, (t+1) & 1 is written
        self.h_att, self.c_att = [z(rows, R), z(rows, R)], [z(rows, R), z(rows, R)]
        self.h_lang, self.c_lang = [z(rows, R), z(rows, R)], [z(rows, R), z(rows, R)]
        self.q = z(rows, A)
        self.scores_r, self.scores_f = z(rows, N), z(rows, Fr)
        self.attn_f = z(rows, Fr)
        self.ctx_sum = z(rows, R)
        self.logits = z(rows, V)
        self.gate_fc = z(rows, 4 * R)      # step-invariant part of the att-LSTM gates: fc x W_ih[:, R:2R] + b_ih + b_hh
        self.QSPLIT = 8                    # h2attn runs split-K over the chip; attn_scores sums the slices
        self.q_parts = z(self.QSPLIT, rows, A)
        self.emb = z(rows, W.E)            # relu(Emb[word_t]), written by the word-selection kernel of step t-1
        self.top2_part = z((V + 31) // 32, 64, 6)
        self.att_steps = z(self.T, rows, N)                       # post-softmax region attention per step
        self.words = torch.zeros(self.T + 1, rows, dtype=torch.int64, device=dev)   # words[0] = BOS = 0
        self.logprob = z(self.T, rows)
        # fc is per clip; beams of a clip read the same row through a row-gather index
        self.fc = fc
        self.clip_of_row = torch.arange(rows, device=dev, dtype=torch.int64) // self.beam
        if self.beam > 1:
            self.score = z(2, rows)
            self.done = torch.zeros(2, rows, dtype=torch.uint8, device=dev)
            self.parent = torch.zeros(self.T, rows, dtype=torch.int64, device=dev)
            self.bt_seq = torch.zeros(self.B, self.T, dtype=torch.int64, device=dev)      # rank-0 hypothesis (cvc_beam_backtrack)
            self.bt_att = z(self.B, self.T, N)
            self.gather_tmp = [z(rows, R) for _ in range(4)]
            self.beam_ws = z(17 * rows)
        self.inv_temp = float(inv_temp)
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self._keep: List = []
        self.packed = self.beam == 1 and rows <= 64 and R % 32 == 0 and W.E % 32 == 0 and A % 32 == 0
        # packed path: K-split gate GEMMs (activations shared through LDS, csrc/gemm_packed_ks.hip) where the shape allows
        # (True: partial tiles + a finishing launch; "fused": the last-arriving K slice of a tile finishes it in the same launch)
        self.gate_ksplit = GATE_KSPLIT_DEFAULT if gate_ksplit is None else gate_ksplit
        self.gate_fused = self.gate_ksplit == "fused"
        self.gate_ksplit = bool(self.gate_ksplit)
        gsk_ok = self.packed and R % 64 == 0 and not self.gate_ksplit and hip.gemm_packed_split(-1) == 2
        if gsk and not gsk_ok:
            raise RuntimeError("DecodeEngine: the stream-K schedule needs the packed path, R % 64 == 0 and cvc_gemm_packed_split(2)")
        self.gsk = False if gsk is None else bool(gsk)
        # more than 64 live rows (beam search, big greedy batches): bf16-fragment tile GEMMs (csrc/gemm_tile.hip)
        self.tile = (not self.packed) and (self.beam > 1 or rows > 64) and R % 16 == 0 and W.E % 16 == 0 and path != "ring"
        eg_ok = (self.packed and not self.gsk and not self.gate_ksplit) or self.tile
        if embgate and not eg_ok:
            raise RuntimeError("DecodeEngine: the embedding-gate schedule needs the packed path (without gsk / gate_ksplit) or the tile path")
        self.embgate = (eg_ok and 4 * V * 4 * R <= EMBGATE_MAX_BYTES) if embgate is None else bool(embgate)
        # what stays in the Infinity Cache between steps: small linear weights, then (embedding-gate schedule on the packed path) the
        # attention cell's gate matrix over K = 2R if it fits, then the largest subset of the feature tensors
        keep = cache_plan(4 * (V * R + A * R), {"ppool": nb(ppool), "pconv": nb(pconv), "pool": nb(pool), "conv": nb(conv)},
                          gate_weight_bytes=4 * 4 * R * 2 * R if (self.packed and self.embgate and rows > 32) else None)
        self.att_w_cached = bool(keep.get("att_w", False))
        # cvc_attn_set.stream: bit 0 = proj read non-temporally, bit 1 = ctx
        self.stream_r = (0 if keep["ppool"] else 1) | (0 if keep["pool"] else 2)
        self.stream_f = (0 if keep["pconv"] else 1) | (0 if keep["conv"] else 2)
        self._plan = None
        self._driver = driver
        exp = hip.experimental_built()          # gsk / gate_ksplit / lang_ksx: forms of include/cvc_hip_experimental.h
        if not exp and (self.gsk or self.gate_ksplit or lang_ksx):
            raise RuntimeError("DecodeEngine: gsk / gate_ksplit / lang_ksx are experimental schedules (include/cvc_hip_experimental.h): "
                               "this libcvc_hip.so was built without CVC_EXPERIMENTAL=1")
        ksx_ok = (exp and self.packed and not self.gsk and not self.gate_ksplit and R == 2048 and self.T > 1 and
                  hip.gemm_packed_split(-1) == 2 and int(hip.lib().cvc_packed_lstm_ks_slices(3 * R, R)) == 8)
        if lang_ksx and not ksx_ok:
            raise RuntimeError("DecodeEngine: lang_ksx needs the packed path at R = 2048, T > 1, split-product arithmetic")
        self.lang_ksx = ksx_ok and (LANG_KSX_DEFAULT if lang_ksx is None else bool(lang_ksx))
        self._ksx_checked = False
        if self.lang_ksx:
            self.ksx_slab = torch.empty(8 * (R // 8) * 2048, device=dev, dtype=torch.float32)
            self.ksx_flags = torch.zeros(R // 8 + 1, device=dev, dtype=torch.int32)
        if self.packed:
            self._alloc_packed()
            self._launches = self._build_packed()
        elif self.tile:
            self._alloc_tile()
            self._launches = self._build_tile()
        else:
            self._launches = self._build()
        if driver and (self.tile or (self.packed and not (self.ks_att or self.ks_lang))):
            self._bind_driver()

    # ------------------------------------------------------------------ C-ABI decode driver (csrc/decode_driver.hip)

    def _bind_driver(self):
        """Bind every buffer of this engine into a cvc_decode_desc and create the plan: run() / capture() then enqueue the
        whole decode with ONE call (cvc_decode_greedy / cvc_decode_beam) instead of walking the launch list in Python.  The
        Python launch list stays for run_timed() (per-launch HIP events) and as the reference the driver is tested against."""
        W, L = self.W, hip.lib()
        ptr = lambda t: None if t is None else t.data_ptr()
        d = hip.DecodeDesc()
        d.B, d.beam, d.T, d.N, d.F, d.R, d.A, d.E, d.V = self.B, self.beam, self.T, self.N, self.F, W.R, W.A, W.E, W.V
        d.unk_idx, d.attn_kind, d.inv_temp = self.unk, W.kind, self.inv_temp
        d.stream_r, d.stream_f = self.stream_r, self.stream_f
        for k in ("b_ih_att", "b_hh_att", "b_ih_lang", "b_hh_lang", "b_h", "w_a", "b_a", "b_o", "embed"):
            setattr(d, k, ptr(getattr(W, k)))
        fc, conv, pconv, pool, ppool = self.feats
        d.fc, d.conv, d.pconv, d.pool, d.ppool, d.mask = ptr(fc), ptr(conv), ptr(pconv), ptr(pool), ptr(ppool), ptr(self.mask)
        d.words, d.att_steps, d.logprob = ptr(self.words), ptr(self.att_steps), ptr(self.logprob)
        d.scores_r, d.scores_f, d.attn_f = ptr(self.scores_r), ptr(self.scores_f), ptr(self.attn_f)
        if self.packed:
            R = W.R
            d.path, d.qsplit = 0, self.QSPLIT
            d.w_att, d.w_lang, d.w_h, d.w_o = ptr(W.p_att2 if self.embgate else W.p_att), ptr(W.p_lang), ptr(W.p_h), ptr(W.p_o)
            w_fc = W.w_ih_att[:, R:2 * R]
            d.w_fc, d.ld_w_fc = w_fc.data_ptr(), w_fc.stride(0)
            d.gate_fc, d.q_parts, d.top2_part = ptr(self.gate_fc), ptr(self.q_parts), ptr(self.top2_part)
            for name, bufs in (("xa", self.XA), ("xl", self.XL), ("ca", self.cA), ("cl", self.cL)):
                arr = getattr(d, name)
                arr[0], arr[1] = ptr(bufs[0]), ptr(bufs[1])
            d.xa0_init = ptr(self.XA0_init)
            if self.embgate:
                d.w_att = ptr(W.p_att2)
                d.emb_gate, d.sel_counter = ptr(W.t_embgate), ptr(self.sel_counter)
                d.att_w_cached = int(self.att_w_cached)
            if self.lang_ksx:
                d.lang_ksx, d.ksx_slab, d.ksx_flags = 1, ptr(self.ksx_slab), ptr(self.ksx_flags)
            if self.gsk:
                d.gsk_nwg = self.gsk_nwg
                d.slab_att, d.slab_lang, d.slab_q, d.slab_o = (ptr(self.slab_att), ptr(self.slab_lang), ptr(self.slab_q),
                                                              ptr(self.slab_o))
        else:
            d.path = 1
            d.ks_gate, d.ks_q, d.ks_o, d.ks_fc = self.ks_gate, self.ks_q, self.ks_o, self.ks_fc
            d.w_att, d.w_lang, d.w_h, d.w_o, d.w_fc_frag = ptr(W.t_att2 if self.embgate else W.t_att), ptr(W.t_lang), ptr(W.t_h), ptr(W.t_o), ptr(W.t_fc)
            if self.embgate:
                d.emb_gate = ptr(W.t_embgate)
            d.gate_fc, d.q, d.q_parts, d.logits = ptr(self.gate_fc_clip), ptr(self.q), ptr(self.parts_q), ptr(self.logits)
            for name, t in (("xaf", self.XAf), ("xlf", self.XLf), ("xhf", self.XHf), ("xff", self.XFf)):
                p_, s_ = hip._frag_ptr(t)
                setattr(d, name, p_)
                setattr(d, name + "_stride", s_)
            d.parts_gate, d.parts_o, d.parts_fc = ptr(self.parts_gate), ptr(self.parts_o), ptr(self.parts_fc)
            d.h_att, d.c_att, d.h_lang, d.c_lang = ptr(self.t_h_att), ptr(self.t_c_att), ptr(self.t_h_lang), ptr(self.t_c_lang)
            d.c_att_prev, d.c_lang_prev, d.zero_state = ptr(self.t_c_att_prev), ptr(self.t_c_lang_prev), ptr(self.t_zero)
            if self.beam > 1:
                d.score, d.done, d.parent, d.beam_ws = ptr(self.score), ptr(self.done), ptr(self.parent), ptr(self.beam_ws)
        plan = C.c_void_p()
        hip._check(L.cvc_decode_plan_create(C.byref(d), C.byref(plan)), "cvc_decode_plan_create")
        self._desc, self._plan = d, plan
        self._plan_call = L.cvc_decode_beam if self.beam > 1 else L.cvc_decode_greedy

    def __del__(self):
        plan = getattr(self, "_plan", None)
        if plan is not None and plan.value:
            try:
                hip.lib().cvc_decode_plan_destroy(plan)
            except Exception:
                pass
            self._plan = None

    def _run_driver(self):
        hip._check(self._plan_call(self._plan, torch.cuda.current_stream().cuda_stream), "cvc_decode_greedy/beam")

    # ------------------------------------------------------------------ packed path (greedy, rows <= 64)

    def _reset(self):
        if self.tile:
            self.words[0].zero_()
            if self.beam > 1:
                self.score.zero_()
                self.done.zero_()
            return
        if self.packed:
            self.XA[0].copy_(self.XA0_init)
            self.XL[0].zero_()
            self.cA[0].zero_()
            self.cL[0].zero_()
            self.words[0].zero_()
            return
        for bufs in (self.h_att, self.c_att, self.h_lang, self.c_lang):
            bufs[0].zero_()
        self.words[0].zero_()
        if self.beam > 1:
            self.score.zero_()
            self.done.zero_()

    def _run_launches(self, timers=None):
        """timers: optional dict name -> list of (start_event, end_event), filled per launch
        (HIP events on the launch stream; used by bench.py for per-kernel durations)."""
        stream = torch.cuda.current_stream().cuda_stream
        for name, fn, args in self._python_launches():
            if timers is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            if fn == "copy":
                args[0].copy_(args[1])
            else:
                rc = fn(*args, stream)
                if rc != 0:
                    hip._check(rc, name)
            if timers is not None:
                e1.record()
                timers.setdefault(name, []).append((e0, e1))

    def load_features(self, feats: Dict[str, torch.Tensor]):
        """Next batch of the same shape into the engine's own feature buffers (own_features=True): ~0.2 ms of device
        copies at cfg2 instead of a new binding and a new graph capture (~6 ms)."""
        if not self.own_features:
            raise RuntimeError("DecodeEngine.load_features needs own_features=True")
        pool = feats["pool_feats"]
        mask = feats["pnt_mask"][:, 1:] if feats["pnt_mask"].shape[1] == pool.shape[1] + 1 else feats["pnt_mask"]
        for dst, src in zip(self.feats, (feats["fc_feats"], feats["conv_feats"], feats["p_conv_feats"], pool, feats["p_pool_feats"])):
            if dst.shape != src.shape:
                raise RuntimeError(f"DecodeEngine.load_features: shape {tuple(src.shape)} != bound {tuple(dst.shape)}")
            dst.copy_(src)
        self.mask.copy_(hip._mask(mask))
        return self

    def bind_features(self, feats: Dict[str, torch.Tensor]):
        """Next batch of the same shape WITHOUT copying it: the C-ABI plan (and this engine) is pointed at the caller's
        feature tensors.  Only for engines that run through the driver without a captured graph (a graph keeps the pointers it
        was captured with: use load_features there)."""
        if self._plan is None or self.graph is not None:
            raise RuntimeError("DecodeEngine.bind_features needs a driver-bound engine without a captured graph")
        pool = feats["pool_feats"]
        mask = feats["pnt_mask"][:, 1:] if feats["pnt_mask"].shape[1] == pool.shape[1] + 1 else feats["pnt_mask"]
        new = (feats["fc_feats"], feats["conv_feats"], feats["p_conv_feats"], pool, feats["p_pool_feats"])
        for dst, src, name in zip(self.feats, new, ("fc_feats", "conv_feats", "p_conv_feats", "pool_feats", "p_pool_feats")):
            if dst.shape != src.shape:
                raise RuntimeError(f"DecodeEngine.bind_features: shape {tuple(src.shape)} != bound {tuple(dst.shape)}")
            hip._dev(src, name=name)
        self.mask = hip._mask(mask)
        self.feats = new
        self.fc = new[0]
        hip._check(hip.lib().cvc_decode_plan_set_features(self._plan, *(t.data_ptr() for t in new), self.mask.data_ptr()),
                   "cvc_decode_plan_set_features")
        self._launches = None                       # the Python launch list holds the old pointers: rebuilt on demand
        return self

    def _python_launches(self):
        if self._launches is None:
            self._keep = []
            self._launches = self._build_packed() if self.packed else (self._build_tile() if self.tile else self._build())
        return self._launches

    def run_timed(self):
        """One eager decode with a HIP-event pair around every launch.  Returns name -> list of ms."""
        timers = {}
        self._reset()
        self._run_launches(timers)
        torch.cuda.synchronize()
        return {k: [a.elapsed_time(b) for a, b in v] for k, v in timers.items()}

    def _run_once(self):
        """One decode on the current stream: through the C-ABI driver when bound (it resets its state itself), else the
        Python launch list."""
        if self._plan is not None:
            self._run_driver()
        else:
            self._reset()
            self._run_launches()
        if self.beam > 1:                                  # rank-0 hypothesis: one launch (was ~60 indexing launches per decode)
            hip._check(hip.lib().cvc_beam_backtrack(self.words[1:].data_ptr(), self.parent.data_ptr(), self.att_steps.data_ptr(),
                                                    self.B, self.beam, self.T, self.N, self.bt_seq.data_ptr(),
                                                    self.bt_att.data_ptr(), hip._stream()), "cvc_beam_backtrack")

    def capture(self):
        """Capture the T-step loop into a HIP graph (launch-bound inner loop -> one replay)."""
        if self.lang_ksx and not self._ksx_checked:
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                self._run_once()
            torch.cuda.current_stream().wait_stream(s)
            self.check_ksx()
        key = (self.packed, self.tile, self.beam > 1)
        if key not in DecodeEngine._warm:                 # first capture of this path in the process: run once outside capture
            s = torch.cuda.Stream()                       # (module load, lazy init); later engines skip the extra decode
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                self._run_once()
            torch.cuda.current_stream().wait_stream(s)
            DecodeEngine._warm.add(key)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode=_capture_mode()):
            self._run_once()
        self.graph = g
        return self

    def run(self):
        """One full T-step decode.  Returns (seq [B,T] int64, att2_weights [B,T,N]) -- views of
        engine-owned buffers (clone to keep across runs)."""
        if self.graph is not None:
            self.graph.replay()
        else:
            self._run_once()
            if self.lang_ksx and not self._ksx_checked and self.check_ksx():
                self._run_once()                           # the fallback's results
        if self.beam == 1:
            return self.words[1:].t(), self.att_steps.permute(1, 0, 2)
        return self._backtrack()

    def _backtrack(self):
        """Rank-0 hypothesis of every clip (cvc_beam_backtrack, enqueued with the decode) and the final beam scores."""
        return self.bt_seq, self.bt_att, self.score[self.T & 1].view(self.B, self.beam)

    def _backtrack_host(self):
        """The same by indexing on the host side of torch (kept as the cross-check of the kernel in the tests)."""
        B, beam, T, N = self.B, self.beam, self.T, self.N
        words = self.words[1:].view(T, B, beam)
        parent = self.parent.view(T, B, beam)
        att = self.att_steps.view(T, B, beam, N)
        k = torch.zeros(B, dtype=torch.int64, device=words.device)
        ar = torch.arange(B, device=words.device)
        seq, atts = [], []
        for t in range(T - 1, -1, -1):
            seq.append(words[t, ar, k])
            k_parent = parent[t, ar, k]
            atts.append(att[t, ar, k_parent])     # attention was computed for the parent row at step t
            k = k_parent
        seq.reverse()
        atts.reverse()
        final_scores = self.score[self.T & 1].view(B, beam)
        return torch.stack(seq, 1), torch.stack(atts, 1), final_scores

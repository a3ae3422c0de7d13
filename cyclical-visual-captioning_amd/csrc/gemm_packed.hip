// Packed GEMM of the decode engine (split out of gemm_skinny.hip; the row-major kernels live there).
#include "cvc_common.h"
#include "gemm_split.h"
#include "gsk.h"
#include "dropout_rng.h"

// ==========================================================================================
// Packed path for the decode engine: both MFMA operands are stored fragment-native in HBM, so a
// wave loads them straight into VGPRs with fully coalesced dwordx4 loads -- no LDS, no DMA, no
// swizzle -- and the prefetch depth is bounded by the 512-entry register file (one wave per SIMD)
// instead of LDS: DEPTH chunks (DEPTH x 4 KB of weights per wave) in flight.
//
// Measured motivation (cfg2 lang-LSTM): with the LDS ring of gemm_skinny.hip, the memory side alone needs 53 us
// (3.8 TB/s: only 2 x 4 KB of weights in flight per wave, because activations take 2/3 of the
// ring and vmcnt retires in order), the MFMA side alone 57 us, together 71 us.
//
// Layouts ("quad" = 4 consecutive k):
//   weights  Wp[blk][quad][32][4]   blk = 32 output rows (LSTM: the 4 gates x 8 hidden units of
//            workgroup blk, i.e. row (i>>3)*R + blk*8 + (i&7)), packed once when the engine binds
//            the checkpoint (cvc/decode.py::pack_weights); the concat over K segments is baked in.
//   activations XQ[quad][64][4]     written in this form by the producers (this kernel's own
//            epilogue, attn_wsum, top2_final), 64 = padded batch rows.
// Lane l (i = l & 31, kh = l >> 5) of a wave handling chunk c (32 k = 8 quads) loads quads
// 8c + 4kh + {0..3}: each half-wave reads 512 contiguous bytes per instruction.
// ==========================================================================================
struct PackedArgs {
    const float* wp;          // packed weights of this GEMM
    const float* xq;          // packed activations, first quad of this GEMM's K range
    int nquad;                // K / 4 (multiple of 8)
    int M, Nout, R;
    const float* bias;        // linear: [Nout]; lstm: b_ih [4R] (nullable)
    const float* bias2;       // lstm: b_hh (nullable)
    const float* gate_bias;   // lstm: [M, 4R] row-major (nullable)
    const float* c_prev_q;    // lstm: cell state, quad layout [R/4][64][4]
    float* c_out_q;
    float* h_dst1_q;          // lstm: h' in quad layout at (quad offset baked into the pointer); nullable
    float* h_dst2_q;
    float* y;                 // linear: row-major [M, ldy] (+ split slices), nullable
    int ldy;
    int ksplit;
    long long split_stride;
    float* top2_part;
    // lstm, training form (cvc_packed_lstm_train_fwd): row-major state and the activated gates autograd keeps
    const float* c_prev_rm;   // [M, R]; used instead of c_prev_q when set
    float* h_rm;              // [M, R]
    float* c_rm;              // [M, R]
    float* gates_rm;          // [M, 4R] activated (i, f, g, o)
    float* h_rm2;             // further copies of h' [M, R] (one tensor per consumer: autograd then has nothing to accumulate), nullable
    float* h_rm3;
    // GRU step (cvc_gru_seq_fwd): blockIdx.y = direction d; R = hidden size H; block rows = (r, z, n, zero) x 8 hidden units
    long long gru_w_stride;   // floats between the directions' weight packs
    long long gru_h_stride;   // floats between the directions' hidden states (quad layout)
    const float* gru_gi[2];   // this step's input projections of direction d: row m at + m * gru_gi_ld, columns [3H] (r, z, n)
    long long gru_gi_ld;
    float* gru_y[2];          // this step's output rows of direction d: row m at + m * gru_y_ld, H columns
    long long gru_y_ld;
    float* gru_gates[2];      // training form (nullable): (r, z, n, W_hn h + b_hn) of this step, row m at + m * gru_g_ld, columns [4][H]
    long long gru_g_ld;
    // embedding-gate table (cvc_packed_lstm_embgate_fwd): the embedded word's share of the gates is a row gather, not a GEMM
    const float* emb_gate;    // [V][4R] = relu(Emb[v]) x W_ih[:, emb columns]^T, checkpoint gate order (gate * R + unit), or null
    const int64_t* word;      // [M] the word of every batch row
    // word selection fused into the vocabulary projection (cvc_packed_linear_select_fwd): the last workgroup to arrive merges the
    // per-block top-2 records of all rows
    unsigned* sel_counter;    // one word of device memory, zero between launches
    int64_t* sel_word; int sel_word_stride; float* sel_logprob; int sel_unk;
    int w_cached;             // lstm decode form: 1 = the gate weights keep the default cache policy (Infinity-Cache resident by plan)
    DropSpec h3_drop;         // training form: h_rm3 receives nn.Dropout(h') with the counter-based mask of element m * R + j
    long long wstride;        // floats between consecutive 32-row blocks of wp (0: dense, nquad * 128)
    GskSegs early;            // SLAB form (cvc_packed_lstm_late_fwd): partial tiles of the K range a stream-K launch already covered
};

#ifdef CVC_TS
// diagnostic build (-DCVC_TS): per-workgroup timestamps (100 MHz constant clock) of the LSTM gate GEMM's phases, last launch wins;
// slot 0: K >= 6144 (language cell), slot 1: shorter K.  [slot][workgroup][wave][4] = entry, first chunk multiplied, K loop done, end
__device__ unsigned long long cvc_ts_buf[2 * 256 * 8 * 4];
extern "C" int cvc_debug_ts_read(unsigned long long* dst) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(cvc_ts_buf), sizeof(cvc_ts_buf)) == hipSuccess ? 0 : -1;
}
#define CVC_TS_MARK(k) do { if (LSTM && !GRU && lane == 0 && blockIdx.x < 256) ts_[k] = wall_clock64(); } while (0)
#else
#define CVC_TS_MARK(k) do {} while (0)
#endif

#ifndef CVC_LIN_W_NT
#define CVC_LIN_W_NT 0      // 1: also stream the linear layers' weights non-temporally (A/B switch)
#endif

template <int MT>
struct PFrag {
    f32x4 w[4];
    f32x4 x[MT][4];
#if defined(CVC_PABL) && CVC_PABL == 6
    f32x4 x2[MT][2];     // ablation: the extra 50 % of activation bytes a pre-split (3 x bf16) operand would bring in
#endif
};

// NW waves split K (chunk c goes to wave c % NW).  NW = 8 puts two waves on every SIMD, each with a shallower
// ring: while one waits on HBM the other multiplies -- the split-product variant needs that, its compute per
// chunk being too short for one wave's ring to cover the memory latency.
template <int NW>
__device__ __forceinline__ float sum_partials(const float* red, int row, int ldm, int m) {
    float v = (red[(0 * 32 + row) * ldm + m] + red[(1 * 32 + row) * ldm + m]) +
              (red[(2 * 32 + row) * ldm + m] + red[(3 * 32 + row) * ldm + m]);
    if constexpr (NW == 8)
        v += (red[(4 * 32 + row) * ldm + m] + red[(5 * 32 + row) * ldm + m]) +
             (red[(6 * 32 + row) * ldm + m] + red[(7 * 32 + row) * ldm + m]);
    return v;
}

// running top-2 / log-sum-exp of word selection (captioner.py:415-422, 437), merged record by record in a fixed order
struct SelState { float t1, t2, gm, gs; int j1, j2; };
__device__ __forceinline__ SelState sel_init() { return SelState{-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), 0.f, 0x7fffffff, 0x7fffffff}; }
__device__ __forceinline__ bool sel_better(float va, int ia, float vb, int ib) { return (va > vb) | ((va == vb) & (ia < ib)); }
__device__ __forceinline__ SelState sel_merge(SelState s, float u1, int k1, float u2, int k2, float um, float us) {
    if (sel_better(u1, k1, s.t1, s.j1)) {
        if (sel_better(s.t1, s.j1, u2, k2)) { s.t2 = s.t1; s.j2 = s.j1; } else { s.t2 = u2; s.j2 = k2; }
        s.t1 = u1; s.j1 = k1;
    } else if (sel_better(u1, k1, s.t2, s.j2)) { s.t2 = u1; s.j2 = k1; }
    const float nm = fmaxf(s.gm, um);
    s.gs = (nm == -__builtin_inff()) ? 0.f : s.gs * __expf(s.gm - nm) + us * __expf(um - nm);
    s.gm = nm;
    return s;
}

// NB = 32-row weight blocks per workgroup (LSTM decode form only; selectable experiment, see cvc_packed_lstm_wg_blocks).
// NB = 2: waves w and w + NW/2 take the SAME K chunks for two different blocks, on the same SIMD and in lock step, so the
// second request for an activation line is served by the CU's L1 (or merged with the pending miss): L2 activation reads per
// launch halve (402 -> 201 MB for the lang cell), with half as many workgroups.
// SLAB (LSTM decode form, 8 waves): this launch covers only the LATE K range of the cell; the partial tiles of the rest, produced
// earlier by the grouped stream-K kernel (gemm_gsk.hip), are summed in segment order and join the cross-wave reduction as a
// ninth partial.
// WC: the LSTM gate weights keep the default cache policy instead of streaming non-temporally (experiment: cvc_packed_lstm_cached_weights)
template <int MT, bool LSTM, int DEPTH, bool SPLIT, int NW, bool GRU = false, int NB = 1, bool SLAB = false, bool WC = false>
__global__ __launch_bounds__(NW * 64) void skinny_gemm_packed_kernel(PackedArgs a) {
    static_assert(!GRU || LSTM, "the GRU step shares the LSTM form's work split");
    static_assert(!SLAB || (LSTM && !GRU && NB == 1 && NW == 8), "slab sum: LSTM decode form, 8 waves, one block per workgroup");
    static_assert(NB == 1 || (LSTM && !GRU && NW == 8 && NB == 2), "two blocks per workgroup: LSTM form, 8 waves");
    constexpr int NWK = NW / NB;                               // waves that split K for one block
    if constexpr (GRU) {                                       // direction of this workgroup
        a.wp += (size_t)blockIdx.y * a.gru_w_stride;
        a.xq += (size_t)blockIdx.y * a.gru_h_stride;
        a.h_dst1_q += (size_t)blockIdx.y * a.gru_h_stride;
        a.bias += (size_t)blockIdx.y * 3 * a.R;
        a.bias2 += (size_t)blockIdx.y * 3 * a.R;
    }
    static_assert(NW == 4 || NW == 8, "4 or 8 waves per workgroup");
    constexpr int LDM = MT * 32 + 1;
    __shared__ float red[(NW + (SLAB ? 1 : 0)) * 32 * LDM + NW * 64 * 6];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, kh = lane >> 5;
    const int M = a.M, R = a.R;
#ifdef CVC_TS
    unsigned long long ts_[4] = {0, 0, 0, 0};
#endif
    CVC_TS_MARK(0);

    int nchunk = a.nquad >> 3, c0 = 0;
    if (!LSTM && a.ksplit > 1) {                              // K slice of this workgroup (whole chunks)
        const int lo = nchunk * (int)blockIdx.y / a.ksplit, hi = nchunk * ((int)blockIdx.y + 1) / a.ksplit;
        c0 = lo;
        nchunk = hi - lo;
    }
    const int wblk = NB == 1 ? 0 : wave / NWK, kw = NB == 1 ? wave : wave % NWK;       // this wave's block and K slot
    const int n_my = nchunk > kw ? (nchunk - kw + NWK - 1) / NWK : 0;     // chunks c0 + kw + NWK*j
    // per-lane bases: quad q of this block lives at wp + ((blk * nquad + q) * 32 + i) * 4
    const float* wl = a.wp + (size_t)((int)blockIdx.x * NB + wblk) * a.wstride + (size_t)i * 4 + (size_t)(c0 + kw) * 8 * 128 + kh * 4 * 128;
    const float* xl = a.xq + (size_t)i * 4 + (size_t)(c0 + kw) * 8 * 256 + kh * 4 * 256;
    constexpr size_t WSTEP = (size_t)NWK * 8 * 128, XSTEP = (size_t)NWK * 8 * 256;   // floats per wave-chunk step
#ifndef CVC_ROT_MUL
#define CVC_ROT_MUL 5
#endif
    const int rot = n_my > 0 ? (int)((blockIdx.x * CVC_ROT_MUL) % (unsigned)n_my) : 0;

    auto load = [&](PFrag<MT>& f, int j) __attribute__((always_inline)) {
#if defined(CVC_PABL) && CVC_PABL == 1
        if (j > 0) { asm volatile("" : "+v"(f.w[0])); return; }     // ablation: only the first chunk is ever loaded (MFMA side only)
#endif
        // every workgroup walks K from a different starting chunk: all 256 of them read the SAME activation lines,
        // and in lock step they would queue on the same L2 channels (the order of a wave's partial sums changes
        // with the block index, the result of a given block is still deterministic)
        int jr = j + rot;
        jr = jr >= n_my ? jr - n_my : jr;
        const float* w = wl + (size_t)jr * WSTEP;
#if defined(CVC_PABL) && CVC_PABL == 4
        const float* x = a.xq + (size_t)i * 4 + kh * 4 * 256;     // ablation: every wave re-reads ONE activation chunk (L1 hits)
#else
        const float* x = xl + (size_t)jr * XSTEP;
#endif
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            // gate weights (369 MB per step) are streamed; the small linear layers' weights (vocabulary head, h2attn:
            // 49 MB) keep the default policy so that they can stay in the Infinity Cache between steps
            if constexpr (((LSTM && !GRU) || CVC_LIN_W_NT) && !WC) f.w[q] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(w + q * 128));
            else f.w[q] = ld4(w + q * 128);
#if defined(CVC_PABL) && CVC_PABL == 3
            if (j > 0) continue;                                     // ablation: stream the weights only
#endif
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) f.x[mt][q] = ld4(x + q * 256 + mt * 128);
        }
#if defined(CVC_PABL) && CVC_PABL == 6
        {
            int ja = jr + (n_my >> 1);
            ja = ja >= n_my ? ja - n_my : ja;
            const float* xa = xl + (size_t)ja * XSTEP;                // lines this workgroup touched half a loop ago: L2 hits
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) f.x2[mt][s2] = ld4(xa + (s2 + 2) * 256 + mt * 128 + 64);
        }
#endif
    };

    // embedding-gate form: the table row of this thread's epilogue work item (batch row tid & 63, hidden quad (tid >> 6) & 1) is
    // requested before the K loop -- word, then 4 x 16 bytes of a row that nobody else touches (HBM latency): left to the epilogue
    // it was 4 us of exposed round trips per launch
    f32x4 eadd4[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    // (Folding the previous step's word selection into this launch -- every workgroup merging the 157 x 64 top-2 records itself
    // -- was built and measured: +8 us on this kernel against 6.5 + 1.5 us for the selection launch it removes; not kept.)
    if constexpr (LSTM && !GRU) {
        if (a.emb_gate != nullptr && tid < NB * 2 * 64) {
            const int em0 = tid & 63;
            const long long eword = a.word[em0 < a.M ? em0 : a.M - 1];
            // table in checkpoint order [V][4R]: gate g of hidden unit u at g * R + u
            const float* trow = a.emb_gate + (size_t)eword * 4 * a.R + (size_t)((int)blockIdx.x * NB + (tid >> 7)) * 8 + ((tid >> 6) & 1) * 4;
#pragma unroll
            for (int g = 0; g < 4; ++g) eadd4[g] = ld4(trow + (size_t)g * a.R);
        }
    }

    f32x16 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;

    auto mma = [&](const PFrag<MT>& f) __attribute__((always_inline)) {
#if defined(CVC_PABL) && CVC_PABL >= 2 && CVC_PABL <= 4
#pragma unroll
        for (int q = 0; q < 4; ++q) {                               // ablation: memory side only, keep the loads live
            asm volatile("" ::"v"(f.w[q]));
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) asm volatile("" ::"v"(f.x[mt][q]));
        }
        return;
#endif
        if constexpr (SPLIT) {
            // the lane half's 4 quads = 16 k-slots = two K=16 steps (quads 2s, 2s+1); W and X use the same slot map
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const Split3 W = split8(f.w[2 * s2], f.w[2 * s2 + 1]);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
#if defined(CVC_PABL) && (CVC_PABL == 5 || CVC_PABL == 6)
                    // ablation: activations arrive pre-split (no VALU work on them; 5: same bytes, 6: 1.5 x the bytes) -- timing only
                    Split3 X;
                    X.hi = __builtin_bit_cast(u32x4, f.x[mt][2 * s2]); X.mid = __builtin_bit_cast(u32x4, f.x[mt][2 * s2 + 1]);
#if CVC_PABL == 6
                    X.lo = __builtin_bit_cast(u32x4, f.x2[mt][s2]);
#else
                    X.lo = X.hi;
#endif
#else
                    const Split3 X = split8(f.x[mt][2 * s2], f.x[mt][2 * s2 + 1]);
#endif
                    acc[mt] = mfma_bf16(W.mid, X.mid, acc[mt]);
                    acc[mt] = mfma_bf16(W.lo, X.hi, acc[mt]);
                    acc[mt] = mfma_bf16(W.hi, X.lo, acc[mt]);
                    acc[mt] = mfma_bf16(W.mid, X.hi, acc[mt]);
                    acc[mt] = mfma_bf16(W.hi, X.mid, acc[mt]);
                    acc[mt] = mfma_bf16(W.hi, X.hi, acc[mt]);
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.w[q][e], f.x[mt][q][e], acc[mt], 0, 0, 0);
        }
    };

    // register ring, DEPTH chunks in flight; fully unrolled so that every fragment is statically named.
    // The steady-state loop issues its loads UNCONDITIONALLY: with a conditional load on any path the
    // compiler's s_waitcnt insertion has to assume the fewest loads outstanding and degrades the
    // counted vmcnt(N) of the oldest slot to a near-full drain.
    PFrag<MT> ring[DEPTH];
    if (n_my >= DEPTH) {
        // slots 0 .. DEPTH-2 are filled up front; every step multiplies slot s while it (re)fills the slot
        // consumed one step earlier, the 12 loads spread between the 32 MFMAs of the step
#pragma unroll
        for (int s = 0; s < DEPTH - 1; ++s) load(ring[s], s);
        int j = 0;
        for (; j + 2 * DEPTH - 1 <= n_my; j += DEPTH) {
#pragma unroll
            for (int s = 0; s < DEPTH; ++s) {
                load(ring[(s + DEPTH - 1) % DEPTH], j + s + DEPTH - 1);
                mma(ring[s]);
#ifdef CVC_TS
                if (s == 0 && j == 0) CVC_TS_MARK(1);
#endif
                if constexpr (SPLIT) {
#pragma unroll
                    for (int g = 0; g < 4 + 4 * MT; ++g) {
                        __builtin_amdgcn_sched_group_barrier(0x002, 18, 0);  // VALU (operand split + addresses)
                        __builtin_amdgcn_sched_group_barrier(0x008, MT, 0);  // MFMA (12 * MT per slot)
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // VMEM read
                    }
                } else {
#pragma unroll
                    for (int g = 0; g < 4 + 4 * MT; ++g) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);   // MFMA
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // VMEM read
                        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);   // VALU (addresses)
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // drain: chunks j .. n_my-1 (fewer than 2*DEPTH-1 left); slots 0..DEPTH-2 hold chunks j..j+DEPTH-2
#pragma unroll
        for (int s = 0; s < DEPTH; ++s) {
            if (j + s + DEPTH - 1 < n_my) load(ring[(s + DEPTH - 1) % DEPTH], j + s + DEPTH - 1);
            if (j + s < n_my) mma(ring[s]);
        }
#pragma unroll
        for (int s = 0; s < DEPTH - 1; ++s)
            if (j + DEPTH + s < n_my) mma(ring[s]);
    } else {
        for (int j = 0; j < n_my; ++j) {                       // short K: no pipeline
            load(ring[0], j);
            mma(ring[0]);
        }
    }

    CVC_TS_MARK(2);
    // LSTM: the cell update's global operands (one work item per thread: batch row m, 4 hidden units) are requested
    // BEFORE the cross-wave LDS stage, so that their latency runs under it
    const int em = tid & 63, eqd = (tid >> 6) & 1, eb = tid >> 7;       // (batch row, hidden quad, block of the workgroup)
    const bool ework = LSTM && tid < NB * 2 * 64 && em < M && em < MT * 32;
    const int ejq = ((int)blockIdx.x * NB + eb) * 8 + eqd * 4;         // first of this thread's 4 hidden units
    const float* ered = red + (NB == 1 ? 0 : eb * NWK * 32 * LDM);     // the partial tiles of this thread's block
    const size_t eqoff = ((size_t)(ejq / 4) * 64 + em) * 4;
    f32x4 ecp = {0, 0, 0, 0}, eadd[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    f32x4 eadd2[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}}, eadd3[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};

    if (GRU && ework) {
        // eadd: r, z: x-projection + both biases; n: x-projection + b_in; [3]: b_hn (multiplied by r with the h-projection)
        // (selects, not a runtime index: indexing the kernel-argument arrays made hipcc copy the whole struct to scratch, 344 B per lane)
        const float* gi = (blockIdx.y == 0 ? a.gru_gi[0] : a.gru_gi[1]) + (size_t)em * a.gru_gi_ld + ejq;
        ecp = ld4(a.xq + eqoff);                                      // h_prev of these 4 hidden units
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            eadd[g] = ld4(gi + g * R) + ld4(a.bias + g * R + ejq);
            if (g < 2) eadd[g] += ld4(a.bias2 + g * R + ejq);
        }
        eadd[3] = ld4(a.bias2 + 2 * R + ejq);
    } else if (ework) {
        // Requests only -- the sums are taken after the cross-wave stage.  With "eadd[g] += load" under each (uniform) branch
        // every addition waited for its load: up to eight dependent L2 round trips on the critical path of waves 0 and 1.
        ecp = a.c_prev_rm != nullptr ? ld4(a.c_prev_rm + (size_t)em * R + ejq) : ld4(a.c_prev_q + eqoff);
        if (a.bias != nullptr) {
#pragma unroll
            for (int g = 0; g < 4; ++g) eadd[g] = ld4(a.bias + g * R + ejq);
        }
        if (a.bias2 != nullptr) {
#pragma unroll
            for (int g = 0; g < 4; ++g) eadd2[g] = ld4(a.bias2 + g * R + ejq);
        }
        if (a.gate_bias != nullptr) {
#pragma unroll
            for (int g = 0; g < 4; ++g) eadd3[g] = ld4(a.gate_bias + (size_t)em * 4 * R + g * R + ejq);
        }
    }

    // SLAB: this block's early partial tiles, 16 bytes per thread and segment (row m = tid >> 3, gate rows 4 (tid & 7) ..+3);
    // unconditional loads of a clamped segment index, summed in segment order below
    constexpr int SLAB_MAXS = 10;
    f32x4 sv[SLAB ? SLAB_MAXS : 1];
    int s_nseg = 0;
    const float* s_p0 = nullptr;
    if constexpr (SLAB) {
        const int tile = (int)blockIdx.x >> 3, j = (int)blockIdx.x & 7;
        s_nseg = gsk_nseg(a.early, tile);
        s_p0 = gsk_part(a.early, tile, 0, j) + (size_t)tid * 4;          // float4 #tid of the [64 rows][32 gate rows] tile: coalesced
#pragma unroll
        for (int s = 0; s < SLAB_MAXS; ++s) sv[s] = ld4(s_p0 + (size_t)(s < s_nseg ? s : s_nseg - 1) * (8 * 2048));
    }

    // linear variant with the fused top-2 epilogue: the biases of the columns this wave scans, requested here too
    constexpr int ECPW = 32 / NW;
    float ebias[ECPW];
#pragma unroll
    for (int c = 0; c < ECPW; ++c) {
        const int n = (int)blockIdx.x * 32 + wave * ECPW + c;
        ebias[c] = (!LSTM && a.top2_part != nullptr && a.bias != nullptr) ? a.bias[n < a.Nout ? n : a.Nout - 1] : 0.f;
    }

    // ---- ordered cross-wave reduction (same scheme as combine_and_store)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * kh;
            red[(wave * 32 + row) * LDM + mt * 32 + i] = acc[mt][r];
        }
    if constexpr (SLAB) {
        f32x4 t = sv[0];
#pragma unroll
        for (int s = 1; s < SLAB_MAXS; ++s) t += s < s_nseg ? sv[s] : f32x4{0.f, 0.f, 0.f, 0.f};
        for (int s = SLAB_MAXS; s < s_nseg; ++s) t += ld4(s_p0 + (size_t)s * (8 * 2048));          // (more segments than the unrolled part)
        if ((tid >> 3) < MT * 32) {
#pragma unroll
            for (int e = 0; e < 4; ++e) red[(NW * 32 + (tid & 7) * 4 + e) * LDM + (tid >> 3)] = t[e];
        }
    }
    __syncthreads();

    if constexpr (GRU) {
        if (ework) {
            f32x4 hv, gr, gz, gn, gh;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int jj = eqd * 4 + e;
                float pre[3];
#pragma unroll
                for (int g = 0; g < 3; ++g) pre[g] = sum_partials<NWK>(ered, g * 8 + jj, LDM, em);
                const float rg = fast_sigmoid(pre[0] + eadd[0][e]), zg = fast_sigmoid(pre[1] + eadd[1][e]);
                const float hn = pre[2] + eadd[3][e];
                const float ng = fast_tanh(eadd[2][e] + rg * hn);
                hv[e] = ng + zg * (ecp[e] - ng);                       // (1 - z) n + z h
                gr[e] = rg; gz[e] = zg; gn[e] = ng; gh[e] = hn;
            }
            st4(a.h_dst1_q + eqoff, hv);
            st4((blockIdx.y == 0 ? a.gru_y[0] : a.gru_y[1]) + (size_t)em * a.gru_y_ld + ejq, hv);
            float* gp = blockIdx.y == 0 ? a.gru_gates[0] : a.gru_gates[1];
            if (gp != nullptr) {          // what autograd keeps of the step (cvc_gru_seq_train_fwd)
                gp += (size_t)em * a.gru_g_ld + ejq;
                st4(gp, gr); st4(gp + a.R, gz); st4(gp + 2 * a.R, gn); st4(gp + 3 * a.R, gh);
            }
        }
    } else if (LSTM) {
        // unit u -> (batch row m fastest, quad-of-hidden qd in 0..1): a thread finishes 4 hidden units
        // and stores them as one float4 in quad layout (64 rows x 16 B contiguous per quad)
        if (ework) {
            f32x4 hv, cv, gv[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) eadd[g] = ((eadd[g] + eadd2[g]) + eadd3[g]) + eadd4[g];      // (an absent term is an exact zero)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int jj = eqd * 4 + e;
                float pre[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float sp = sum_partials<NWK>(ered, g * 8 + jj, LDM, em);
                    if constexpr (SLAB) sp += red[(NW * 32 + g * 8 + jj) * LDM + em];
                    pre[g] = sp + eadd[g][e];
                }
                const float ig = fast_sigmoid(pre[0]), fg = fast_sigmoid(pre[1]);
                const float gg = fast_tanh(pre[2]), og = fast_sigmoid(pre[3]);
                const float c2 = fg * ecp[e] + ig * gg;
                cv[e] = c2;
                hv[e] = og * fast_tanh(c2);
                gv[0][e] = ig; gv[1][e] = fg; gv[2][e] = gg; gv[3][e] = og;
            }
            if (a.c_out_q != nullptr) st4(a.c_out_q + eqoff, cv);
            if (a.h_dst1_q != nullptr) st4(a.h_dst1_q + eqoff, hv);
            if (a.h_dst2_q != nullptr) st4(a.h_dst2_q + eqoff, hv);
            if (a.h_rm != nullptr) st4(a.h_rm + (size_t)em * R + ejq, hv);
            if (a.h_rm2 != nullptr) st4(a.h_rm2 + (size_t)em * R + ejq, hv);
            if (a.h_rm3 != nullptr) {
                f32x4 hd = hv;
                if (a.h3_drop.state != nullptr) {                  // decoder_core.py:62, 109: output = dropout(h_lang), fused
                    const uint32_t s0 = a.h3_drop.state[0], s1 = a.h3_drop.state[1], s2 = a.h3_drop.state[2];
                    const uint32_t i0 = (uint32_t)em * (uint32_t)R + (uint32_t)ejq;
#pragma unroll
                    for (int e = 0; e < 4; ++e) hd[e] *= cvc_drop_mult(a.h3_drop, s0, s1, s2, i0 + e);
                }
                st4(a.h_rm3 + (size_t)em * R + ejq, hd);
            }
            if (a.c_rm != nullptr) st4(a.c_rm + (size_t)em * R + ejq, cv);
            if (a.gates_rm != nullptr) {
#pragma unroll
                for (int g = 0; g < 4; ++g) st4(a.gates_rm + (size_t)em * 4 * R + g * R + ejq, gv[g]);
            }
        }
#ifdef CVC_TS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        CVC_TS_MARK(3);
        if (lane == 0 && blockIdx.x < 256) {
            unsigned long long* d = cvc_ts_buf + ((size_t)((a.nquad >= 1536 ? 0 : 1) * 256 + blockIdx.x) * 8 + wave) * 4;
            d[0] = ts_[0]; d[1] = ts_[1]; d[2] = ts_[2]; d[3] = ts_[3];
        }
#endif
    } else {
        const int n0 = blockIdx.x * 32;
        const bool lead = blockIdx.y == 0;
        float* y = a.y != nullptr ? a.y + (long long)blockIdx.y * a.split_stride : nullptr;
        if (y != nullptr) {
            for (int u = tid; u < 32 * MT * 32; u += NW * 64) {
                const int nl = u & 31, m = u >> 5;
                const int n = n0 + nl;
                if (m >= M || n >= a.Nout) continue;
                float v = sum_partials<NW>(red, nl, LDM, m);
                if (lead && a.bias != nullptr) v += a.bias[n];
                y[(size_t)m * a.ldy + n] = v;
            }
        }
        if (a.top2_part != nullptr) {
            float* scratch = red + NW * 32 * LDM;
            float v1 = -__builtin_inff(), v2 = -__builtin_inff(), mx = -__builtin_inff(), se = 0.f;
            int i1 = 0x7fffffff, i2 = 0x7fffffff;
            const int m = lane < MT * 32 ? lane : MT * 32 - 1;
            constexpr int CPW = 32 / NW;                          // columns scanned per wave
#pragma unroll
            for (int c = 0; c < CPW; ++c) {
                const int nl = wave * CPW + c, n = n0 + nl;
                if (n >= a.Nout) break;
                const float v = sum_partials<NW>(red, nl, LDM, m) + ebias[c];
                if (v > v1) { v2 = v1; i2 = i1; v1 = v; i1 = n; }
                else if (v > v2) { v2 = v; i2 = n; }
                const float nm = fmaxf(mx, v);
                se = se * __expf(mx - nm) + __expf(v - nm);
                mx = nm;
            }
            float* r4 = scratch + ((size_t)wave * 64 + lane) * 6;
            r4[0] = v1; r4[1] = __int_as_float(i1); r4[2] = v2; r4[3] = __int_as_float(i2); r4[4] = mx; r4[5] = se;
            __syncthreads();
            if (wave == 0 && lane < M) {
                for (int w = 1; w < NW; ++w) {
                    const float* q4 = scratch + ((size_t)w * 64 + lane) * 6;
                    const float u1 = q4[0], u2 = q4[2];
                    const int k1 = __float_as_int(q4[1]), k2 = __float_as_int(q4[3]);
                    if (u1 > v1) { if (v1 >= u2) { v2 = v1; i2 = i1; } else { v2 = u2; i2 = k2; } v1 = u1; i1 = k1; }
                    else if (u1 > v2) { v2 = u1; i2 = k1; }
                    const float nm = fmaxf(mx, q4[4]);
                    se = (nm == -__builtin_inff()) ? 0.f : se * __expf(mx - nm) + q4[5] * __expf(q4[4] - nm);
                    mx = nm;
                }
                float* rec = a.top2_part + ((size_t)blockIdx.x * 64 + lane) * 6;
                if (a.sel_counter != nullptr) {
                    // write-through (sc1) stores: the merging workgroup may run on another XCD (cdna guide, Guideline 16 R1)
                    const float r6[6] = {v1, __int_as_float(i1), v2, __int_as_float(i2), mx, se};
#pragma unroll
                    for (int k = 0; k < 6; ++k) __hip_atomic_store(rec + k, r6[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    rec[0] = v1; rec[1] = __int_as_float(i1); rec[2] = v2; rec[3] = __int_as_float(i2); rec[4] = mx; rec[5] = se;
                }
            }
            if (a.sel_counter != nullptr) {
                // ---- word selection (captioner.py:415-422, 437) by the LAST workgroup to arrive: every record it reads was written
                // in this launch and never read before in it, so no cache can hold a stale copy; the counter goes back to zero
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                int* flag = reinterpret_cast<int*>(scratch);
                if (tid == 0) {
                    const unsigned old = __hip_atomic_fetch_add(a.sel_counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const bool last = old == gridDim.x - 1;
                    if (last) {
                        __hip_atomic_store(a.sel_counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    }
                    flag[0] = last ? 1 : 0;
                }
                __syncthreads();
                if (flag[0] == 0) return;
                __syncthreads();                                          // (flag read by everyone before scratch is reused)
                // thread (row m = lane, slice w = wave): records w, w + NW, ... merged in index order; then the NW slices per row
                const int nb = (int)gridDim.x;
                SelState ss = sel_init();
                const int mrow = lane < M ? lane : M - 1;
                // the records were stored write-through: every read is a round trip to memory, so a batch of them is requested
                // before the first merge (unconditional loads of a clamped index; one record at a time this tail took 20 us)
                constexpr int RB = 8;
                for (int b0 = wave; b0 < nb; b0 += NW * RB) {
                    f32x2 r0[RB], r1[RB], r2[RB];
#pragma unroll
                    for (int k = 0; k < RB; ++k) {
                        const int b = b0 + k * NW;
                        const f32x2* q2 = reinterpret_cast<const f32x2*>(a.top2_part + ((size_t)(b < nb ? b : nb - 1) * 64 + mrow) * 6);
                        r0[k] = q2[0]; r1[k] = q2[1]; r2[k] = q2[2];
                    }
#pragma unroll
                    for (int k = 0; k < RB; ++k)
                        if (b0 + k * NW < nb) ss = sel_merge(ss, r0[k].x, __float_as_int(r0[k].y), r1[k].x, __float_as_int(r1[k].y), r2[k].x, r2[k].y);
                }
                float* r8 = scratch + ((size_t)wave * 64 + lane) * 6;
                r8[0] = ss.t1; r8[1] = __int_as_float(ss.j1); r8[2] = ss.t2; r8[3] = __int_as_float(ss.j2); r8[4] = ss.gm; r8[5] = ss.gs;
                __syncthreads();
                if (wave == 0 && lane < M) {
                    for (int w = 1; w < NW; ++w) {
                        const float* q4 = scratch + ((size_t)w * 64 + lane) * 6;
                        ss = sel_merge(ss, q4[0], __float_as_int(q4[1]), q4[2], __float_as_int(q4[3]), q4[4], q4[5]);
                    }
                    const bool use2 = (ss.j1 == a.sel_unk) && ss.j2 != 0x7fffffff;          // captioner.py:417-421
                    int wsel = use2 ? ss.j2 : ss.j1;
                    if (wsel == 0x7fffffff || wsel < 0) wsel = 0;                      // all-NaN logits (see top2_final_kernel)
                    a.sel_word[(size_t)lane * a.sel_word_stride] = wsel;
                    if (a.sel_logprob != nullptr) a.sel_logprob[lane] = (use2 ? ss.t2 : ss.t1) - (ss.gm + __logf(ss.gs));
                }
            }
        }
    }
}

#ifndef CVC_PACKED_DEPTH
#define CVC_PACKED_DEPTH 4
#endif
#ifndef CVC_PACKED_DEPTH8
#define CVC_PACKED_DEPTH8 3
#endif

static int cvc_packed_lstm_blocks = 1;
// A/B + test hook: weight blocks per workgroup of the decode LSTM gate GEMM (1 = default, or 2); returns the previous setting,
// < 1 queries.  Measured at cfg2: two blocks per workgroup 85.6 / 74.1 us (lang / att) against 53.3 / 45.8 -- the L2 activation
// reads do halve, but with 128 workgroups every CU does twice the operand splitting (44 us of VALU work per SIMD) and twice the
// MFMAs (35 us) while half the chip idles: compute-bound.
extern "C" int cvc_packed_lstm_wg_blocks(int n) {
    const int prev = cvc_packed_lstm_blocks;
    if (n >= 1) cvc_packed_lstm_blocks = n >= 2 ? 2 : 1;
    return prev;
}

template <bool LSTM>
static int launch_packed(const PackedArgs& a_in, int blocks, hipStream_t st) {
    PackedArgs a = a_in;
    if (a.M < 1 || a.M > 64 || (a.nquad & 7) || a.nquad < 8) return CVC_E_BADARG;
    if (a.wstride == 0) a.wstride = (long long)a.nquad * 128;
    const dim3 grid(blocks, LSTM || a.ksplit < 1 ? 1 : a.ksplit);
    if constexpr (LSTM) {
        // decode form, 64-row workgroups (two blocks each): halves the L2 activation reads
        if (cvc_gemm_split_mode == 2 && cvc_packed_lstm_blocks == 2 && (blocks & 1) == 0 && a.M > 32 && a.h_rm == nullptr && a.c_prev_rm == nullptr) {
            hipLaunchKernelGGL((skinny_gemm_packed_kernel<2, true, CVC_PACKED_DEPTH8, true, 8, false, 2>), dim3(blocks / 2), dim3(512), 0, st, a);
            return cvc_launch_status();
        }
    }
    if constexpr (LSTM) {
        // (gate weights with the default cache policy: cvc_packed_lstm_embgate_ex_fwd's w_cached)
        // (measured alternatives at cfg2: the language cell's 201 MB instead -- its launch 54.2 -> 48.8 us, the attention cell's back
        // to 40.5: 323-325 k against 326 k; both matrices: over the cache's size, slower than none)
        if (cvc_gemm_split_mode == 2 && a.w_cached) {
            if (a.M > 32) hipLaunchKernelGGL((skinny_gemm_packed_kernel<2, true, CVC_PACKED_DEPTH8, true, 8, false, 1, false, true>), grid, dim3(512), 0, st, a);
            else hipLaunchKernelGGL((skinny_gemm_packed_kernel<1, true, CVC_PACKED_DEPTH8, true, 8, false, 1, false, true>), grid, dim3(512), 0, st, a);
            return cvc_launch_status();
        }
    }
    if (cvc_gemm_split_mode == 2) {            // split products, 8 waves (2 per SIMD), ring depth CVC_PACKED_DEPTH8
        if (a.M <= 32) hipLaunchKernelGGL((skinny_gemm_packed_kernel<1, LSTM, CVC_PACKED_DEPTH8, true, 8>), grid, dim3(512), 0, st, a);
        else hipLaunchKernelGGL((skinny_gemm_packed_kernel<2, LSTM, CVC_PACKED_DEPTH8, true, 8>), grid, dim3(512), 0, st, a);
    } else if (cvc_gemm_split_mode) {
        if (a.M <= 32) hipLaunchKernelGGL((skinny_gemm_packed_kernel<1, LSTM, CVC_PACKED_DEPTH, true, 4>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((skinny_gemm_packed_kernel<2, LSTM, CVC_PACKED_DEPTH, true, 4>), grid, dim3(256), 0, st, a);
    } else {
        if (a.M <= 32) hipLaunchKernelGGL((skinny_gemm_packed_kernel<1, LSTM, CVC_PACKED_DEPTH, false, 4>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((skinny_gemm_packed_kernel<2, LSTM, CVC_PACKED_DEPTH, false, 4>), grid, dim3(256), 0, st, a);
    }
    return cvc_launch_status();
}

extern "C" int cvc_packed_lstm_fwd(const float* wp, const float* xq, int K, const float* b_ih, const float* b_hh,
                                   const float* gate_bias, const float* c_prev_q, int M, int R, float* h_dst1_q,
                                   float* h_dst2_q, float* c_out_q, cvc_stream_t stream) {
    if (!wp || !xq || !c_prev_q || !c_out_q || (K & 31) || R < 8 || (R & 7)) return CVC_E_BADARG;
    PackedArgs a{};
    a.wp = wp; a.xq = xq; a.nquad = K / 4; a.M = M; a.Nout = 4 * R; a.R = R;
    a.bias = b_ih; a.bias2 = b_hh; a.gate_bias = gate_bias; a.c_prev_q = c_prev_q; a.c_out_q = c_out_q;
    a.h_dst1_q = h_dst1_q; a.h_dst2_q = h_dst2_q; a.ksplit = 1;
    return launch_packed<true>(a, R / 8, (hipStream_t)stream);
}

extern "C" int cvc_packed_lstm_embgate_fwd(const float* wp, const float* xq, int K, const float* b_ih, const float* b_hh,
                                           const float* gate_bias, const float* emb_gate, const int64_t* word,
                                           const float* c_prev_q, int M, int R, float* h_dst1_q, float* h_dst2_q,
                                           float* c_out_q, cvc_stream_t stream) {
    if (!wp || !xq || !c_prev_q || !c_out_q || !emb_gate || !word || (K & 31) || R < 8 || (R & 7)) return CVC_E_BADARG;
    PackedArgs a{};
    a.wp = wp; a.xq = xq; a.nquad = K / 4; a.M = M; a.Nout = 4 * R; a.R = R;
    a.bias = b_ih; a.bias2 = b_hh; a.gate_bias = gate_bias; a.c_prev_q = c_prev_q; a.c_out_q = c_out_q;
    a.h_dst1_q = h_dst1_q; a.h_dst2_q = h_dst2_q; a.ksplit = 1; a.emb_gate = emb_gate; a.word = word;
    return launch_packed<true>(a, R / 8, (hipStream_t)stream);
}

// ... the general form.  w_cached: the gate weights are read under the default cache policy instead of streamed non-temporally --
// for a gate matrix that the caller's cache plan keeps in the 256 MiB Infinity Cache between steps (cvc.decode.cache_plan: at
// config 2 the attention cell's 134 MB; its launch 40.3 -> 35.4 us, the decode +1.8 %; 64-row split-product form, other shapes
// run the streaming kernel).  w_blk_stride / K: the contraction may stop short of the packed matrix's K (K a multiple of 32,
// w_blk_stride = floats between its 32-row blocks, 0 = dense K / 4 * 128) -- the first decode step multiplies an all-zero
// recurrent state, and an exact zero times a finite weight adds nothing: the driver passes K = 32 there.
extern "C" int cvc_packed_lstm_embgate_ex_fwd(const float* wp, long long w_blk_stride, const float* xq, int K, const float* b_ih,
                                              const float* b_hh, const float* gate_bias, const float* emb_gate, const int64_t* word,
                                              const float* c_prev_q, int M, int R, float* h_dst1_q, float* h_dst2_q,
                                              float* c_out_q, int w_cached, cvc_stream_t stream) {
    if (!wp || !xq || !c_prev_q || !c_out_q || !emb_gate || !word || (K & 31) || K < 32 || R < 8 || (R & 7)) return CVC_E_BADARG;
    if (w_blk_stride != 0 && (w_blk_stride < (long long)(K / 4) * 128 || (w_blk_stride & 3))) return CVC_E_BADARG;
    PackedArgs a{};
    a.wp = wp; a.xq = xq; a.nquad = K / 4; a.M = M; a.Nout = 4 * R; a.R = R; a.wstride = w_blk_stride;
    a.bias = b_ih; a.bias2 = b_hh; a.gate_bias = gate_bias; a.c_prev_q = c_prev_q; a.c_out_q = c_out_q;
    a.h_dst1_q = h_dst1_q; a.h_dst2_q = h_dst2_q; a.ksplit = 1; a.emb_gate = emb_gate; a.word = word; a.w_cached = w_cached ? 1 : 0;
    return launch_packed<true>(a, R / 8, (hipStream_t)stream);
}

extern "C" int cvc_packed_lstm_late_fwd(const float* wp, long long w_blk_stride, const float* xq, int K, const float* b_ih,
                                        const float* b_hh, const float* gate_bias, const float* c_prev_q, int M, int R,
                                        float* h_dst1_q, float* h_dst2_q, float* c_out_q, const cvc_gsk_segs* early,
                                        cvc_stream_t stream) {
    if (!wp || !xq || !c_prev_q || !c_out_q || (K & 31) || K < 32 || R < 8 || (R & 7) || M < 1 || M > 64 ||
        w_blk_stride < (long long)(K / 4) * 128 || (w_blk_stride & 3))
        return CVC_E_BADARG;
    if (cvc_gemm_split_mode != 2) return CVC_E_BADARG;                  // the stream-K schedule is built on the 8-wave split form
    PackedArgs a{};
    a.wp = wp; a.xq = xq; a.nquad = K / 4; a.M = M; a.Nout = 4 * R; a.R = R; a.wstride = w_blk_stride;
    a.bias = b_ih; a.bias2 = b_hh; a.gate_bias = gate_bias; a.c_prev_q = c_prev_q; a.c_out_q = c_out_q;
    a.h_dst1_q = h_dst1_q; a.h_dst2_q = h_dst2_q; a.ksplit = 1;
    const dim3 grid(R / 8);
    hipStream_t st = (hipStream_t)stream;
    if (early != nullptr) {
        if (!early->slab || early->nchunk < 1 || early->U < 1 || early->maxseg < 1 || early->unit0 < 0 || (R & 63)) return CVC_E_BADARG;
        a.early = *early;
        if (M <= 32) hipLaunchKernelGGL((skinny_gemm_packed_kernel<1, true, CVC_PACKED_DEPTH8, true, 8, false, 1, true>), grid, dim3(512), 0, st, a);
        else hipLaunchKernelGGL((skinny_gemm_packed_kernel<2, true, CVC_PACKED_DEPTH8, true, 8, false, 1, true>), grid, dim3(512), 0, st, a);
    } else {
        if (M <= 32) hipLaunchKernelGGL((skinny_gemm_packed_kernel<1, true, CVC_PACKED_DEPTH8, true, 8>), grid, dim3(512), 0, st, a);
        else hipLaunchKernelGGL((skinny_gemm_packed_kernel<2, true, CVC_PACKED_DEPTH8, true, 8>), grid, dim3(512), 0, st, a);
    }
    return cvc_launch_status();
}

static int packed_lstm_train_impl(const float* wp, const float* xq, int K, const float* b_ih, const float* b_hh, const float* gate_pre,
                                  const float* c_prev, int M, int R, float* h_out, float* c_out, float* gates_out, float* h_out2,
                                  float* h_out3, cvc_stream_t stream, DropSpec h3_drop = DropSpec{nullptr, 0, 0, 0.f});

extern "C" int cvc_packed_lstm_train_drop_fwd(const float* wp, const float* xq, int K, const float* b_ih, const float* b_hh,
                                              const float* gate_pre, const float* c_prev, int M, int R, float* h_out, float* c_out,
                                              float* gates_out, float* h_out2, float* h_drop_out, const uint32_t* rng_state,
                                              unsigned site, float p, cvc_stream_t stream) {
    if (!h_drop_out || !rng_state || p < 0.f || p >= 1.f) return CVC_E_BADARG;
    return packed_lstm_train_impl(wp, xq, K, b_ih, b_hh, gate_pre, c_prev, M, R, h_out, c_out, gates_out, h_out2, h_drop_out, stream,
                                  cvc_drop_spec(rng_state, site, p));
}

extern "C" int cvc_packed_lstm_train_fwd(const float* wp, const float* xq, int K, const float* b_ih, const float* b_hh,
                                         const float* c_prev, int M, int R, float* h_out, float* c_out, float* gates_out,
                                         float* h_out2, float* h_out3, cvc_stream_t stream) {
    return packed_lstm_train_impl(wp, xq, K, b_ih, b_hh, nullptr, c_prev, M, R, h_out, c_out, gates_out, h_out2, h_out3, stream);
}

extern "C" int cvc_packed_lstm_train_pre_fwd(const float* wp, const float* xq, int K, const float* b_ih, const float* b_hh,
                                             const float* gate_pre, const float* c_prev, int M, int R, float* h_out, float* c_out,
                                             float* gates_out, float* h_out2, float* h_out3, cvc_stream_t stream) {
    if (!gate_pre) return CVC_E_BADARG;
    return packed_lstm_train_impl(wp, xq, K, b_ih, b_hh, gate_pre, c_prev, M, R, h_out, c_out, gates_out, h_out2, h_out3, stream);
}

static int packed_lstm_train_impl(const float* wp, const float* xq, int K, const float* b_ih, const float* b_hh, const float* gate_pre,
                                  const float* c_prev, int M, int R, float* h_out, float* c_out, float* gates_out, float* h_out2,
                                  float* h_out3, cvc_stream_t stream, DropSpec h3_drop) {
    if (!wp || !xq || !c_prev || !h_out || !c_out || (K & 31) || R < 8 || (R & 7)) return CVC_E_BADARG;
    PackedArgs a{};
    a.gate_bias = gate_pre;
    a.h3_drop = h3_drop;
    a.h_rm2 = h_out2; a.h_rm3 = h_out3;
    a.wp = wp; a.xq = xq; a.nquad = K / 4; a.M = M; a.Nout = 4 * R; a.R = R;
    a.bias = b_ih; a.bias2 = b_hh; a.c_prev_rm = c_prev; a.h_rm = h_out; a.c_rm = c_out; a.gates_rm = gates_out; a.ksplit = 1;
    return launch_packed<true>(a, R / 8, (hipStream_t)stream);
}

// General training form (include/cvc_hip.h, "Training loops driven from C"): row-major state in / out, activated gates, up to
// two plain row-major copies of h', the dropped copy, and the quad destinations of the decode form -- the C-driven training
// loops hand h' to the next GEMMs without a packing launch.
extern "C" int cvc_packed_lstm_step_fwd(const cvc_lstm_step* s, cvc_stream_t stream) {
    if (!s || !s->wp || !s->xq || !s->c_prev || !s->c_out || (s->K & 31) || s->K < 32 || s->R < 8 || (s->R & 7) || s->M < 1 || s->M > 64)
        return CVC_E_BADARG;
    if ((s->row_bias != nullptr) != (s->row_index != nullptr)) return CVC_E_BADARG;
    if (s->p < 0.f || s->p >= 1.f) return CVC_E_BADARG;
    PackedArgs a{};
    a.wp = s->wp; a.xq = s->xq; a.nquad = s->K / 4; a.M = s->M; a.Nout = 4 * s->R; a.R = s->R; a.ksplit = 1;
    a.bias = s->b_ih; a.bias2 = s->b_hh; a.gate_bias = s->gate_pre;
    a.emb_gate = s->row_bias; a.word = s->row_index;            // (the embedding-gate gather: one table row per batch row)
    a.c_prev_rm = s->c_prev; a.c_rm = s->c_out; a.gates_rm = s->gates_out;
    a.h_rm = s->h_out; a.h_rm2 = s->h_out2; a.h_rm3 = s->h_drop_out;
    a.h3_drop = cvc_drop_spec(s->rng_state, s->site, s->p);
    a.h_dst1_q = s->h_dst1_q; a.h_dst2_q = s->h_dst2_q;
    a.w_cached = s->w_cached ? 1 : 0;
    return launch_packed<true>(a, s->R / 8, (hipStream_t)stream);
}

// ---- GRU over a whole sequence (the encoder's frame context, backbone.py:335-338): one launch per time step, both directions
namespace {
__global__ __launch_bounds__(256) void zero_kernel(float* p, long long n) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t < n) p[t] = 0.f;
}
}  // namespace

static int gru_seq_impl(const float* wp, const float* gi, long long gi_ld_m, long long gi_ld_t, const float* b_ih, const float* b_hh, int M,
                        int F, int H, int ndir, float* hq, float* y, long long y_ld_m, long long y_ld_t, float* gates, long long g_ld_m,
                        long long g_ld_t, cvc_stream_t stream);

extern "C" int cvc_gru_seq_fwd(const float* wp, const float* gi, long long gi_ld_m, long long gi_ld_t, const float* b_ih,
                               const float* b_hh, int M, int F, int H, int ndir, float* hq, float* y, long long y_ld_m,
                               long long y_ld_t, cvc_stream_t stream) {
    return gru_seq_impl(wp, gi, gi_ld_m, gi_ld_t, b_ih, b_hh, M, F, H, ndir, hq, y, y_ld_m, y_ld_t, nullptr, 0, 0, stream);
}

// Training form of the per-step recurrence (any H % 8 == 0 -- config 5's encoder width H = 2048 is beyond what the persistent
// form keeps in registers): additionally writes (r, z, n, W_hn h + b_hn) of every step and direction, as
// cvc_gru_seq_persistent_train_fwd does.
extern "C" int cvc_gru_seq_train_fwd(const float* wp, const float* gi, long long gi_ld_m, long long gi_ld_t, const float* b_ih,
                                     const float* b_hh, int M, int F, int H, int ndir, float* hq, float* y, long long y_ld_m,
                                     long long y_ld_t, float* gates, long long g_ld_m, long long g_ld_t, cvc_stream_t stream) {
    if (!gates || (g_ld_m & 3) || (g_ld_t & 3)) return CVC_E_BADARG;
    return gru_seq_impl(wp, gi, gi_ld_m, gi_ld_t, b_ih, b_hh, M, F, H, ndir, hq, y, y_ld_m, y_ld_t, gates, g_ld_m, g_ld_t, stream);
}

static int gru_seq_impl(const float* wp, const float* gi, long long gi_ld_m, long long gi_ld_t, const float* b_ih, const float* b_hh, int M,
                        int F, int H, int ndir, float* hq, float* y, long long y_ld_m, long long y_ld_t, float* gates, long long g_ld_m,
                        long long g_ld_t, cvc_stream_t stream) {
    if (!wp || !gi || !b_ih || !b_hh || !hq || !y || M < 1 || M > 64 || F < 1 || H < 8 || (H & 7) || ndir < 1 || ndir > 2 ||
        (gi_ld_m & 3) || (gi_ld_t & 3) || (y_ld_m & 3) || (y_ld_t & 3))
        return CVC_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    const int Kp = (H + 31) / 32 * 32;                             // contraction length, zero-padded to whole 32-k chunks
    const long long hsz = (long long)Kp * 64;                      // one direction's state in quad layout
    // h0 = 0, and the padding quads of both parities stay zero
    hipLaunchKernelGGL(zero_kernel, dim3((unsigned)((2 * hsz * ndir + 255) / 256)), dim3(256), 0, st, hq, 2 * hsz * ndir);
    PackedArgs a{};
    a.nquad = Kp / 4; a.M = M; a.Nout = 4 * H; a.R = H; a.bias = b_ih; a.bias2 = b_hh; a.ksplit = 1;
    a.wp = wp; a.gru_w_stride = (long long)(H / 8) * (Kp / 4) * 128; a.gru_h_stride = hsz;
    a.gru_gi_ld = gi_ld_m; a.gru_y_ld = y_ld_m; a.gru_g_ld = g_ld_m;
    a.wstride = (long long)a.nquad * 128;
    const dim3 grid(H / 8, ndir);
    for (int s = 0; s < F; ++s) {
        a.xq = hq + (size_t)(s & 1) * hsz * ndir;
        a.h_dst1_q = hq + (size_t)((s + 1) & 1) * hsz * ndir;
        for (int d = 0; d < ndir; ++d) {
            const long long t = d == 0 ? s : F - 1 - s;
            a.gru_gi[d] = gi + t * gi_ld_t + (long long)d * 3 * H;
            a.gru_y[d] = y + t * y_ld_t + (long long)d * H;
            a.gru_gates[d] = gates ? gates + t * g_ld_t + (long long)d * 4 * H : nullptr;
        }
        if (cvc_gemm_split_mode == 2) {
            if (M <= 32) hipLaunchKernelGGL((skinny_gemm_packed_kernel<1, true, CVC_PACKED_DEPTH8, true, 8, true>), grid, dim3(512), 0, st, a);
            else hipLaunchKernelGGL((skinny_gemm_packed_kernel<2, true, CVC_PACKED_DEPTH8, true, 8, true>), grid, dim3(512), 0, st, a);
        } else if (cvc_gemm_split_mode) {
            if (M <= 32) hipLaunchKernelGGL((skinny_gemm_packed_kernel<1, true, CVC_PACKED_DEPTH, true, 4, true>), grid, dim3(256), 0, st, a);
            else hipLaunchKernelGGL((skinny_gemm_packed_kernel<2, true, CVC_PACKED_DEPTH, true, 4, true>), grid, dim3(256), 0, st, a);
        } else {
            if (M <= 32) hipLaunchKernelGGL((skinny_gemm_packed_kernel<1, true, CVC_PACKED_DEPTH, false, 4, true>), grid, dim3(256), 0, st, a);
            else hipLaunchKernelGGL((skinny_gemm_packed_kernel<2, true, CVC_PACKED_DEPTH, false, 4, true>), grid, dim3(256), 0, st, a);
        }
    }
    return cvc_launch_status();
}

// ---- operands of the training form: both are rebuilt from the row-major tensors autograd and the optimizer own
namespace {

struct PackWArgs {
    const float* w[4];                      // up to 4 column ranges [4R, width_s] (pointer at the range's first column), row-major
    long long ld[4];                        // leading dimension of the matrix each range lives in
    int q_end[4];                           // running quad count after each range
    int nseg, R;
    float* wp;                              // [R/8][sum width_s / 4][32][4]
};

// one workgroup: the 32 gate rows of one block x 64 quads, transposed through LDS so that both the reads (1 KB runs of a
// weight row) and the writes (one 32 KB run of the packed block) are contiguous
__global__ __launch_bounds__(256) void pack_lstm_w_kernel(PackWArgs a) {
    __shared__ f32x4 tile[64][33];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int blk = blockIdx.x, q0 = blockIdx.y * 64;
    const int nquad = a.q_end[a.nseg - 1], q = q0 + lane;
    int s = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) s += (i < a.nseg - 1 && q >= a.q_end[i]) ? 1 : 0;
    const int qs = q - (s ? a.q_end[s - 1] : 0);
    const float* src = s == 0 ? a.w[0] : (s == 1 ? a.w[1] : (s == 2 ? a.w[2] : a.w[3]));
    const long long ld = s == 0 ? a.ld[0] : (s == 1 ? a.ld[1] : (s == 2 ? a.ld[2] : a.ld[3]));
#pragma unroll
    for (int r = wave; r < 32; r += 4) {
        const size_t n = (size_t)(r >> 3) * a.R + blk * 8 + (r & 7);
        f32x4 v = {0, 0, 0, 0};
        if (q < nquad) v = ld4(src + n * ld + qs * 4);
        tile[lane][r] = v;
    }
    __syncthreads();
    float* out = a.wp + ((size_t)blk * nquad + q0) * 128;
#pragma unroll
    for (int e = tid; e < 64 * 32; e += 256) {
        const int qq = e >> 5, r = e & 31;
        if (q0 + qq < nquad) st4(out + (size_t)e * 4, tile[qq][r]);
    }
}

struct PackXArgs {
    const float* x[6];
    int q_end[6];          // running quad count after each segment
    long long ldx[6];
    int nseg, M;
    float* xq;
};

// up to 6 row-major segments [M <= 64, k_s] -> one quad-layout operand [sum k_s / 4][64][4] (rows beyond M zero)
__global__ __launch_bounds__(256) void pack_quad_segs_kernel(PackXArgs a) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int m = t & 63, q = t >> 6;
    if (q >= a.q_end[a.nseg - 1]) return;
    int s = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i) s += (i < a.nseg - 1 && q >= a.q_end[i]) ? 1 : 0;
    const int qs = q - (s ? a.q_end[s - 1] : 0);
    f32x4 v = {0, 0, 0, 0};
    if (m < a.M) v = ld4(a.x[s] + (size_t)m * a.ldx[s] + qs * 4);
    st4(a.xq + ((size_t)q * 64 + m) * 4, v);
}

}  // namespace

extern "C" int cvc_pack_lstm_segs(const float* const* ws, const long long* lds, const int* widths, int nseg, int R, float* wp,
                                  cvc_stream_t stream) {
    if (!ws || !lds || !widths || !wp || nseg < 1 || nseg > 4 || R < 8 || (R & 7)) return CVC_E_BADARG;
    PackWArgs a{};
    int q = 0;
    for (int s = 0; s < nseg; ++s) {
        if (!ws[s] || widths[s] < 4 || (widths[s] & 3) || lds[s] < widths[s] || (lds[s] & 3) || ((uintptr_t)ws[s] & 15)) return CVC_E_BADARG;
        q += widths[s] / 4;
        a.w[s] = ws[s]; a.ld[s] = lds[s]; a.q_end[s] = q;
    }
    if (q & 7) return CVC_E_BADARG;                              // K a multiple of 32
    a.nseg = nseg; a.R = R; a.wp = wp;
    hipLaunchKernelGGL(pack_lstm_w_kernel, dim3(R / 8, (q + 63) / 64), dim3(256), 0, (hipStream_t)stream, a);
    return cvc_launch_status();
}

extern "C" int cvc_pack_lstm_weights(const float* w_ih, int K_ih, const float* w_hh, int K_hh, int R, float* wp,
                                     cvc_stream_t stream) {
    if (!w_ih || !w_hh || !wp || K_ih < 4 || K_hh < 4 || (K_ih & 3) || (K_hh & 3) || ((K_ih + K_hh) & 31) || R < 8 || (R & 7))
        return CVC_E_BADARG;
    const float* ws[2] = {w_ih, w_hh};
    const long long lds[2] = {K_ih, K_hh};
    const int widths[2] = {K_ih, K_hh};
    return cvc_pack_lstm_segs(ws, lds, widths, 2, R, wp, stream);
}

extern "C" int cvc_pack_quad_segs(const float* const* xs, const long long* ldx, const int* widths, int nseg, int M, float* xq,
                                  cvc_stream_t stream) {
    if (!xs || !ldx || !widths || !xq || nseg < 1 || nseg > 6 || M < 1 || M > 64) return CVC_E_BADARG;
    PackXArgs a{};
    int q = 0;
    for (int s = 0; s < nseg; ++s) {
        if (!xs[s] || widths[s] < 4 || (widths[s] & 3) || (ldx[s] & 3) || ((uintptr_t)xs[s] & 15)) return CVC_E_BADARG;
        q += widths[s] / 4;
        a.x[s] = xs[s]; a.ldx[s] = ldx[s]; a.q_end[s] = q;
    }
    a.nseg = nseg; a.M = M; a.xq = xq;
    hipLaunchKernelGGL(pack_quad_segs_kernel, dim3((q * 64 + 255) / 256), dim3(256), 0, (hipStream_t)stream, a);
    return cvc_launch_status();
}

extern "C" int cvc_packed_linear_select_fwd(const float* wp, const float* xq, int K, const float* bias, int M, int Nout,
                                            float* top2_part, unsigned* counter, int unk_idx, int64_t* word, int word_stride,
                                            float* logprob, cvc_stream_t stream) {
    if (!wp || !xq || (K & 31) || Nout < 2 || !top2_part || !counter || !word || word_stride < 1 || M < 1 || M > 64) return CVC_E_BADARG;
    PackedArgs a{};
    a.wp = wp; a.xq = xq; a.nquad = K / 4; a.M = M; a.Nout = Nout; a.R = 0;
    a.bias = bias; a.ksplit = 1; a.top2_part = top2_part;
    a.sel_counter = counter; a.sel_unk = unk_idx; a.sel_word = word; a.sel_word_stride = word_stride; a.sel_logprob = logprob;
    return launch_packed<false>(a, (Nout + 31) / 32, (hipStream_t)stream);
}

extern "C" int cvc_packed_linear_fwd(const float* wp, const float* xq, int K, const float* bias, int M, int Nout,
                                     int ksplit, float* y, int ldy, float* top2_part, cvc_stream_t stream) {
    if (!wp || !xq || (K & 31) || Nout < 1 || ksplit < 1 || (!y && !top2_part)) return CVC_E_BADARG;
    if (ksplit > 1 && top2_part != nullptr) return CVC_E_BADARG;
    PackedArgs a{};
    a.wp = wp; a.xq = xq; a.nquad = K / 4; a.M = M; a.Nout = Nout; a.R = 0;
    a.bias = bias; a.y = y; a.ldy = ldy; a.ksplit = ksplit; a.split_stride = (long long)M * ldy; a.top2_part = top2_part;
    return launch_packed<false>(a, (Nout + 31) / 32, (hipStream_t)stream);
}

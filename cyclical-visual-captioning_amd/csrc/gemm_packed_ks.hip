// K-split gate GEMM of the decode engine (<= 64 rows) and its finishing kernel.
//
// What it fixes: in gemm_packed.hip one workgroup = 32 gate rows x all 64 batch rows x FULL K, so each of the 256 workgroups
// reads the whole activation block: 402 MB of L2 reads per lang-LSTM GEMM next to the 201 MB of weights that come from HBM
// (profiles/r01: loads-only 45 us, weights-only 31 us).  The activations cannot be shared below L2 there -- the waves of a
// workgroup split K, so no two of them want the same chunk.
//
// Here one workgroup = 256 gate rows (8 packed blocks, one per wave) x 64 batch rows x K / S, S = 8 K slices for R = 2048:
//   * weights: unchanged -- every wave streams ITS 32 rows straight into a register ring with non-temporal dwordx4 loads;
//   * activations: all 8 waves want the SAME 32-k chunk, so it is fetched ONCE per workgroup (each wave loads one quad = 1 KB),
//     split ONCE into the three bf16 terms (the per-wave split of X was 2/3 of the kernel's VALU work) and parked in LDS as
//     MFMA-ready fragments; the waves read them back with conflict-free ds_read_b128.  L2 activation reads: 402 -> 50 MB;
//   * the 32 x 64 partial tiles of the S slices go to fp32 slabs (16.8 MB, L2 / MALL resident) and a small finishing kernel
//     adds them in slice order, applies biases + cell update and writes h' / c' in the quad layout of the next GEMMs.
//     S maps onto the XCDs (slice = workgroup index mod 8), so an XCD's L2 holds one K slice of the activations.
// Arithmetic: identical products to the full-K kernel's split mode (6 bf16 MFMAs per 16 k); the summation order over K differs
// (fixed, deterministic).
#include "cvc_common.h"
#include "gemm_split.h"

#ifdef CVC_TS
__device__ unsigned long long cvc_ks_ts_buf[256 * 8];      // diagnostic build: per-workgroup phase timestamps of the exchange finish
extern "C" int cvc_debug_ks_ts_read(unsigned long long* dst) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(cvc_ks_ts_buf), sizeof(cvc_ks_ts_buf)) == hipSuccess ? 0 : -1;
}
#define KS_TS(k) do { if (tid == 0) cvc_ks_ts_buf[blockIdx.x * 8 + (k)] = wall_clock64(); } while (0)
#else
#define KS_TS(k) do {} while (0)
#endif

namespace {

struct KsArgs {
    const float* wp;     // packed weights [nblk][nquad][32][4]
    const float* xq;     // packed activations [nquad][64][4]
    int nquad;           // K / 4 (multiple of 8)
    long long wstride;   // floats between consecutive 32-row blocks of wp (>= nquad * 128)
    int nblk;            // R / 8 (multiple of 8)
    int ksplit;
    float* slab;         // [ksplit][nblk][64][32]
    // fused finish (cvc_packed_lstm_ksf_fwd): the last K slice of a 256-row tile to arrive sums the tile's slabs and does the
    // cell update in place of a finishing launch
    unsigned* counters;  // [R / 64] arrival counters, zero between launches (the last arriver resets its tile's)
    const float* b_ih; const float* b_hh; const float* gate_bias; const float* c_prev_q;
    float* c_out_q; float* h_dst1_q; float* h_dst2_q;
    int M, R;
    // exchange finish (cvc_packed_lstm_ksx_fwd): every K slice of a tile finishes one of the tile's 8 blocks
    unsigned* flags;     // [R / 64][8] arrival words (+ one error word at [R / 8]): slice ks of a tile stores `seq` when its slab rows are out
    unsigned seq;        // differs from the value of the previous launch on these flags
    const float* emb_gate; const int64_t* word;     // embedding-gate table [V][4R] + the batch rows' words, or null
    int local;           // 1: XCD-local exchange (plain slab stores, L2-served reads, the XCD id checked); 0: system-scope exchange
};

using u16x4 = __attribute__((ext_vector_type(4))) uint16_t;

#ifndef CVC_KS_DEPTH
#define CVC_KS_DEPTH 4
#endif
constexpr int XSTAGE = 12 * 1024;          // one 32-k activation chunk as fragments: [k16 step][row tile][term] x 1 KiB

__device__ __forceinline__ void stage_x(char* stage, const f32x4 v, int wave, int lane) {
    // this wave's quad of the chunk (k = 4 wave + e) for batch row `lane`: split, store 4 bf16 per term
    u16x4 p[3];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const unsigned u0 = __float_as_uint(v[e]);
        const float r1 = v[e] - __uint_as_float(u0 & 0xffff0000u);
        const unsigned u1 = __float_as_uint(r1);
        const float r2 = r1 - __uint_as_float(u1 & 0xffff0000u);
        p[0][e] = (uint16_t)(u0 >> 16); p[1][e] = (uint16_t)(u1 >> 16); p[2][e] = (uint16_t)(__float_as_uint(r2) >> 16);
    }
    // slot map shared with the weight fragments (gemm_packed.hip): lane half kh holds k = 16 kh + 8 s2 + slot
    const int kh = wave >> 2, s2 = (wave & 3) >> 1, slot0 = 4 * (wave & 1);
    const int mt = lane >> 5, i = lane & 31;
    char* base = stage + ((s2 * 2 + mt) * 3) * 1024 + ((kh * 32 + i) * 8 + slot0) * 2;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<u16x4*>(base + pl * 1024) = p[pl];
}

template <int NS> __device__ __forceinline__ void ks_finish_item(const KsArgs& a, int blk, int m, int hq);

// 16-byte load / store that no cache level may serve or keep (sc0 sc1: system scope) -- the exchange between workgroups that may
// sit on different XCDs (cdna guide, Guideline 16: {sc0 sc1 stores and loads on both sides})
__device__ __forceinline__ f32x4 ld4_sys(const float* p) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// ... served by L2, never by this CU's L1 (sc1): a same-XCD producer's acknowledged stores are there
__device__ __forceinline__ f32x4 ld4_l2(const float* p) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// ... non-temporal: bypasses this CU's L1, served by the XCD's L2 wherever the line is (dirty lines of same-XCD producers included)
__device__ __forceinline__ f32x4 ld4_nt(const float* p) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ unsigned ld1_nt(const unsigned* p) {
    unsigned v;
    asm volatile("global_load_dword %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// FUSED_NS = 0: partial tiles only (a finishing launch follows); > 0: that many K slices, fused finish -- by the last slice of a
// tile to arrive, or with XCHG by ALL of them: slice ks finishes block ks of the tile once the tile's 8 slabs are out
template <int FUSED_NS, bool XCHG = false>
__global__ __launch_bounds__(512) void packed_ks_kernel(KsArgs a) {
    __shared__ __attribute__((aligned(16))) char lds[2 * XSTAGE];
    __shared__ int last_arrival;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, kh = lane >> 5;
    const int S = a.ksplit;
    KS_TS(0);
    int ks = (int)blockIdx.x % S, tile = (int)blockIdx.x / S;
    if (FUSED_NS > 0 && ((a.nblk >> 3) & 7) == 0) {
        // all K slices of a tile on ONE XCD (workgroup x runs on XCD x % 8): the slabs the last arriver sums are then in its
        // own L2.  Only a placement hint -- correctness rests on the write-through stores and the arrival counter below.
        const int x = (int)blockIdx.x, l = x >> 3;
        ks = l % S;
        tile = (l / S) * 8 + (x & 7);
    }
    const int blk = tile * 8 + wave;
    const int nchunk_all = a.nquad >> 3;
    const int c_lo = nchunk_all * ks / S, c_hi = nchunk_all * (ks + 1) / S;
    const int n = c_hi - c_lo;
    // weights: quad q of block blk at wp + ((blk * nquad + q) * 32 + i) * 4; this lane takes quads 8c + 4kh + {0..3}
    const float* wl = a.wp + (size_t)blk * a.wstride + (size_t)i * 4 + (size_t)c_lo * 8 * 128 + kh * 4 * 128;
    // activations: this wave's quad of chunk c, one row per lane
    const float* xl = a.xq + ((size_t)(c_lo * 8 + wave) * 64 + lane) * 4;

    // XCHG: this workgroup finishes block (tile, ks); a thread of waves 0 / 1 owns (batch row m, hidden quad hq) of it and
    // requests the cell update's operands NOW -- nothing of them depends on the GEMM, and the table row is an HBM round trip
    const int fm = tid >> 1, fhq = tid & 1, fblk = tile * 8 + ks, fj = fblk * 8 + fhq * 4;
    const bool fwork = XCHG && tid < 128 && fm < a.M;
    const size_t fqoff = ((size_t)(fj >> 2) * 64 + fm) * 4;
    f32x4 fcp = {0, 0, 0, 0}, fadd[4][4];
    if constexpr (XCHG) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int k = 0; k < 4; ++k) fadd[g][k] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (fwork) {
            fcp = ld4(a.c_prev_q + fqoff);
            if (a.b_ih != nullptr) {
#pragma unroll
                for (int g = 0; g < 4; ++g) fadd[g][0] = ld4(a.b_ih + g * a.R + fj);
            }
            if (a.b_hh != nullptr) {
#pragma unroll
                for (int g = 0; g < 4; ++g) fadd[g][1] = ld4(a.b_hh + g * a.R + fj);
            }
            if (a.gate_bias != nullptr) {
#pragma unroll
                for (int g = 0; g < 4; ++g) fadd[g][2] = ld4(a.gate_bias + (size_t)fm * 4 * a.R + g * a.R + fj);
            }
            if (a.emb_gate != nullptr) {
                const float* trow = a.emb_gate + (size_t)a.word[fm] * 4 * a.R + fj;
#pragma unroll
                for (int g = 0; g < 4; ++g) fadd[g][3] = ld4(trow + (size_t)g * a.R);
            }
        }
    }

    f32x16 acc[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
    if (n > 0) {
        // every load below is unconditional (clamped chunk index): a load under a branch makes hipcc's s_waitcnt insertion
        // fall back to vmcnt(0) and the ring stops overlapping anything
        auto cl = [&](int c) { return c < n ? c : n - 1; };
        auto ldw = [&](f32x4 (&w)[4], int c) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                w[q] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(wl + (size_t)cl(c) * 8 * 128 + q * 128));
        };
        auto ldx = [&](int c) __attribute__((always_inline)) { return ld4(xl + (size_t)cl(c) * 8 * 256); };
        auto compute = [&](const f32x4 (&w)[4], const char* stage) __attribute__((always_inline)) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const Split3 W = split8(w[2 * s2], w[2 * s2 + 1]);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const char* f = stage + ((s2 * 2 + mt) * 3) * 1024 + lane * 16;
                    const u32x4 xh = *reinterpret_cast<const u32x4*>(f);
                    const u32x4 xm = *reinterpret_cast<const u32x4*>(f + 1024);
                    const u32x4 xo = *reinterpret_cast<const u32x4*>(f + 2048);
                    acc[mt] = mfma_bf16(W.mid, xm, acc[mt]);
                    acc[mt] = mfma_bf16(W.lo, xh, acc[mt]);
                    acc[mt] = mfma_bf16(W.hi, xo, acc[mt]);
                    acc[mt] = mfma_bf16(W.mid, xh, acc[mt]);
                    acc[mt] = mfma_bf16(W.hi, xm, acc[mt]);
                    acc[mt] = mfma_bf16(W.hi, xh, acc[mt]);
                }
            }
        };

        // Register rings of D chunks (weights: 4 KB per wave and chunk; activations: this wave's 1 KB quad), statically indexed
        // by full unrolling.  Per chunk c: request activations then weights of chunk c + D - 1 (unconditional, clamped index);
        // stage chunk c + 1's quad (requested D - 2 chunks ago) into the other LDS slot; multiply chunk c; barrier (publishes
        // c + 1, everyone is done reading c).  D - 1 chunks of weights (16 KB per wave at D = 5) stay in flight under the MFMAs.
        constexpr int D = CVC_KS_DEPTH;
        f32x4 wr[D][4], xr[D];
#pragma unroll
        for (int s = 0; s < D - 1; ++s) { xr[s] = ldx(s); ldw(wr[s], s); }
        stage_x(lds, xr[0], wave, lane);
        __syncthreads();
        for (int j = 0; j < n; j += D) {
#pragma unroll
            for (int s = 0; s < D; ++s) {
                const int c = j + s;                                  // wave-uniform; slots past the end only re-request
                xr[(s + D - 1) % D] = ldx(c + D - 1);
                ldw(wr[(s + D - 1) % D], c + D - 1);
                __builtin_amdgcn_sched_barrier(0);                    // requests first: the scheduler would sink them behind the MFMAs
                stage_x(lds + ((c + 1) & 1) * XSTAGE, xr[(s + 1) % D], wave, lane);
                if (c < n) compute(wr[s], lds + (c & 1) * XSTAGE);
                __syncthreads();
            }
        }
    }

    KS_TS(1);
    // partial tile -> slab[ks][blk][m][row]: row = e + 8 rq + 4 kh for register 4 rq + e, so a lane stores float4s
    float* out = a.slab + (((size_t)ks * a.nblk + blk) * 64) * 32;
    if constexpr (FUSED_NS > 0) {
        // the stores below are inline assembly and the compiler's hazard recognizer does not look inside: the wait for the
        // last MFMAs' results is spelled out here, and every store carries its own "s_nop 1" (a store of more than 8 bytes
        // followed by a write to its data registers needs wait states -- without them some lanes stored address bits)
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int m = mt * 32 + i;
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            const f32x4 v = {acc[mt][4 * rq], acc[mt][4 * rq + 1], acc[mt][4 * rq + 2], acc[mt][4 * rq + 3]};
            if (FUSED_NS > 0 && !(XCHG && a.local)) {
                // straight through L2 to memory: visible to whichever workgroup arrives last, without a write-back fence
                // (buffer_wbl2 walks the whole L2: measured 30 us per use in csrc/gru_persistent.hip)
                float* pp = out + (size_t)m * 32 + 8 * rq + 4 * kh;
                asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(pp), "v"(v) : "memory");
            } else {
                st4(out + (size_t)m * 32 + 8 * rq + 4 * kh, v);
            }
        }
    }
    if constexpr (XCHG) {
        // ---- exchange finish.  Slab rows out (write-through, acknowledged) -> this slice's arrival word -> wait for the tile's
        // 8 words -> sum block (tile, ks) over the slices in slice order -> cell update.  All 8 slices of a tile are resident
        // together (grid = 256 workgroups of 512 threads, one per CU); the wait is bounded and reports through the error word.
        constexpr int S = FUSED_NS;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        KS_TS(2);
        unsigned* fl = a.flags + (size_t)tile * 8;
        // XCD-local form: the slab rows were ordinary stores -- acknowledged = in THIS XCD's L2, which is what a reader on the same
        // XCD is served from when it bypasses its L1 (sc1 loads).  Whether the tile's 8 slices do share an XCD is not assumed: every
        // arrival word carries its writer's XCC_ID and a reader accepts the tile only when all 8 equal its own; anything else (another
        // placement, words that never show up because they sit in another XCD's L2) ends in the error word, never in a result.
        const unsigned xcc = a.local ? (__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u) : 0u;      // HW_REG_XCC_ID[3:0]
        const unsigned mine = (a.seq << 4) | xcc;
        if (tid == 0) {
            if (a.local) __hip_atomic_store(fl + ks, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      // plain store
            else __hip_atomic_store(fl + ks, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        if (wave == 0) {
            bool ok = false;
            for (int spin = 0; spin < (1 << 20); ++spin) {
                // (agent-scope relaxed load = sc1: served by L2, never by this CU's L1)
                unsigned v = mine;
                if (lane < S) {
                    if (a.local == 1 || a.local == 2) v = ld1_nt(fl + lane);
                    else if (a.local) v = __hip_atomic_load(fl + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else v = __hip_atomic_load(fl + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                if (__builtin_amdgcn_ballot_w64(v != mine) == 0) { ok = true; break; }
                if (__builtin_amdgcn_ballot_w64((v >> 4) == a.seq && v != mine) != 0) break;       // a slice of this launch on another XCD
                __builtin_amdgcn_s_sleep(1);
            }
            if (lane == 0) last_arrival = ok ? 1 : 0;
        }
        __syncthreads();
        if (!last_arrival) {                                              // not co-resident / a lost workgroup: say so, do not hang
            if (tid == 0) __hip_atomic_store(a.flags + (size_t)(a.nblk >> 3) * 8, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            return;
        }
        KS_TS(3);
        // thread (m = tid >> 3, rows 4 (tid & 7) .. + 3 of the block): the 8 slices' float4s requested together, summed in slice order
        const int sm = tid >> 3, srq = tid & 7;
        const float* sp = a.slab + (((size_t)fblk * 64 + sm) * 32) + srq * 4;
        f32x4 sv[S];
#pragma unroll
        for (int k = 0; k < S; ++k) {
            const float* q = sp + (size_t)k * a.nblk * 64 * 32;
            // (1: ordinary loads -- this CU has not touched these lines since the launch began, its L1 cannot hold them)
            sv[k] = a.local == 1 ? ld4(q) : (a.local == 2 ? ld4_nt(q) : (a.local == 3 ? ld4_l2(q) : ld4_sys(q)));
        }
        static_assert(S == 8, "the wait below names the 8 destination registers");
        // (the loads are inline assembly: the wait carries their destinations so that no use can be scheduled above it)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(sv[0]), "+v"(sv[1]), "+v"(sv[2]), "+v"(sv[3]), "+v"(sv[4]), "+v"(sv[5]), "+v"(sv[6]), "+v"(sv[7]) :: "memory");
        f32x4 t = sv[0];
#pragma unroll
        for (int k = 1; k < S; ++k) t += sv[k];
        KS_TS(4);
        float* sums = reinterpret_cast<float*>(lds);                      // [64][33] (the activation stages are done with)
#pragma unroll
        for (int e = 0; e < 4; ++e) sums[sm * 33 + srq * 4 + e] = t[e];
        __syncthreads();
        if (fwork) {
            f32x4 hv, cv;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float pre[4];
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    pre[g] = (((sums[fm * 33 + g * 8 + fhq * 4 + e] + fadd[g][0][e]) + fadd[g][1][e]) + fadd[g][2][e]) + fadd[g][3][e];
                const float ig = fast_sigmoid(pre[0]), fg = fast_sigmoid(pre[1]);
                const float gg = fast_tanh(pre[2]), og = fast_sigmoid(pre[3]);
                const float c2 = fg * fcp[e] + ig * gg;
                cv[e] = c2;
                hv[e] = og * fast_tanh(c2);
            }
            st4(a.c_out_q + fqoff, cv);
            if (a.h_dst1_q != nullptr) st4(a.h_dst1_q + fqoff, hv);
            if (a.h_dst2_q != nullptr) st4(a.h_dst2_q + fqoff, hv);
        }
        KS_TS(5);
    } else if constexpr (FUSED_NS > 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                                  // every wave's slab rows are acknowledged
        if (tid == 0) {
            const unsigned old = __hip_atomic_fetch_add(a.counters + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last_arrival = old == (unsigned)(S - 1);
            if (old == (unsigned)(S - 1)) __hip_atomic_store(a.counters + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (!last_arrival) return;
        // the tile's 8 blocks x 64 batch rows x 2 hidden quads: wave = block, two items per lane; slabs summed in slice order
        // whichever slice arrived last (these addresses were last read before this kernel started: no stale cache lines)
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int t = lane + 64 * it;
            ks_finish_item<FUSED_NS>(a, blk, t >> 1, t & 1);
        }
    }
}

template <int NS>
__device__ __forceinline__ void ks_finish_item(const KsArgs& a, int blk, int m, int hq) {
    if (m >= a.M) return;
    const int R = a.R;
    const int j = blk * 8 + hq * 4;
    // every read is requested before the first sum (bias terms behind uniform branches: see tile_lstm_finish_kernel)
    f32x4 pre[4], v[4][NS], bi[4], bh[4], gb[4];
    const float* p0 = a.slab + (((size_t)blk * 64 + m) * 32) + hq * 4;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int k = 0; k < NS; ++k) v[g][k] = ld4(p0 + g * 8 + (size_t)k * a.nblk * 64 * 32);
    const size_t qoff = ((size_t)(j >> 2) * 64 + m) * 4;
    const f32x4 cp = ld4(a.c_prev_q + qoff);
#pragma unroll
    for (int g = 0; g < 4; ++g) bi[g] = bh[g] = gb[g] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (a.b_ih != nullptr) {
#pragma unroll
        for (int g = 0; g < 4; ++g) bi[g] = ld4(a.b_ih + g * R + j);
    }
    if (a.b_hh != nullptr) {
#pragma unroll
        for (int g = 0; g < 4; ++g) bh[g] = ld4(a.b_hh + g * R + j);
    }
    if (a.gate_bias != nullptr) {
#pragma unroll
        for (int g = 0; g < 4; ++g) gb[g] = ld4(a.gate_bias + (size_t)m * 4 * R + g * R + j);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        f32x4 s = v[g][0];
#pragma unroll
        for (int k = 1; k < NS; ++k) s += v[g][k];
        pre[g] = ((s + bi[g]) + bh[g]) + gb[g];
    }
    f32x4 hv, cv;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float ig = fast_sigmoid(pre[0][e]), fg = fast_sigmoid(pre[1][e]);
        const float gg = fast_tanh(pre[2][e]), og = fast_sigmoid(pre[3][e]);
        const float c2 = fg * cp[e] + ig * gg;
        cv[e] = c2;
        hv[e] = og * fast_tanh(c2);
    }
    st4(a.c_out_q + qoff, cv);
    if (a.h_dst1_q != nullptr) st4(a.h_dst1_q + qoff, hv);
    if (a.h_dst2_q != nullptr) st4(a.h_dst2_q + qoff, hv);
}

struct KsFinishArgs {
    const float* slab; int ksplit; int nblk;
    const float* b_ih; const float* b_hh;     // [4R], nullable
    const float* gate_bias;                   // [M, 4R] row-major, nullable
    const float* c_prev_q;                    // quad layout [R/4][64][4]
    float* c_out_q; float* h_dst1_q; float* h_dst2_q;
    int M, R;
};

// NS = number of slabs (compile time: the slab reads of a gate are then issued together instead of one L2 round trip after
// the other)
template <int NS>
__global__ __launch_bounds__(128) void packed_ks_finish_kernel(KsFinishArgs a) {
    const int t = blockIdx.x * 128 + threadIdx.x;
    const int hq = t & 1, m = (t >> 1) & 63, blk = t >> 7;
    if (blk >= a.nblk || m >= a.M) return;
    const int R = a.R;
    const int j = blk * 8 + hq * 4;                            // first of this thread's 4 hidden units
    // every read is requested before the first sum (bias terms behind uniform branches: see tile_lstm_finish_kernel)
    f32x4 pre[4], v[4][NS], bi[4], bh[4], gb[4];
    const float* p0 = a.slab + (((size_t)blk * 64 + m) * 32) + hq * 4;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int k = 0; k < NS; ++k) v[g][k] = ld4(p0 + g * 8 + (size_t)k * a.nblk * 64 * 32);
    const size_t qoff = ((size_t)(j >> 2) * 64 + m) * 4;
    const f32x4 cp = ld4(a.c_prev_q + qoff);
#pragma unroll
    for (int g = 0; g < 4; ++g) bi[g] = bh[g] = gb[g] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (a.b_ih != nullptr) {
#pragma unroll
        for (int g = 0; g < 4; ++g) bi[g] = ld4(a.b_ih + g * R + j);
    }
    if (a.b_hh != nullptr) {
#pragma unroll
        for (int g = 0; g < 4; ++g) bh[g] = ld4(a.b_hh + g * R + j);
    }
    if (a.gate_bias != nullptr) {
#pragma unroll
        for (int g = 0; g < 4; ++g) gb[g] = ld4(a.gate_bias + (size_t)m * 4 * R + g * R + j);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        f32x4 s = v[g][0];
#pragma unroll
        for (int k = 1; k < NS; ++k) s += v[g][k];
        pre[g] = ((s + bi[g]) + bh[g]) + gb[g];
    }
    f32x4 hv, cv;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float ig = fast_sigmoid(pre[0][e]), fg = fast_sigmoid(pre[1][e]);
        const float gg = fast_tanh(pre[2][e]), og = fast_sigmoid(pre[3][e]);
        const float c2 = fg * cp[e] + ig * gg;
        cv[e] = c2;
        hv[e] = og * fast_tanh(c2);
    }
    st4(a.c_out_q + qoff, cv);
    if (a.h_dst1_q != nullptr) st4(a.h_dst1_q + qoff, hv);
    if (a.h_dst2_q != nullptr) st4(a.h_dst2_q + qoff, hv);
}

}  // namespace

extern "C" int cvc_packed_lstm_ks_slices(int K, int R) {
    if (R < 64 || (R & 63) || (K & 31)) return 0;           // not covered: use cvc_packed_lstm_fwd
    const int ntile = R / 64, nchunk = K / 32;
    int s = 8;                                               // a power of two: 8, 4, 2 or 1 slices
    while (s > 1 && (ntile * s > 256 || nchunk / s < 6)) s >>= 1;     // <= 256 workgroups, at least a few chunks per slice
    return s;
}

extern "C" int cvc_packed_lstm_ks_fwd(const float* wp, const float* xq, int K, const float* b_ih, const float* b_hh,
                                      const float* gate_bias, const float* c_prev_q, int M, int R, float* h_dst1_q,
                                      float* h_dst2_q, float* c_out_q, float* slab, long long w_blk_stride, cvc_stream_t stream) {
    if (!wp || !xq || !c_prev_q || !c_out_q || !slab || M < 1 || M > 64) return CVC_E_BADARG;
    const int S = cvc_packed_lstm_ks_slices(K, R);
    if (S < 1) return CVC_E_BADARG;
    KsArgs a{};
    a.wp = wp; a.xq = xq; a.nquad = K / 4; a.nblk = R / 8; a.ksplit = S; a.slab = slab;
    a.wstride = w_blk_stride > 0 ? w_blk_stride : (long long)(K / 4) * 128;
    if (a.wstride < (long long)(K / 4) * 128 || (a.wstride & 3)) return CVC_E_BADARG;
    hipLaunchKernelGGL(packed_ks_kernel<0>, dim3((R / 64) * S), dim3(512), 0, (hipStream_t)stream, a);
    int rc = cvc_launch_status();
    if (rc) return rc;
    KsFinishArgs f;
    f.slab = slab; f.ksplit = S; f.nblk = R / 8; f.b_ih = b_ih; f.b_hh = b_hh; f.gate_bias = gate_bias; f.c_prev_q = c_prev_q;
    f.c_out_q = c_out_q; f.h_dst1_q = h_dst1_q; f.h_dst2_q = h_dst2_q; f.M = M; f.R = R;
    const dim3 g(R / 8);                                     // one 128-thread workgroup per packed block: (64 rows) x (2 hidden quads)
    switch (S) {
        case 8: hipLaunchKernelGGL(packed_ks_finish_kernel<8>, g, dim3(128), 0, (hipStream_t)stream, f); break;
        case 4: hipLaunchKernelGGL(packed_ks_finish_kernel<4>, g, dim3(128), 0, (hipStream_t)stream, f); break;
        case 2: hipLaunchKernelGGL(packed_ks_finish_kernel<2>, g, dim3(128), 0, (hipStream_t)stream, f); break;
        default: hipLaunchKernelGGL(packed_ks_finish_kernel<1>, g, dim3(128), 0, (hipStream_t)stream, f); break;
    }
    return cvc_launch_status();
}

// K-split gate GEMM with the finish fused into its last-arriving slice: same operands and results as cvc_packed_lstm_ks_fwd, one
// launch.  counters: R / 64 words of device memory, zero before the first use (every launch leaves them zero).
// Measured (cfg2, in the decode graph): lang / att 62.9 / 57.5 us against 53.1 / 46.0 for the two-launch form and 52.0 / 43.9
// for the full-K kernel -- the finish of a tile is 14 us when ONE workgroup (the last arriver) does it (32 workgroups busy, 224
// idle, 64 dependent-latency loads per thread) against 7.4 us for the chip-wide finishing launch; placing a tile's slices on one
// XCD matters (without it 33 us: the slabs then come over the fabric), write-through vs ordinary stores is 3 us.  Kept
// selectable (DecodeEngine(gate_ksplit="fused")) and tested; not the default.
extern "C" int cvc_packed_lstm_ksf_fwd(const float* wp, const float* xq, int K, const float* b_ih, const float* b_hh,
                                       const float* gate_bias, const float* c_prev_q, int M, int R, float* h_dst1_q,
                                       float* h_dst2_q, float* c_out_q, float* slab, unsigned* counters, cvc_stream_t stream) {
    if (!wp || !xq || !c_prev_q || !c_out_q || !slab || !counters || M < 1 || M > 64) return CVC_E_BADARG;
    const int S = cvc_packed_lstm_ks_slices(K, R);
    if (S < 1) return CVC_E_BADARG;
    KsArgs a{};
    a.wp = wp; a.xq = xq; a.nquad = K / 4; a.nblk = R / 8; a.ksplit = S; a.slab = slab; a.wstride = (long long)(K / 4) * 128;
    a.counters = counters; a.b_ih = b_ih; a.b_hh = b_hh; a.gate_bias = gate_bias; a.c_prev_q = c_prev_q;
    a.c_out_q = c_out_q; a.h_dst1_q = h_dst1_q; a.h_dst2_q = h_dst2_q; a.M = M; a.R = R;
    const dim3 g((R / 64) * S);
    switch (S) {
        case 8: hipLaunchKernelGGL(packed_ks_kernel<8>, g, dim3(512), 0, (hipStream_t)stream, a); break;
        case 4: hipLaunchKernelGGL(packed_ks_kernel<4>, g, dim3(512), 0, (hipStream_t)stream, a); break;
        case 2: hipLaunchKernelGGL(packed_ks_kernel<2>, g, dim3(512), 0, (hipStream_t)stream, a); break;
        default: hipLaunchKernelGGL(packed_ks_kernel<1>, g, dim3(512), 0, (hipStream_t)stream, a); break;
    }
    return cvc_launch_status();
}

static int cvc_ksx_local = 3;
// The exchange's memory path: 1 / 2 / 3 (default 3) = XCD-local -- ordinary slab stores (acknowledged = in the XCD's L2), every arrival
// word carrying its writer's XCC_ID and checked; the slab rows read back with ordinary (1), non-temporal (2) or sc1 (3) loads --; 0 =
// system scope (write-through stores, system-scope loads: placement-independent, 5-6 us slower per launch: 16.8 MB of write-through
// stores take ~10 us to be acknowledged).  Returns the previous setting; < 0 queries.
extern "C" int cvc_packed_lstm_ksx_local(int on) {
    const int prev = cvc_ksx_local;
    if (on >= 0) cvc_ksx_local = on > 3 ? 1 : on;
    return prev;
}

// K-split gate GEMM whose finish is shared by all K slices of a tile ("exchange finish"): same operands and results as
// cvc_packed_lstm_ks_fwd, one launch, no finishing launch and no idle chip while one workgroup per tile finishes.
//   flags : R / 8 + 1 words of device memory ([R / 64][8] arrival words + the error word), zero before the first use;
//   seq   : any value that differs from the previous launch's on these flags (and from 0);
//   emb_gate / word : the embedding-gate form of the attention cell (cvc_packed_lstm_embgate_fwd), or null.
// Needs its R / 8 workgroups of 512 threads resident together (R <= 2048 on a 256-CU chip; one per CU); a slice that waits longer
// than the bound stores 1 to the error word flags[R / 8] and leaves its block unfinished.
extern "C" int cvc_packed_lstm_ksx_fwd(const float* wp, const float* xq, int K, const float* b_ih, const float* b_hh,
                                       const float* gate_bias, const float* emb_gate, const int64_t* word, const float* c_prev_q,
                                       int M, int R, float* h_dst1_q, float* h_dst2_q, float* c_out_q, float* slab,
                                       unsigned* flags, unsigned seq, cvc_stream_t stream) {
    if (!wp || !xq || !c_prev_q || !c_out_q || !slab || !flags || seq == 0 || M < 1 || M > 64 || (emb_gate && !word)) return CVC_E_BADARG;
    const int S = cvc_packed_lstm_ks_slices(K, R);
    if (S != 8 || (R / 64) * S > 256) return CVC_E_BADARG;             // the exchange is built for 8 slices (R = 2048: 256 workgroups)
    KsArgs a{};
    a.wp = wp; a.xq = xq; a.nquad = K / 4; a.nblk = R / 8; a.ksplit = S; a.slab = slab; a.wstride = (long long)(K / 4) * 128;
    a.b_ih = b_ih; a.b_hh = b_hh; a.gate_bias = gate_bias; a.c_prev_q = c_prev_q;
    a.c_out_q = c_out_q; a.h_dst1_q = h_dst1_q; a.h_dst2_q = h_dst2_q; a.M = M; a.R = R;
    a.flags = flags; a.seq = seq & 0x0fffffffu; a.emb_gate = emb_gate; a.word = word; a.local = cvc_ksx_local;
    if (a.seq == 0) return CVC_E_BADARG;
    hipLaunchKernelGGL((packed_ks_kernel<8, true>), dim3((R / 64) * S), dim3(512), 0, (hipStream_t)stream, a);
    return cvc_launch_status();
}

// Grouped stream-K GEMM of the decode engine (<= 64 rows): see "Grouped stream-K form" in include/cvc_hip.h and gsk.h.
//
// Why: a decode step is seven dependent launches, all of them streaming from HBM -- but only four fill the chip.  The vocabulary
// projection covers 157 of 256 CUs (20 us for 43 MB), h2attn is 7 us for 9 MB, and the LSTM gate GEMMs in their one-launch form
// make every workgroup re-read all activations from L2 (0.5 of the HBM spec).  Most of a gate GEMM's K range does not depend on
// what the critical path is computing at that moment: 80 % of the att-LSTM's (h_lang, h_att) is known once the language LSTM of
// the previous step is done, 2/3 of the lang-LSTM's (h_att, h_lang) once the att-LSTM is done.  So those parts ride in the SAME
// launch as the small GEMM of that moment (logits; h2attn) -- horizontal fusion, no second stream (two streams measured slower:
// cross-queue dependencies cost more than they hide) -- and the launch is balanced by construction:
//   * work unit = (256 weight rows, one 32-k chunk) = 8 waves x 4 KB of weights; all units of all groups form one linear space,
//     U = ceil(total / CUs) consecutive units per workgroup (stream-K): every CU streams the same bytes;
//   * inside a workgroup wave w owns block 8 tile + w (32 weight rows) and streams its weights straight into a register ring with
//     non-temporal dwordx4 loads; the activation chunk is fetched ONCE per workgroup (wave w loads quad w = 1 KB), split ONCE into
//     the three bf16 terms and parked in LDS as MFMA-ready fragments (the per-wave split of X was 2/3 of the one-launch kernel's
//     VALU work; its L2 activation reads were 2 x the weight bytes, here 1/4 of them);
//   * a workgroup's run inside one tile is a segment; its 256 x 64 partial tile goes to the group's slab and the consumer sums a
//     tile's segments in segment order (deterministic).
// Arithmetic: the packed path's split products (6 bf16 MFMAs per 16 k, gemm_split.h).
#include "cvc_common.h"
#include "gemm_split.h"
#include "gsk.h"
#include <stdlib.h>

namespace {

using u16x4 = __attribute__((ext_vector_type(4))) uint16_t;

#ifndef CVC_GSK_DEPTH
#define CVC_GSK_DEPTH 4
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

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also fences global memory: once a partial-tile store can
// precede it, hipcc waits before every barrier until those stores -- and, the counter being in-order, all but the newest loads
// -- have completed (s_waitcnt vmcnt(13) per chunk: the ring lost a third of its depth, +3 us per launch).
__device__ __forceinline__ void wg_barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct GskArgs {
    cvc_gsk_group g[3];
    int unit0[3];
    int ngroups, U, total;
    int abl;                  // measurement only (CVC_GSK_ABL): bit 0 = skip the partial-tile stores
};

constexpr int OUT_LD = 36;                                  // floats per batch row of the store staging tile (32 + pad: conflict-free b128)
constexpr int OUTSTAGE = 64 * OUT_LD * 4;                   // one wave's [64 rows][32 gate rows] partial tile, padded

// A workgroup's U units are ONE stream of chunks: the register ring keeps requesting across segment boundaries (the next tile's
// weights and activations are already in flight while the current segment's partial tile is stored), so a workgroup whose run
// straddles two tiles pays one pipeline fill, not two.  Two scalar cursors walk the unit space: the LOAD cursor D - 1 chunks
// ahead (pointers of the next request) and the COMPUTE cursor (chunks left in the current segment, where its tile goes).
__global__ __launch_bounds__(512) void gsk_kernel(GskArgs a) {
    __shared__ __attribute__((aligned(16))) char lds[2 * XSTAGE + 8 * OUTSTAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, kh = lane >> 5;
    const int wg = (int)blockIdx.x;
    const int ubeg = wg * a.U;
    const int ntot = min(a.U, a.total - ubeg);                       // chunks of this workgroup
    if (ntot <= 0) return;

    // the group's fields by selects: a runtime index into the kernel-argument array would be copied to scratch
#define GF(gi, f) ((gi) == 0 ? a.g[0].f : ((gi) == 1 ? a.g[1].f : a.g[2].f))
#define GU0(gi) ((gi) == 0 ? a.unit0[0] : ((gi) == 1 ? a.unit0[1] : a.unit0[2]))
    const int g0 = (a.ngroups > 2 && ubeg >= a.unit0[2]) ? 2 : ((a.ngroups > 1 && ubeg >= a.unit0[1]) ? 1 : 0);
    const int rel0 = ubeg - GU0(g0), nch0 = GF(g0, nchunk);
    const int tile0 = rel0 / nch0, c0 = rel0 - tile0 * nch0;

    // ---- load cursor
    int l_g = g0, l_tile = tile0, l_c = c0, l_nch = nch0, l_ntile = (GF(g0, nblk) + 7) >> 3, l_skip_at = GF(g0, skip_at), l_skip_n = GF(g0, skip_n);
    int l_left = ntot - 1;                                            // advances left; past the end the last chunk is re-requested
    auto wbase = [&](int gi, int tile) {                              // a short last tile: the spare waves re-read its last block
        const int nblk = GF(gi, nblk), blk = tile * 8 + wave;
        return GF(gi, wp) + (size_t)(blk < nblk ? blk : nblk - 1) * GF(gi, w_blk_stride);
    };
    const float* l_wb = wbase(g0, tile0);
    const float* l_xb = GF(g0, xq);
    const int loff_w = i * 4 + kh * 4 * 128, loff_x = (wave * 64 + lane) * 4;
    auto l_advance = [&]() {
        if (l_left <= 0) return;
        --l_left;
        if (++l_c == l_nch) {
            l_c = 0;
            if (++l_tile == l_ntile) {
                l_tile = 0;
                ++l_g;                                                // (l_left > 0 guarantees another group exists)
                l_nch = GF(l_g, nchunk); l_ntile = (GF(l_g, nblk) + 7) >> 3; l_skip_at = GF(l_g, skip_at); l_skip_n = GF(l_g, skip_n);
                l_xb = GF(l_g, xq);
            }
            l_wb = wbase(l_g, l_tile);
        }
    };
    // every load is unconditional: a load under a branch makes hipcc's s_waitcnt insertion fall back to vmcnt(0)
    auto request = [&](f32x4 (&w)[4], f32x4& x) __attribute__((always_inline)) {
        const int phys = l_c < l_skip_at ? l_c : l_c + l_skip_n;
        const float* pw = l_wb + (size_t)phys * 8 * 128 + loff_w;
        x = ld4(l_xb + (size_t)phys * 8 * 256 + loff_x);
#pragma unroll
        for (int q = 0; q < 4; ++q) w[q] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(pw + q * 128));
        l_advance();
    };

    // ---- compute cursor: the segment chunk j belongs to
    int c_g = g0, c_tile = tile0;
    int c_left = min(ntot, nch0 - c0);                                // chunks left in the current segment
    int c_after = ntot - c_left;                                      // chunks of this workgroup behind the current segment

    f32x16 acc[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;

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

    // end of a segment: the wave's 64 x 32 partial tile goes through its own LDS staging tile (register r = 4 rq + e holds gate
    // row e + 8 rq + 4 kh of batch row 32 mt + i) and leaves as eight fully coalesced 1 KB stores -- the tile is 8 KB contiguous
    // in the slab; straight from the registers a store instruction touched 32 lines with 32 bytes each
    auto flush = [&]() __attribute__((always_inline)) {
        const int nblk = GF(c_g, nblk), nch = GF(c_g, nchunk), maxseg = GF(c_g, maxseg);
        const int seg = wg - (GU0(c_g) + c_tile * nch) / a.U;
        float* stg = reinterpret_cast<float*>(lds + 2 * XSTAGE + wave * OUTSTAGE);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const f32x4 v = {acc[mt][4 * rq], acc[mt][4 * rq + 1], acc[mt][4 * rq + 2], acc[mt][4 * rq + 3]};
                st4(stg + (mt * 32 + i) * OUT_LD + 8 * rq + 4 * kh, v);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[mt][4 * rq + e] = 0.f;
            }
        asm volatile("" ::: "memory");                                 // (the reads below are of other lanes' writes: keep the order)
        if (c_tile * 8 + wave < nblk && !(a.abl & 1)) {
            float* out = GF(c_g, slab) + ((size_t)(c_tile * maxseg + seg) * 8 + wave) * 2048;
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int q = it * 64 + lane;                         // float4 #q of the tile: batch row q >> 3, gate rows 4 (q & 7) ..
                st4(out + (size_t)q * 4, ld4(stg + (q >> 3) * OUT_LD + (q & 7) * 4));
            }
        }
        // next segment of this workgroup: the following tile (of the following group after a group's last tile)
        if (c_after > 0) {
            if (++c_tile == ((nblk + 7) >> 3)) { c_tile = 0; ++c_g; }
            c_left = min(c_after, GF(c_g, nchunk));
            c_after -= c_left;
        }
    };

    // Register rings of D chunks (weights: 4 KB per wave and chunk; activations: this wave's 1 KB quad), statically indexed by
    // full unrolling.  Per chunk c: request chunk c + D - 1; stage chunk c + 1's quad (requested D - 2 chunks ago) into the other
    // LDS slot; multiply chunk c; barrier (publishes c + 1, everyone is done reading c).
    constexpr int D = CVC_GSK_DEPTH;
    f32x4 wr[D][4], xr[D];
#pragma unroll
    for (int s = 0; s < D - 1; ++s) request(wr[s], xr[s]);
    stage_x(lds, xr[0], wave, lane);
    wg_barrier_lds();
    for (int j = 0; j < ntot; j += D) {
#pragma unroll
        for (int s = 0; s < D; ++s) {
            const int c = j + s;                                      // wave-uniform; slots past the end only re-request
            request(wr[(s + D - 1) % D], xr[(s + D - 1) % D]);
            __builtin_amdgcn_sched_barrier(0);                        // requests first: the scheduler would sink them behind the MFMAs
            stage_x(lds + ((c + 1) & 1) * XSTAGE, xr[(s + 1) % D], wave, lane);
            if (c < ntot) {
                compute(wr[s], lds + (c & 1) * XSTAGE);
                if (--c_left == 0) flush();
            }
            wg_barrier_lds();
        }
    }
#undef GF
#undef GU0
}

}  // namespace

extern "C" int cvc_gsk_plan(const int* ntile, const int* nchunk, int ngroups, int nwg, int* U, int* unit0, int* maxseg) {
    if (!ntile || !nchunk || !U || !unit0 || !maxseg || ngroups < 1 || ngroups > 3) return CVC_E_BADARG;
    if (nwg <= 0) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
            cus < 1)
            cus = 256;
        nwg = cus;
    }
    long long total = 0;
    for (int g = 0; g < ngroups; ++g) {
        if (ntile[g] < 1 || nchunk[g] < 1) return CVC_E_BADARG;
        unit0[g] = (int)total;
        total += (long long)ntile[g] * nchunk[g];
    }
    if (total > (1ll << 30)) return CVC_E_TOOBIG;
    const int u = (int)((total + nwg - 1) / nwg);
    *U = u;
    for (int g = 0; g < ngroups; ++g) {
        int mx = 1;
        for (int t = 0; t < ntile[g]; ++t) {
            const int s = gsk_nseg(unit0[g], nchunk[g], u, t);
            mx = s > mx ? s : mx;
        }
        maxseg[g] = mx;
    }
    return 0;
}

extern "C" int cvc_gsk_gemm(const cvc_gsk_group* groups, int ngroups, int U, cvc_stream_t stream) {
    if (!groups || ngroups < 1 || ngroups > 3 || U < 1) return CVC_E_BADARG;
    GskArgs a{};
    long long total = 0;
    for (int g = 0; g < ngroups; ++g) {
        const cvc_gsk_group& G = groups[g];
        if (!G.wp || !G.xq || !G.slab || G.nblk < 1 || G.nchunk < 1 || G.maxseg < 1 || G.skip_at < 0 || G.skip_n < 0 ||
            (G.w_blk_stride & 3) || ((uintptr_t)G.wp & 15) || ((uintptr_t)G.xq & 15) || ((uintptr_t)G.slab & 15))
            return CVC_E_BADARG;
        a.g[g] = G;
        a.unit0[g] = (int)total;
        const int ntile = (G.nblk + 7) / 8;
        total += (long long)ntile * G.nchunk;
        // the slab must hold every segment this U produces
        for (int t = 0; t < ntile; ++t)
            if (gsk_nseg(a.unit0[g], G.nchunk, U, t) > G.maxseg) return CVC_E_BADARG;
    }
    if (total > (1ll << 30)) return CVC_E_TOOBIG;
    a.ngroups = ngroups; a.U = U; a.total = (int)total;
    const int nwg = (int)((total + U - 1) / U);
    static const int abl_mode = [] { const char* e = getenv("CVC_GSK_ABL"); return e ? atoi(e) : 0; }();      // ablations (benchmarks only)
    a.abl = abl_mode;
    hipLaunchKernelGGL(gsk_kernel, dim3(nwg), dim3(512), 0, (hipStream_t)stream, a);
    return cvc_launch_status();
}

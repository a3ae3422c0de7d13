// Persistent form of the GRU recurrence (the encoder's frame context, reference backbone.py:103-106, 335-338).
//
// cvc_gru_seq_fwd (gemm_packed.hip) launches the packed gate GEMM once per time step: at H = 1024 that is 19 us per step, of
// which ~12 us are launch / drain latency and ~7 us the re-read of W_hh (33 MB per step).  Here ONE cooperative launch runs
// the whole sequence:
//   * a workgroup owns 8 hidden units of one direction (their r, z, n rows of W_hh: a 32-row MFMA tile whose last 8 rows are
//     zero) for all F steps, and keeps those weights IN REGISTERS, already split into the three bf16 terms of the split-product
//     arithmetic (gemm_split.h): 4 waves x K/4, up to 192 VGPRs per lane -- W_hh is read from memory once per sequence;
//   * per step it reads the whole previous state of its direction (quad layout, 256 KB at 64 clips, from L2), multiplies, sums
//     the 4 waves' partial tiles through LDS in a fixed order, applies the gate arithmetic and writes its 8 units of h_t;
//   * steps are separated by a per-direction barrier in global memory (release add / spin at agent scope).  The 8 XCDs' L2s
//     are not coherent with each other, and invalidating them at every step costs more than the step (measured: 52 us per
//     step, every workgroup re-fetching the state over the fabric).  Instead every step's state gets its OWN slot
//     ([F + 1][ndir][H/4][64][4], 252 MB at config 2): an address is written once, before the barrier, and first read after
//     it, so no cache can hold a stale copy and the readers need no invalidate -- the first workgroup of an XCD misses, the
//     other 31 hit its L2.  The spin is bounded: if a peer never arrives (not all workgroups resident) the kernel raises the
//     error word instead of hanging the GPU, and the host falls back to the per-step form.
#include "cvc_common.h"
#include <stdlib.h>
#include "gemm_split.h"

namespace {

// sync buffer: word SYNC_ERR = error flag; then 4 groups (direction x batch half) of CNT arrival counters, CNT_STRIDE words apart
constexpr int CNT = 32, CNT_STRIDE = 1024, SYNC_ERR = 4;
constexpr long long SYNC_WORDS = SYNC_ERR + 8 + 4LL * CNT * CNT_STRIDE;

struct GruPArgs {
    const float* wp; long long w_stride;          // packed W_hh [ndir][H/8][Kp/4][32][4]
    const float* gi; long long gi_ld_m, gi_ld_t;  // input projections (no bias), columns [ndir][3H]
    const float* b_ih; const float* b_hh;         // [ndir][3H]
    int M, F, H, Kp;
    float* hq; long long h_stride;                // state slots, quad layout: [F + 1][ndir][Kp/4][64][4], slot 0 = h0
    float* y; long long y_ld_m, y_ld_t;
    float* gates; long long g_ld_m, g_ld_t;       // training: (r, z, n, W_hn h + b_hn) of every step, columns [ndir][4][H]; nullable
    unsigned* sync;                               // SYNC_WORDS words: error word + arrival counters (see above)
    unsigned spin_limit;
};

// NC = 32-k chunks per wave (K = 128 * NC).  The batch is processed as NH independent groups of MT 32-clip tiles
// (NH x MT = 1 or 2): with NH = 2 the two halves of a 64-clip batch are separate recurrences with their own arrival counters,
// walked alternately -- while the other workgroups' states of one half are still in flight, this workgroup computes the other
// half, so that a barrier's latency hides behind useful work.
// NW = 4 waves (one per SIMD, K / 4 each) or 8 (two per SIMD, K / 8 each: while one wave of a SIMD splits its activations on
// the VALU the other's MFMAs run -- measured 7.4 us of a 14 us step are this compute chain with 4 waves).
template <int NH, int MT, int NC, int NW>
__global__ __launch_bounds__(NW * 64, 1) void gru_persistent_kernel(GruPArgs a) {
    constexpr int LDM = MT * 32 + 1;
    __shared__ float red[NW * 32 * LDM];
    __shared__ int gave_up;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, kh = lane >> 5;
    const int dir = blockIdx.y, blk = blockIdx.x, H = a.H, M = a.M;
    const int nquad = a.Kp >> 2;
    const unsigned nblk = gridDim.x;

    // ---- this wave's share of the weights, split once.  Register slot c holds chunk wave + 4 * ((c + rot) % NC).  (Measured:
    // starting every workgroup at a different chunk, which helps the per-step kernel, is 10 % SLOWER here -- 18.5 vs 16.7 us
    // per step -- the workgroups of an XCD asking for the same state lines at the same time is what the L2 serves best.)
    Split3 W[NC][2];
#ifndef CVC_GRU_ROT
#define CVC_GRU_ROT 0
#endif
    const int rot = CVC_GRU_ROT ? (blk * 5 + dir * 3) % NC : 0;
    int chunk_of[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) chunk_of[c] = wave + NW * ((c + rot) % NC);
    {
        const float* wl = a.wp + (size_t)dir * a.w_stride + ((size_t)blk * nquad * 32 + i) * 4 + kh * 4 * 128;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const float* w = wl + (size_t)chunk_of[c] * 8 * 128;
            const f32x4 q0 = ld4(w), q1 = ld4(w + 128), q2 = ld4(w + 256), q3 = ld4(w + 384);
            W[c][0] = split8(q0, q1);
            W[c][1] = split8(q2, q3);
        }
    }

    // ---- epilogue role: thread (em = clip within the group, eqd = which 4 of the 8 hidden units)
    constexpr int ET = MT * 32;
    const int em = tid % ET, eqd = tid / ET;
    const int ejq = blk * 8 + (eqd & 1) * 4;
    f32x4 ebias[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    if (eqd < 2) {
        const float* bi = a.b_ih + (size_t)dir * 3 * H + ejq;
        const float* bh = a.b_hh + (size_t)dir * 3 * H + ejq;
        ebias[0] = ld4(bi) + ld4(bh);
        ebias[1] = ld4(bi + H) + ld4(bh + H);
        ebias[2] = ld4(bi + 2 * H);
        ebias[3] = ld4(bh + 2 * H);
    }

    if (tid == 0) gave_up = 0;
    __syncthreads();
    for (int s = 0; s < a.F; ++s) {
        const long long t = dir == 0 ? s : a.F - 1 - s;
        const float* hprev = a.hq + ((size_t)s * gridDim.y + dir) * a.h_stride;
        float* hnext = a.hq + ((size_t)(s + 1) * gridDim.y + dir) * a.h_stride;
#pragma unroll
        for (int hf = 0; hf < NH; ++hf) {
            const int m = hf * ET + em;                                // clip row of the epilogue role
            const bool ework = eqd < 2 && m < M;
            const size_t eqoff = ((size_t)(ejq / 4) * 64 + (m < 64 ? m : 63)) * 4;
            // arrivals are spread over CNT counters 4 KB apart (different memory channels): 128 increments of ONE word queue up
            // behind each other at the memory-side atomic unit, and the last arrival is the one everybody waits for
            unsigned* counter = a.sync + SYNC_ERR + 8 + (size_t)((dir * 2 + hf) * CNT) * CNT_STRIDE;

            // x-projections of this step: independent of the other workgroups, requested before the wait
            f32x4 egi[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
            if (ework) {
                const float* gi = a.gi + (size_t)m * a.gi_ld_m + t * a.gi_ld_t + (size_t)dir * 3 * H + ejq;
#pragma unroll
                for (int g = 0; g < 3; ++g) egi[g] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(gi + g * H));
            }

            // ---- wait until every workgroup of this direction has published step s - 1 of this group
            if (s > 0) {
                if (wave == 0) {                                       // lanes 0 .. CNT-1 read one counter each
                    const unsigned target = nblk * (unsigned)s;
                    unsigned it = 0;
                    for (;;) {
                        unsigned v = lane < CNT ? __hip_atomic_load(counter + (size_t)lane * CNT_STRIDE, __ATOMIC_RELAXED,
                                                                    __HIP_MEMORY_SCOPE_AGENT) : 0u;
#pragma unroll
                        for (int o = 1; o < CNT; o <<= 1) v += __shfl_xor(v, o, 64);
                        v = __builtin_amdgcn_readfirstlane(v);         // lane 0 holds the sum: one decision for the wave
                        if (v >= target) break;
                        if (++it > a.spin_limit || __hip_atomic_load(a.sync + SYNC_ERR, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                            if (lane == 0) {
                                __hip_atomic_store(a.sync + SYNC_ERR, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                gave_up = 1;                           // tell the workgroup
                            }
                            break;
                        }
                        __builtin_amdgcn_s_sleep(2);
                    }
                }
                __syncthreads();                                       // (also: the previous group's readers of `red` are done)
                if (gave_up) return;                                   // no invalidate: slot s has never been read before
            } else if (hf > 0) {
                __syncthreads();
            }

            // ---- partial tiles: this wave's K slice (chunks wave, wave + 4, ...) of this group's clip tiles
            f32x16 acc[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
            // The activations are requested in phases of (half of the wave's chunks) x (one 32-clip tile), two phases in
            // flight (128 registers next to the 192 of the weights); the schedule is pinned, otherwise the compiler requests
            // the later phases one load at a time with a full wait behind each
            constexpr int HC = (NC + 1) / 2;                          // chunks per phase
            const float* xl = hprev + (size_t)i * 4 + kh * 4 * 256 + hf * ET * 4;
            f32x4 xb[2][HC][4];
            auto load_phase = [&](f32x4 (&buf)[HC][4], const int half, const int mt) __attribute__((always_inline)) {
#pragma unroll
                for (int j = 0; j < HC; ++j) {
                    const int c = half * HC + j;
                    if (c < NC) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) buf[j][q] = ld4(xl + (size_t)chunk_of[c] * 8 * 256 + q * 256 + mt * 128);
                    }
                }
            };
            auto mma_phase = [&](const f32x4 (&buf)[HC][4], const int half, f32x16& d) __attribute__((always_inline)) {
#pragma unroll
                for (int j = 0; j < HC; ++j) {
                    const int c = half * HC + j;
                    if (c < NC) {
#pragma unroll
                        for (int s2 = 0; s2 < 2; ++s2) {
                            const Split3 X = split8(buf[j][2 * s2], buf[j][2 * s2 + 1]);
                            const Split3& Wc = W[c][s2];
                            d = mfma_bf16(Wc.mid, X.mid, d);
                            d = mfma_bf16(Wc.lo, X.hi, d);
                            d = mfma_bf16(Wc.hi, X.lo, d);
                            d = mfma_bf16(Wc.mid, X.hi, d);
                            d = mfma_bf16(Wc.hi, X.mid, d);
                            d = mfma_bf16(Wc.hi, X.hi, d);
                        }
                    }
                }
            };
            load_phase(xb[0], 0, 0);
            load_phase(xb[1], 1, 0);
            const f32x4 ehp = ld4(hprev + eqoff);                     // (every thread: an unconditional load keeps the waits counted)
            __builtin_amdgcn_sched_barrier(0);
            mma_phase(xb[0], 0, acc[0]);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (MT == 2) load_phase(xb[0], 0, 1);
            __builtin_amdgcn_sched_barrier(0);
            mma_phase(xb[1], 1, acc[0]);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (MT == 2) {
                load_phase(xb[1], 1, 1);
                __builtin_amdgcn_sched_barrier(0);
                mma_phase(xb[0], 0, acc[1]);
                __builtin_amdgcn_sched_barrier(0);
                mma_phase(xb[1], 1, acc[1]);
            }

            // ---- ordered cross-wave sum, gate arithmetic
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * kh;
                    red[(wave * 32 + row) * LDM + mt * 32 + i] = acc[mt][r];
                }
            __syncthreads();
            if (ework) {
                f32x4 hv, gsave[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int jj = eqd * 4 + e;
                    float pre[3];
#pragma unroll
                    for (int g = 0; g < 3; ++g) {
                        const int row = g * 8 + jj;
                        pre[g] = (red[(0 * 32 + row) * LDM + em] + red[(1 * 32 + row) * LDM + em]) +
                                 (red[(2 * 32 + row) * LDM + em] + red[(3 * 32 + row) * LDM + em]);
                        if constexpr (NW == 8)
                            pre[g] += (red[(4 * 32 + row) * LDM + em] + red[(5 * 32 + row) * LDM + em]) +
                                      (red[(6 * 32 + row) * LDM + em] + red[(7 * 32 + row) * LDM + em]);
                    }
                    const float rg = fast_sigmoid(pre[0] + egi[0][e] + ebias[0][e]);
                    const float zg = fast_sigmoid(pre[1] + egi[1][e] + ebias[1][e]);
                    const float ng = fast_tanh(egi[2][e] + ebias[2][e] + rg * (pre[2] + ebias[3][e]));
                    hv[e] = ng + zg * (ehp[e] - ng);
                    gsave[0][e] = rg; gsave[1][e] = zg; gsave[2][e] = ng; gsave[3][e] = pre[2] + ebias[3][e];
                }
                if (a.gates != nullptr) {
                    float* gp = a.gates + (size_t)m * a.g_ld_m + t * a.g_ld_t + (size_t)dir * 4 * H + ejq;
#pragma unroll
                    for (int g = 0; g < 4; ++g) st4(gp + g * H, gsave[g]);
                }
                // the state goes straight through this XCD's L2 to memory (sc0 sc1): a release fence would instead walk the
                // whole L2 for dirty lines (buffer_wbl2) once per workgroup and step.  (Inline assembly is invisible to the
                // compiler's hazard recognizer: the s_nop covers "wide store followed by a write to its data registers".)
                float* hp = hnext + eqoff;
                asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(hp), "v"(hv) : "memory");
                st4(a.y + (size_t)m * a.y_ld_m + t * a.y_ld_t + (size_t)dir * H + ejq, hv);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            // ---- publish: one arrival per workgroup, after all of its state stores have been acknowledged
            __syncthreads();
            if (tid == 0)
                __hip_atomic_fetch_add(counter + (size_t)(blk % CNT) * CNT_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

__global__ __launch_bounds__(256) void gru_zero_kernel(float* p, long long n, unsigned* sync) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t < n) p[t] = 0.f;
    if (t < SYNC_WORDS) sync[t] = 0u;
}

template <int NH, int MT, int NC, int NW>
int launch_persistent(GruPArgs& a, int ndir, hipStream_t st) {
    void* params[] = {&a};
    const dim3 grid(a.H / 8, ndir);
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)gru_persistent_kernel<NH, MT, NC, NW>, NW * 64, 0) != hipSuccess) {
        (void)hipGetLastError();
        return CVC_E_BADARG;
    }
    int devid = 0, cus = 0;
    if (hipGetDevice(&devid) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, devid) != hipSuccess)
        return CVC_E_BADARG;
    if ((long long)per_cu * cus < (long long)grid.x * grid.y) return CVC_E_BADARG;     // would not be co-resident
    // An ORDINARY launch, not hipLaunchCooperativeKernel (round 6).  Co-residency is what the kernel needs, and the occupancy check
    // above plus the stream's in-order execution give it (the grid is at most one workgroup per CU on an otherwise idle chip);
    // what the cooperative launch adds is a trip through the runtime's device-wide cooperative queue -- and with it a state of the
    // runtime's hardware queues in which, once any other stream capture has happened in the process, EVERY later kernel of the
    // step took 10 - 25 us longer (the captured end-to-end training step 91 -> 122 ms; tools/runs/r06_e2e_after_decode.py,
    // GPU_MAX_HW_QUEUES <= 2 or per-step GRU forms made it disappear).  A grid that is not resident after all is caught as before:
    // the barrier's spin is bounded and raises the error word (the caller falls back / the step is voided and re-run).
    // CVC_GRU_COOPERATIVE=1 restores the cooperative launch (A/B).
    static const bool coop = [] { const char* e = getenv("CVC_GRU_COOPERATIVE"); return e && e[0] == '1'; }();
    if (coop) {
        if (hipLaunchCooperativeKernel((const void*)gru_persistent_kernel<NH, MT, NC, NW>, grid, dim3(NW * 64), params, 0, st) != hipSuccess) {
            (void)hipGetLastError();
            return CVC_E_BADARG;
        }
    } else {
        hipLaunchKernelGGL((gru_persistent_kernel<NH, MT, NC, NW>), grid, dim3(NW * 64), 0, st, a);
    }
    return cvc_launch_status();
}

}  // namespace

extern "C" int cvc_gru_persistent_sync_words(void) { return (int)SYNC_WORDS; }

static int cvc_gru_waves8 = 1;
// A/B + test hook: 1 (default) = 8 waves per workgroup where K is a multiple of 256, 0 = always 4.  Returns the previous setting.
extern "C" int cvc_gru_persistent_waves8(int on) {
    const int prev = cvc_gru_waves8;
    if (on >= 0) cvc_gru_waves8 = on ? 1 : 0;
    return prev;
}

static int cvc_gru_halves = 0;
// A/B + test hook: 1 = a batch of more than 32 clips runs as two interleaved 32-clip recurrences with their own counters,
// 0 (default) = all clips in one recurrence.  Returns the previous setting; < 0 queries.
// Measured: the interleaved form is SLOWER (18.6 vs 14.9 us per step at 64 clips, H = 1024, both directions): a step is not
// waiting for the other workgroups but walking its own chain of dependent latencies (counter poll -> state loads from L2 ->
// MFMAs -> LDS sum -> write-through store acknowledged -> arrival), about 9.5 us for 32 clips whatever else is going on, and
// a workgroup executes its two halves one after the other.
extern "C" int cvc_gru_persistent_halves(int on) {
    const int prev = cvc_gru_halves;
    if (on >= 0) cvc_gru_halves = on ? 1 : 0;
    return prev;
}

// Same operands and results as cvc_gru_seq_fwd (include/cvc_hip.h) except hq: (F + 1) * ndir * H * 64 floats; `sync` = cvc_gru_persistent_sync_words() words of device memory (arrival counters
// and, at word 4, an error word that is non-zero afterwards when the barrier timed out -- the outputs are then invalid).  Returns
// CVC_E_BADARG for shapes outside the persistent form (H % 128 != 0, H > 1024, more workgroups than can be resident): use
// cvc_gru_seq_fwd then.
static int gru_persistent_impl(const float* wp, const float* gi, long long gi_ld_m, long long gi_ld_t, const float* b_ih,
                               const float* b_hh, int M, int F, int H, int ndir, float* hq, float* y, long long y_ld_m,
                               long long y_ld_t, float* gates, long long g_ld_m, long long g_ld_t, unsigned* sync,
                               cvc_stream_t stream);

extern "C" int cvc_gru_seq_persistent_fwd(const float* wp, const float* gi, long long gi_ld_m, long long gi_ld_t, const float* b_ih,
                                          const float* b_hh, int M, int F, int H, int ndir, float* hq, float* y,
                                          long long y_ld_m, long long y_ld_t, unsigned* sync, cvc_stream_t stream) {
    return gru_persistent_impl(wp, gi, gi_ld_m, gi_ld_t, b_ih, b_hh, M, F, H, ndir, hq, y, y_ld_m, y_ld_t, nullptr, 0, 0, sync, stream);
}

// Training form: additionally writes, for every step and direction, the gates autograd needs -- (r, z, n, W_hn h + b_hn) at
// gates + m * g_ld_m + t * g_ld_t + d * 4H + {0, H, 2H, 3H}
extern "C" int cvc_gru_seq_persistent_train_fwd(const float* wp, const float* gi, long long gi_ld_m, long long gi_ld_t,
                                                const float* b_ih, const float* b_hh, int M, int F, int H, int ndir, float* hq,
                                                float* y, long long y_ld_m, long long y_ld_t, float* gates, long long g_ld_m,
                                                long long g_ld_t, unsigned* sync, cvc_stream_t stream) {
    if (!gates || (g_ld_m & 3) || (g_ld_t & 3)) return CVC_E_BADARG;
    return gru_persistent_impl(wp, gi, gi_ld_m, gi_ld_t, b_ih, b_hh, M, F, H, ndir, hq, y, y_ld_m, y_ld_t, gates, g_ld_m, g_ld_t, sync,
                               stream);
}

static int gru_persistent_impl(const float* wp, const float* gi, long long gi_ld_m, long long gi_ld_t, const float* b_ih,
                               const float* b_hh, int M, int F, int H, int ndir, float* hq, float* y, long long y_ld_m,
                               long long y_ld_t, float* gates, long long g_ld_m, long long g_ld_t, unsigned* sync,
                               cvc_stream_t stream) {
    if (!wp || !gi || !b_ih || !b_hh || !hq || !y || !sync || M < 1 || M > 64 || F < 1 || H < 128 || (H & 127) || H > 1024 ||
        ndir < 1 || ndir > 2 || (gi_ld_m & 3) || (gi_ld_t & 3) || (y_ld_m & 3) || (y_ld_t & 3))
        return CVC_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    GruPArgs a{};
    a.wp = wp; a.w_stride = (long long)(H / 8) * (H / 4) * 128;
    a.gi = gi; a.gi_ld_m = gi_ld_m; a.gi_ld_t = gi_ld_t; a.b_ih = b_ih; a.b_hh = b_hh;
    a.M = M; a.F = F; a.H = H; a.Kp = H; a.hq = hq; a.h_stride = (long long)H * 64;
    a.y = y; a.y_ld_m = y_ld_m; a.y_ld_t = y_ld_t; a.sync = sync; a.spin_limit = 1u << 20;
    a.gates = gates; a.g_ld_m = g_ld_m; a.g_ld_t = g_ld_t;
    const long long n = a.h_stride * ndir;                          // slot 0 = h0 = 0
    const long long nz = n > SYNC_WORDS ? n : SYNC_WORDS;
    hipLaunchKernelGGL(gru_zero_kernel, dim3((unsigned)((nz + 255) / 256)), dim3(256), 0, st, hq, n, sync);
    // 8 waves (K / 8 per wave) when K is a multiple of 256, else 4 waves (K / 4 per wave)
    const bool w8 = cvc_gru_waves8 && (H % 256) == 0;
    const int NC = w8 ? H / 256 : H / 128;
#define CVC_GRU_P(NH_, MT_, NC_, NW_) return launch_persistent<NH_, MT_, NC_, NW_>(a, ndir, st)
#define CVC_GRU_NC(NH_, MT_)                                                                                              \
    if (w8) {                                                                                                             \
        switch (NC) { case 1: CVC_GRU_P(NH_, MT_, 1, 8); case 2: CVC_GRU_P(NH_, MT_, 2, 8); case 3: CVC_GRU_P(NH_, MT_, 3, 8); \
                      default: CVC_GRU_P(NH_, MT_, 4, 8); }                                                               \
    }                                                                                                                     \
    switch (NC) { case 1: CVC_GRU_P(NH_, MT_, 1, 4); case 2: CVC_GRU_P(NH_, MT_, 2, 4); case 3: CVC_GRU_P(NH_, MT_, 3, 4);     \
                  case 4: CVC_GRU_P(NH_, MT_, 4, 4); case 5: CVC_GRU_P(NH_, MT_, 5, 4); case 6: CVC_GRU_P(NH_, MT_, 6, 4);     \
                  case 7: CVC_GRU_P(NH_, MT_, 7, 4); default: CVC_GRU_P(NH_, MT_, 8, 4); }
    if (M <= 32) { CVC_GRU_NC(1, 1) }
    if (cvc_gru_halves) { CVC_GRU_NC(2, 1) }        // two interleaved 32-clip recurrences
    CVC_GRU_NC(1, 2)
#undef CVC_GRU_NC
#undef CVC_GRU_P
}

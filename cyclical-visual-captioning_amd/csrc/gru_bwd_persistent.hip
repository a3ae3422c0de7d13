// Persistent form of the GRU backward recurrence (cvc_gru_seq_bwd launches three kernels per time step; 70 of the 93 ms of a
// config-2 forward + backward are those 1 440 launches per layer).  One cooperative launch per layer, built like the forward
// (gru_persistent.hip): a workgroup owns 8 hidden units of one direction for all F steps.
//   per step s (the forward direction walks t = F-1 .. 0, the reverse direction t = 0 .. F-1):
//     1. wait until every workgroup of the direction has published dgh of step s - 1 (spread arrival counters);
//     2. dh_mm[m, j] = sum_k dgh_{s-1}[m, k] W_hh[k, j] for its 8 units j: the 3H x 8 weight columns live in registers as split
//        bf16 terms (A operand of v_mfma_f32_16x16x32_bf16: 8 units + 8 zero rows), dgh_{s-1} of ALL units is read from its
//        exchange slot ([3H/8][64 clips][8], written once, read after the barrier: no cache can hold a stale copy), 8 waves split K;
//     3. thread (clip m, unit j): dh = dY_t + z-carried part (a register) + dh_mm; gate gradients; dgi_t, dgh_t rows to global
//        memory (the dense dW / dX products of the host side read them), dgh_t of the own units to the step's exchange slot
//        with write-through stores; arrival.
#include "cvc_common.h"
#include <stdlib.h>
#include "gemm_split.h"

namespace {

constexpr int BCNT = 32, BCNT_STRIDE = 1024, BSYNC_ERR = 4;
constexpr long long BSYNC_WORDS = BSYNC_ERR + 8 + 2LL * BCNT * BCNT_STRIDE;

struct GruBArgs {
    const float* wt;                              // W_hh^T packed [ndir][H/8][3H/8][8 units][8 k]
    const float* dy; long long dy_ld_m, dy_ld_t;  // dL/dh rows: (m, t) at + m * ld_m + t * ld_t, columns [ndir][H]
    const float* gates; long long g_ld_m, g_ld_t; // (r, z, n, hn): columns [ndir][4][H]
    const float* y; long long y_ld_m, y_ld_t;     // forward outputs h_t: columns [ndir][H]
    float* dgi; float* dgh;                       // [F * M rows (t * M + m), ndir * 3H]
    float* slots;                                 // exchange: [F][ndir][3H/8][64][8]
    unsigned* sync;
    int M, F, H;
    unsigned spin_limit;
};

using f32x4v = __attribute__((ext_vector_type(4))) float;

__device__ __forceinline__ f32x4v mfma16(const u32x4 a, const u32x4 b, const f32x4v c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// NKS = 32-k steps per wave: 3H = 256 * NKS
template <int NKS>
__global__ __launch_bounds__(512, 1) void gru_bwd_persistent_kernel(GruBArgs a) {
    __shared__ float red[8][8][64];               // [wave][unit][clip]
    __shared__ int gave_up;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int dir = blockIdx.y, blk = blockIdx.x, ndir = gridDim.y, H = a.H, M = a.M, F = a.F;
    const int K = 3 * H, ngrp = K >> 3;
    const unsigned nblk = gridDim.x;
    unsigned* counter = a.sync + BSYNC_ERR + 8 + (size_t)(dir * BCNT) * BCNT_STRIDE;

    // ---- weights: lane (row r = lane & 15, k group = lane >> 4) of k step ks holds 8 consecutive k of unit r (rows 8..15 zero)
    const int wr = lane & 15, wg = lane >> 4;
    Split3 W[NKS];
    {
        const float* wl = a.wt + ((size_t)dir * (H / 8) + blk) * (size_t)ngrp * 64;
#pragma unroll
        for (int j = 0; j < NKS; ++j) {
            const int ks = wave + 8 * j;
            f32x4 q0 = {0, 0, 0, 0}, q1 = {0, 0, 0, 0};
            if (wr < 8) {
                const float* p = wl + ((size_t)(ks * 4 + wg) * 8 + wr) * 8;
                q0 = ld4(p); q1 = ld4(p + 4);
            }
            W[j] = split8(q0, q1);
        }
    }

    // ---- elementwise role: thread <-> (clip em, unit ej of this block), fixed for the whole sequence
    const int em = tid >> 3, ejr = tid & 7;
    const int ej = blk * 8 + ejr;
    const bool ework = em < M;
    float dh_part = 0.f;                           // dh z of the previous processed step (own unit): never leaves the thread
    const long long row_ld = (long long)ndir * 3 * H;

    if (tid == 0) gave_up = 0;
    __syncthreads();
    for (int s = 0; s < F; ++s) {
        const long long t = dir == 0 ? F - 1 - s : s;
        const long long tp = dir == 0 ? t - 1 : t + 1;
        // operands of the gate arithmetic: independent of the other workgroups, requested before the wait
        float dyv = 0.f, rg = 0.f, zg = 0.f, ng = 0.f, hn = 0.f, hp = 0.f;
        if (ework) {
            dyv = a.dy[(size_t)em * a.dy_ld_m + t * a.dy_ld_t + (size_t)dir * H + ej];
            const float* gp = a.gates + (size_t)em * a.g_ld_m + t * a.g_ld_t + (size_t)dir * 4 * H + ej;
            rg = gp[0]; zg = gp[H]; ng = gp[2 * H]; hn = gp[3 * H];
            if (tp >= 0 && tp < F) hp = a.y[(size_t)em * a.y_ld_m + tp * a.y_ld_t + (size_t)dir * H + ej];
        }
        float dh_mm = 0.f;
        if (s > 0) {
            // ---- wait for dgh of step s - 1
            if (wave == 0) {
                const unsigned target = nblk * (unsigned)s;
                unsigned it = 0;
                for (;;) {
                    unsigned v = lane < BCNT ? __hip_atomic_load(counter + (size_t)lane * BCNT_STRIDE, __ATOMIC_RELAXED,
                                                                 __HIP_MEMORY_SCOPE_AGENT) : 0u;
#pragma unroll
                    for (int o = 1; o < BCNT; o <<= 1) v += __shfl_xor(v, o, 64);
                    v = __builtin_amdgcn_readfirstlane(v);
                    if (v >= target) break;
                    if (++it > a.spin_limit || __hip_atomic_load(a.sync + BSYNC_ERR, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                        if (lane == 0) {
                            __hip_atomic_store(a.sync + BSYNC_ERR, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            gave_up = 1;
                        }
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
            }
            __syncthreads();
            if (gave_up) return;
            // ---- dh_mm = dgh_{s-1} W_hh[:, own units]: this wave's k steps, 4 tiles of 16 clips
            const float* X = a.slots + ((size_t)(s - 1) * ndir + dir) * (size_t)ngrp * 512 + ((size_t)wg * 64 + (lane & 15)) * 8;
            f32x4v acc[4];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) acc[ct] = f32x4v{0, 0, 0, 0};
            // phases of (k step, two clip tiles), two in flight: 32 registers next to the 144 of the weights
            f32x4 xb[2][2][2];
            auto loadx = [&](f32x4 (&buf)[2][2], const int ph) __attribute__((always_inline)) {
                const float* p = X + (size_t)(wave + 8 * (ph >> 1)) * 4 * 512 + (ph & 1) * 256;
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2) { buf[c2][0] = ld4(p + c2 * 128); buf[c2][1] = ld4(p + c2 * 128 + 4); }
            };
            loadx(xb[0], 0);
#pragma unroll
            for (int ph = 0; ph < 2 * NKS; ++ph) {
                if (ph + 1 < 2 * NKS) loadx(xb[(ph + 1) & 1], ph + 1);
                __builtin_amdgcn_sched_barrier(0);
                const int j = ph >> 1;
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2) {
                    const int ct = (ph & 1) * 2 + c2;
                    const Split3 Xs = split8(xb[ph & 1][c2][0], xb[ph & 1][c2][1]);
                    acc[ct] = mfma16(W[j].mid, Xs.mid, acc[ct]);
                    acc[ct] = mfma16(W[j].lo, Xs.hi, acc[ct]);
                    acc[ct] = mfma16(W[j].hi, Xs.lo, acc[ct]);
                    acc[ct] = mfma16(W[j].mid, Xs.hi, acc[ct]);
                    acc[ct] = mfma16(W[j].hi, Xs.mid, acc[ct]);
                    acc[ct] = mfma16(W[j].hi, Xs.hi, acc[ct]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // D[unit = 4 * (lane >> 4) + r][clip = ct * 16 + (lane & 15)]: units 0..7 live in lanes 0..31
            if (wg < 2) {
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                    for (int r = 0; r < 4; ++r) red[wave][4 * wg + r][ct * 16 + (lane & 15)] = acc[ct][r];
            }
            __syncthreads();
            dh_mm = ((red[0][ejr][em] + red[1][ejr][em]) + (red[2][ejr][em] + red[3][ejr][em])) +
                    ((red[4][ejr][em] + red[5][ejr][em]) + (red[6][ejr][em] + red[7][ejr][em]));
        }
        // ---- gate gradients of step s for (clip em, unit ej)
        if (ework) {
            const float dh = (dyv + dh_part) + dh_mm;
            const float dn = dh * (1.f - zg) * (1.f - ng * ng);
            const float dz = dh * (hp - ng) * zg * (1.f - zg);
            const float dr = dn * hn * rg * (1.f - rg);
            const float dnr = dn * rg;
            dh_part = dh * zg;
            float* gi = a.dgi + (size_t)(t * M + em) * row_ld + (size_t)dir * 3 * H + ej;
            gi[0] = dr; gi[H] = dz; gi[2 * H] = dn;
            float* gh = a.dgh + (size_t)(t * M + em) * row_ld + (size_t)dir * 3 * H + ej;
            gh[0] = dr; gh[H] = dz; gh[2 * H] = dnr;
            if (s + 1 < F) {
                // own units of the exchange slot: group (g H + 8 blk) / 8 of gate g, element (clip, unit); straight to memory
                float* sl = a.slots + ((size_t)s * ndir + dir) * (size_t)ngrp * 512 + ((size_t)blk * 64 + em) * 8 + ejr;
                const size_t gs = (size_t)(H / 8) * 512;
                asm volatile("global_store_dword %0, %1, off sc0 sc1\n\ts_nop 0" ::"v"(sl), "v"(dr) : "memory");
                asm volatile("global_store_dword %0, %1, off sc0 sc1\n\ts_nop 0" ::"v"(sl + gs), "v"(dz) : "memory");
                asm volatile("global_store_dword %0, %1, off sc0 sc1\n\ts_nop 0" ::"v"(sl + 2 * gs), "v"(dnr) : "memory");
            }
        }
        if (s + 1 < F) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                              // every wave's slot stores are acknowledged; `red` may be rewritten
            if (tid == 0)
                __hip_atomic_fetch_add(counter + (size_t)(blk % BCNT) * BCNT_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

__global__ __launch_bounds__(256) void gru_bwd_zero_kernel(unsigned* sync) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t < BSYNC_WORDS) sync[t] = 0u;
}

template <int NKS>
int launch_bwd(GruBArgs& a, int ndir, hipStream_t st) {
    void* params[] = {&a};
    const dim3 grid(a.H / 8, ndir);
    int per_cu = 0, devid = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)gru_bwd_persistent_kernel<NKS>, 512, 0) != hipSuccess ||
        hipGetDevice(&devid) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, devid) != hipSuccess) {
        (void)hipGetLastError();
        return CVC_E_BADARG;
    }
    if ((long long)per_cu * cus < (long long)grid.x * grid.y) return CVC_E_BADARG;
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
        if (hipLaunchCooperativeKernel((const void*)gru_bwd_persistent_kernel<NKS>, grid, dim3(512), params, 0, st) != hipSuccess) {
            (void)hipGetLastError();
            return CVC_E_BADARG;
        }
    } else {
        hipLaunchKernelGGL((gru_bwd_persistent_kernel<NKS>), grid, dim3(512), 0, st, a);
    }
    return cvc_launch_status();
}

}  // namespace

extern "C" int cvc_gru_bwd_persistent_sync_words(void) { return (int)BSYNC_WORDS; }

// Same inputs and outputs as cvc_gru_seq_bwd except: wt = W_hh^T packed [ndir][H/8][3H/8][8][8] (cvc.gru.pack_gru_weights_t),
// slots = F * ndir * 3H * 64 floats of exchange memory, sync = cvc_gru_bwd_persistent_sync_words() words (word 4 non-zero
// afterwards = barrier time-out, outputs invalid: use cvc_gru_seq_bwd).  H % 256 == 0, H <= 1024, M <= 64; CVC_E_BADARG
// without launching otherwise.
extern "C" int cvc_gru_seq_bwd_persistent(const float* dy, long long dy_ld_m, long long dy_ld_t, const float* gates, long long g_ld_m,
                                          long long g_ld_t, const float* y, long long y_ld_m, long long y_ld_t, const float* wt,
                                          int M, int F, int H, int ndir, float* dgi, float* dgh, float* slots, unsigned* sync,
                                          cvc_stream_t stream) {
    if (!dy || !gates || !y || !wt || !dgi || !dgh || !slots || !sync || M < 1 || M > 64 || F < 1 || H < 256 || (H & 255) ||
        H > 1024 || ndir < 1 || ndir > 2)
        return CVC_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    GruBArgs a{};
    a.wt = wt; a.dy = dy; a.dy_ld_m = dy_ld_m; a.dy_ld_t = dy_ld_t; a.gates = gates; a.g_ld_m = g_ld_m; a.g_ld_t = g_ld_t;
    a.y = y; a.y_ld_m = y_ld_m; a.y_ld_t = y_ld_t; a.dgi = dgi; a.dgh = dgh; a.slots = slots; a.sync = sync;
    a.M = M; a.F = F; a.H = H; a.spin_limit = 1u << 20;
    hipLaunchKernelGGL(gru_bwd_zero_kernel, dim3((unsigned)((BSYNC_WORDS + 255) / 256)), dim3(256), 0, st, sync);
    switch (3 * H / 256) {
        case 3: return launch_bwd<3>(a, ndir, st);
        case 6: return launch_bwd<6>(a, ndir, st);
        case 9: return launch_bwd<9>(a, ndir, st);
        case 12: return launch_bwd<12>(a, ndir, st);
        default: return CVC_E_BADARG;
    }
}

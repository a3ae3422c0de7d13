// What bounds the gate GEMM's activation ingress?  256 workgroups x 8 waves read an L2-resident activation block with the access
// pattern of skinny_gemm_packed_kernel (lane = (row i, k half kh): 8 dwordx4 loads per 32-k chunk, chunks dealt to waves, every
// workgroup starting at a different chunk), optionally next to a non-temporal weight stream from HBM (4 dwordx4 loads per chunk).
//   copies = 1: every workgroup reads the SAME 1.5 MB (as the kernel does);  copies = c: workgroup b reads copy b % c
//   (c = 8: one copy per XCD slot pattern, c = 32: four workgroups per copy and XCD)
// Build: hipcc --offload-arch=gfx950 -O3 tools/l2_ingress.hip -o l2_ingress
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <bool WEIGHTS, bool ACTS, int DEPTH>
__global__ __launch_bounds__(512) void ingress(const float* __restrict__ xq, size_t copy_stride, int copies, const float* __restrict__ wp,
                                               int nchunk, int rounds, float* out) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, kh = lane >> 5;
    const float* x0 = xq + (size_t)(blockIdx.x % copies) * copy_stride + (size_t)i * 4 + kh * 4 * 256;
    const float* w0 = wp + (size_t)blockIdx.x * nchunk * 8 * 128 + (size_t)i * 4 + kh * 4 * 128;
    const int n_my = nchunk / 8;
    const int rot = (blockIdx.x * 5) % n_my;
    f32x4 s = {0, 0, 0, 0};
    for (int r = 0; r < rounds; ++r) {
        f32x4 ring[DEPTH][12];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
#pragma unroll
            for (int q = 0; q < 12; ++q) ring[d][q] = f32x4{0, 0, 0, 0};
        auto load = [&](f32x4 (&f)[12], int j) __attribute__((always_inline)) {
            int jr = j + rot; jr = jr >= n_my ? jr - n_my : jr;
            const int c = wave + 8 * jr;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (WEIGHTS) f[q] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(w0 + (size_t)c * 8 * 128 + q * 128));
                if (ACTS) {
                    f[4 + 2 * q] = *reinterpret_cast<const f32x4*>(x0 + (size_t)c * 8 * 256 + q * 256);
                    f[5 + 2 * q] = *reinterpret_cast<const f32x4*>(x0 + (size_t)c * 8 * 256 + q * 256 + 128);
                }
            }
        };
#pragma unroll
        for (int d = 0; d < DEPTH - 1; ++d) load(ring[d], d);
        for (int j = 0; j + DEPTH <= n_my + DEPTH - 1; j += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const int jn = j + d + DEPTH - 1;
                load(ring[(d + DEPTH - 1) % DEPTH], jn < n_my ? jn : n_my - 1);
#pragma unroll
                for (int q = 0; q < 12; ++q) s += ring[d][q];
            }
        }
    }
    out[(size_t)blockIdx.x * 512 + tid] = s[0] + s[1] + s[2] + s[3];
}

// ---- is the serialisation of HBM-sourced and L2-sourced loads per WAVE (in-order return behind vmcnt) or per CU?  Same bytes as
// `ingress<true, true>`, but the roles are split: waves 0..NWW-1 stream ALL the weights of the workgroup (nt, HBM), the other waves
// read ALL the activations (L2 hits).  If the two kinds only wait for each other inside a wave, this runs at max(weights, acts).
template <int NWW, int DEPTH>
__global__ __launch_bounds__(512) void ingress_roles(const float* __restrict__ xq, const float* __restrict__ wp, int nchunk, float* out) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, kh = lane >> 5;
    const bool wrole = wave < NWW;
    const int nw = wrole ? NWW : 8 - NWW, w = wrole ? wave : wave - NWW;
    const float* x0 = xq + (size_t)i * 4 + kh * 4 * 256;
    const float* w0 = wp + (size_t)blockIdx.x * nchunk * 8 * 128 + (size_t)i * 4 + kh * 4 * 128;
    const int n_my = nchunk / nw;
    const int rot = (blockIdx.x * 5) % n_my;
    f32x4 s = {0, 0, 0, 0};
    if (wrole) {
        f32x4 ring[DEPTH][4];
        auto load = [&](f32x4 (&f)[4], int j) __attribute__((always_inline)) {
            int jr = j + rot; jr = jr >= n_my ? jr - n_my : jr;
            const int c = w + nw * jr;
#pragma unroll
            for (int q = 0; q < 4; ++q) f[q] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(w0 + (size_t)c * 8 * 128 + q * 128));
        };
#pragma unroll
        for (int d = 0; d < DEPTH - 1; ++d) load(ring[d], d);
        for (int j = 0; j < n_my; j += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const int jn = j + d + DEPTH - 1;
                load(ring[(d + DEPTH - 1) % DEPTH], jn < n_my ? jn : n_my - 1);
#pragma unroll
                for (int q = 0; q < 4; ++q) s += ring[d][q];
            }
        }
    } else {
        f32x4 ring[DEPTH][8];
        auto load = [&](f32x4 (&f)[8], int j) __attribute__((always_inline)) {
            int jr = j + rot; jr = jr >= n_my ? jr - n_my : jr;
            const int c = w + nw * jr;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f[2 * q] = *reinterpret_cast<const f32x4*>(x0 + (size_t)c * 8 * 256 + q * 256);
                f[2 * q + 1] = *reinterpret_cast<const f32x4*>(x0 + (size_t)c * 8 * 256 + q * 256 + 128);
            }
        };
#pragma unroll
        for (int d = 0; d < DEPTH - 1; ++d) load(ring[d], d);
        for (int j = 0; j < n_my; j += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const int jn = j + d + DEPTH - 1;
                load(ring[(d + DEPTH - 1) % DEPTH], jn < n_my ? jn : n_my - 1);
#pragma unroll
                for (int q = 0; q < 8; ++q) s += ring[d][q];
            }
        }
    }
    out[(size_t)blockIdx.x * 512 + tid] = s[0] + s[1] + s[2] + s[3];
}

// ---- skeleton of the K-split kernel (gemm_packed_ks.hip): 8 waves, each streams ITS 32 weight rows (4 nt dwordx4 loads per chunk);
// XA activation loads per wave and chunk (1 = the K-split kernel's share of the common chunk), optional barrier per chunk (the
// lock step the LDS hand-over needs), NM dependent-free bf16 MFMAs and NV VALU ops per chunk (24 / ~130 in the real kernel)
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
template <int XA, bool SYNC, int NM, int NV, int DEPTH>
__global__ __launch_bounds__(512) void ks_skeleton(const float* __restrict__ xq, const float* __restrict__ wp, int nchunk, float* out) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ks = blockIdx.x & 7, tile = blockIdx.x >> 3;
    const int n = nchunk / 8, c_lo = ks * n;
    const float* w0 = wp + ((size_t)(tile * 8 + wave) * nchunk + c_lo) * 8 * 128 + (size_t)lane * 4;
    const float* x0 = xq + ((size_t)(c_lo * 8 + wave) * 64 + lane) * 4;
    f32x16 acc[2] = {};
    f32x4 s = {0, 0, 0, 0};
    f32x4 wr[DEPTH][4], xr[DEPTH][XA > 0 ? XA : 1];
    auto load = [&](int slot, int c) __attribute__((always_inline)) {
        c = c < n ? c : n - 1;
#pragma unroll
        for (int q = 0; q < XA; ++q) xr[slot][q] = *reinterpret_cast<const f32x4*>(x0 + (size_t)c * 8 * 256 + q * 64);
#pragma unroll
        for (int q = 0; q < 4; ++q) wr[slot][q] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(w0 + (size_t)c * 8 * 128 + q * 256));
    };
#pragma unroll
    for (int d = 0; d < DEPTH - 1; ++d) load(d, d);
    for (int j = 0; j < n; j += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            load((d + DEPTH - 1) % DEPTH, j + d + DEPTH - 1);
            __builtin_amdgcn_sched_barrier(0);
            f32x4 t = wr[d][0] + wr[d][1] + wr[d][2] + wr[d][3];
#pragma unroll
            for (int q = 0; q < XA; ++q) t += xr[d][q];
#pragma unroll
            for (int v = 0; v < NV / 4; ++v) { t = t * 1.0001f + s; }             // 4 VALU (pk or not) per step, dependent chain of 2
            s += t;
            if (NM > 0) {
                bf16x8 a8 = __builtin_bit_cast(bf16x8, t), b8 = __builtin_bit_cast(bf16x8, s);
#pragma unroll
                for (int m = 0; m < NM; ++m) acc[m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc[m & 1], 0, 0, 0);
            }
            if (SYNC) __syncthreads();
        }
    }
    out[(size_t)blockIdx.x * 512 + tid] = s[0] + s[1] + s[2] + s[3] + acc[0][0] + acc[1][5];
}

int main() {
    const int K = 6144, nchunk = K / 32, nwg = 256, copies_max = 32;
    const size_t xbytes = (size_t)K / 4 * 64 * 4 * 4;               // 1.5 MB
    float *xq, *wp, *out, *flush;
    hipMalloc(&xq, xbytes * copies_max);
    hipMalloc(&wp, (size_t)nwg * nchunk * 8 * 128 * 4);             // 201 MB
    hipMalloc(&out, nwg * 512 * 4);
    hipMalloc(&flush, 512u << 20);
    hipMemset(xq, 0, xbytes * copies_max);
    hipMemset(wp, 0, (size_t)nwg * nchunk * 8 * 128 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](auto launch, int reps, bool flush_first) {
        float tot = 0;
        for (int r = 0; r < reps + 2; ++r) {
            if (flush_first) hipMemsetAsync(flush, r, 512u << 20);
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (r >= 2) tot += ms;
        }
        return tot / reps * 1e3;
    };
    const double act_mb = nwg * (double)xbytes / 1e6, w_mb = (double)nwg * nchunk * 8 * 128 * 4 / 1e6;
    for (int copies : {1, 2, 8, 32}) {
        const int rounds = 4;
        float us = time([&] { hipLaunchKernelGGL((ingress<false, true, 3>), dim3(nwg), dim3(512), 0, 0, xq, xbytes / 4, copies, wp, nchunk, rounds, out); }, 10, false);
        printf("activations only, %2d copies: %7.1f us per pass  (%.1f TB/s L2->CU, %.1f B/clk/CU at 2.4 GHz)\n", copies, us / rounds,
               act_mb / (us / rounds) , act_mb / (us / rounds) * 1e6 / 256 / 2400.0);
    }
    {
        float us = time([&] { hipLaunchKernelGGL((ingress<true, false, 3>), dim3(nwg), dim3(512), 0, 0, xq, xbytes / 4, 1, wp, nchunk, 1, out); }, 10, true);
        printf("weights only (nt, HBM):      %7.1f us  (%.2f TB/s)\n", us, w_mb / us);
    }
    for (int copies : {1, 8, 32}) {
        float us = time([&] { hipLaunchKernelGGL((ingress<true, true, 3>), dim3(nwg), dim3(512), 0, 0, xq, xbytes / 4, copies, wp, nchunk, 1, out); }, 10, true);
        printf("weights + activations, %2d copies: %7.1f us  (weights %.2f TB/s; L2->CU %.1f TB/s)\n", copies, us, w_mb / us, (act_mb + w_mb) / us);
    }
    {
        float us = time([&] { hipLaunchKernelGGL((ingress<true, true, 4>), dim3(nwg), dim3(512), 0, 0, xq, xbytes / 4, 1, wp, nchunk, 1, out); }, 10, true);
        printf("weights + activations, depth 4: %7.1f us\n", us);
    }
    {
        float us = time([&] { hipLaunchKernelGGL((ingress_roles<4, 3>), dim3(nwg), dim3(512), 0, 0, xq, wp, nchunk, out); }, 10, true);
        printf("roles split, 4 weight waves (depth 3) + 4 activation waves: %7.1f us\n", us);
        us = time([&] { hipLaunchKernelGGL((ingress_roles<4, 6>), dim3(nwg), dim3(512), 0, 0, xq, wp, nchunk, out); }, 10, true);
        printf("roles split, 4 weight waves (depth 6) + 4 activation waves: %7.1f us\n", us);
        us = time([&] { hipLaunchKernelGGL((ingress_roles<6, 4>), dim3(nwg), dim3(512), 0, 0, xq, wp, nchunk, out); }, 10, true);
        printf("roles split, 6 weight waves (depth 4) + 2 activation waves: %7.1f us\n", us);
    }
    printf("K-split skeleton (256 gate rows x K/8 per workgroup), lang cell 201 MB of weights:\n");
#define RUN(XA, SYNC, NM, NV, D) { float us = time([&] { hipLaunchKernelGGL((ks_skeleton<XA, SYNC, NM, NV, D>), dim3(nwg), dim3(512), 0, 0, xq, wp, nchunk, out); }, 10, true); \
        printf("  act loads %d, barrier %d, mfma %2d, valu %3d, depth %d: %6.1f us (%.2f TB/s)\n", XA, (int)SYNC, NM, NV, D, us, w_mb / us); }
    RUN(0, false, 0, 0, 4) RUN(1, false, 0, 0, 4) RUN(1, true, 0, 0, 4) RUN(1, false, 24, 0, 4) RUN(1, false, 24, 128, 4)
    RUN(1, true, 24, 128, 4) RUN(1, true, 24, 128, 3) RUN(1, true, 24, 128, 5) RUN(1, false, 24, 128, 5) RUN(0, false, 24, 128, 4) RUN(1, true, 24, 64, 4)
    RUN(1, false, 0, 128, 4) RUN(2, false, 24, 128, 4)
    return 0;
}

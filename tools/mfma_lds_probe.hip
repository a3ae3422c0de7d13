// What does the computing wave of the tile GEMM (csrc/gemm_tile.hip, tile_gemm_ld2_kernel<MH = 5>) lose against the pure MFMA rate?
// One workgroup per CU, 4 waves (one per SIMD), LDS pre-filled; per "k step" a wave reads fragments with ds_read_b128 and issues
// 60 v_mfma_f32_32x32x16_bf16 (2 weight blocks x 5 row tiles x 6 split terms), as the kernel does, in variants:
//   0  MFMAs only (operands stay in registers)                     1  + the kernel's 21 fragment reads per k step
//   2  1 + one s_barrier per k step among the 4 waves               3  2 + 4 more waves per workgroup that only take the barrier
//   4  1 with 12 fragment reads (the row-tile reads only)           5 - 8  + four loader waves (ds_write / global loads / both)
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_lds_probe.hip -o mfma_lds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

__device__ __forceinline__ f32x16 mfma(const u32x4 a, const u32x4 b, const f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <int VAR>
__global__ __launch_bounds__(512) void probe(float* out, int ksteps, const u32x4* __restrict__ gsrc = nullptr, size_t gmask = 0) {
    constexpr int MH = 5, NF = (4 + 2 * MH) * 3, STAGE = NF * 1024;
    __shared__ __attribute__((aligned(16))) char lds[3 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 3 * STAGE / 4; i += blockDim.x) reinterpret_cast<unsigned*>(lds)[i] = 0x3f803f80u + (i & 7);
    __syncthreads();
    if (wave >= 4) {                                   // (variant 3: the loader waves' side of the per-step barrier)
        if (VAR >= 5) {
            // loader waves as in tile_gemm_ld2_kernel<.., REGLOAD>: 11 pieces (1 KiB each) per wave and k step.  5: ds_write only (data
            // from registers), 6: global loads only (results consumed by a dummy), 7: loads + ds_write, 8: 7 from a small (L2-resident) range
            const int lw = wave - 4;
            u32x4 r[11];
            for (int j = 0; j < 11; ++j) r[j] = u32x4{(unsigned)j, 1u, 2u, 3u};
            size_t off = ((size_t)blockIdx.x * 4 + lw) * 11 * 64 + lane;
            unsigned sink = 0;
            for (int s = 0; s < ksteps; ++s) {
                if (VAR >= 6) {
#pragma unroll
                    for (int j = 0; j < 11; ++j) r[j] = gsrc[(off + (size_t)j * 64) & gmask];
                    off += 4 * 11 * 64 * 256;
                }
                if (VAR == 5 || VAR >= 7) {
#pragma unroll
                    for (int j = 0; j < 11; ++j) *reinterpret_cast<u32x4*>(lds + ((s + 2) % 3) * STAGE + ((lw + 4 * j) % NF) * 1024 + lane * 16) = r[j];
                } else {
#pragma unroll
                    for (int j = 0; j < 11; ++j) sink += r[j].x;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            if (sink == 12345u) out[0] = 1.f;
            return;
        }
        for (int s = 0; s < ksteps; ++s) __builtin_amdgcn_s_barrier();
        return;
    }
    const int wp = wave & 1, mh = wave >> 1;
    f32x16 acc[2][MH];
    for (int b = 0; b < 2; ++b) for (int t = 0; t < MH; ++t) for (int r = 0; r < 16; ++r) acc[b][t][r] = 0.f;
    u32x4 w[2][3], x[2][3];
    auto frag = [&](int buf, int f) __attribute__((always_inline)) { return *reinterpret_cast<const u32x4*>(lds + buf * STAGE + lane * 16 + f * 1024); };
    for (int pl = 0; pl < 3; ++pl) { w[0][pl] = frag(0, (2 * wp) * 3 + pl); w[1][pl] = frag(0, (2 * wp + 1) * 3 + pl); x[0][pl] = frag(0, 12 + mh * MH * 3 + pl); x[1][pl] = x[0][pl]; }
    auto mma = [&](const u32x4* xt, const int t) __attribute__((always_inline)) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            acc[b][t] = mfma(xt[1], w[b][1], acc[b][t]); acc[b][t] = mfma(xt[0], w[b][2], acc[b][t]); acc[b][t] = mfma(xt[2], w[b][0], acc[b][t]);
            acc[b][t] = mfma(xt[0], w[b][1], acc[b][t]); acc[b][t] = mfma(xt[1], w[b][0], acc[b][t]); acc[b][t] = mfma(xt[0], w[b][0], acc[b][t]);
        }
    };
    int buf = 0;
    for (int s = 0; s < ksteps; ++s) {
        const int nbuf = buf == 2 ? 0 : buf + 1;
#pragma unroll
        for (int t = 0; t + 1 < MH; ++t) {
            if (VAR >= 1) {
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) x[(t + 1) & 1][pl] = frag(buf, 12 + (mh * MH + t + 1) * 3 + pl);
            }
            __builtin_amdgcn_sched_barrier(0);
            mma(x[t & 1], t);
            __builtin_amdgcn_sched_barrier(0);
        }
        u32x4 xl[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) xl[pl] = x[(MH - 1) & 1][pl];
        if (VAR >= 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        mma(xl, MH - 1);
        __builtin_amdgcn_sched_barrier(0);
        if (VAR == 2 || VAR == 3 || VAR >= 5) __builtin_amdgcn_s_barrier();
        if (VAR >= 1 && VAR != 4) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) { w[0][pl] = frag(nbuf, (2 * wp) * 3 + pl); w[1][pl] = frag(nbuf, (2 * wp + 1) * 3 + pl); x[0][pl] = frag(nbuf, 12 + mh * MH * 3 + pl); }
        }
        buf = nbuf;
    }
    float sacc = 0.f;
    for (int b = 0; b < 2; ++b) for (int t = 0; t < MH; ++t) sacc += acc[b][t][0] + acc[b][t][9];
    out[blockIdx.x * 256 + (tid & 255)] = sacc;
}

template <int VAR>
void run(const char* what, float* out, const u32x4* gsrc = nullptr, size_t gmask = 0) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int ksteps = 2000, threads = VAR >= 3 ? 512 : 256;
    probe<VAR><<<256, threads>>>(out, 50, gsrc, gmask);
    (void)hipEventRecord(e0);
    probe<VAR><<<256, threads>>>(out, ksteps, gsrc, gmask);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double per_step_ns = ms * 1e6 / ksteps;
    printf("%-62s %7.1f ns per k step = %5.0f cycles at 2.4 GHz (60 MFMAs = 1920): %4.0f TFLOP/s bf16\n", what, per_step_ns, per_step_ns * 2.4,
           2.0 * 32 * 32 * 16 * 60 * 4 * 256 * (double)ksteps / (ms * 1e-3) / 1e12);
}

int main() {
    float* out;
    (void)hipMalloc(&out, 256 * 256 * sizeof(float));
    run<0>("0 MFMAs only", out);
    run<1>("1 + 21 ds_read_b128 per k step (the kernel's reads)", out);
    run<2>("2 + s_barrier per k step (4 waves)", out);
    run<3>("3 + 4 more waves that only take the barrier", out);
    run<4>("4 row-tile reads only (12 per k step)", out);
    u32x4* g;
    const size_t n16 = (size_t)1 << 27;                 // 2 GiB of 16-byte elements
    (void)hipMalloc(&g, n16 * 16);
    (void)hipMemset(g, 1, n16 * 16);
    run<5>("5 + loader waves: 42 KiB of ds_write_b128 per k step", out);
    run<6>("6 + loader waves: 42 KiB of global loads per k step (HBM)", out, g, n16 - 1);
    run<7>("7 + loader waves: loads (HBM) + ds_write", out, g, n16 - 1);
    run<8>("8 + loader waves: loads (32 MiB range: L2 / MALL) + ds_write", out, g, ((size_t)1 << 21) - 1);
    return 0;
}

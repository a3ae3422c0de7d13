// Microbenchmarks for the two roofs the decode path is priced against (run on the GPU box):
//   fp32 MFMA issue rate (v_mfma_f32_32x32x2_f32, 1/2 waves per SIMD, all CUs), bf16 MFMA rate (32x32x16) and read-only HBM
//   streaming (dwordx4 loads, 1 GiB buffer > Infinity Cache).  Build: hipcc --offload-arch=gfx950 -O3 tools/peaks.hip -o peaks
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

__global__ __launch_bounds__(512) void mfma_loop(float* out, int iters, float a, float b) {
    f32x16 acc0 = {0}, acc1 = {0};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc1, 0, 0, 0);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc0[0] + acc1[3];
}

// bf16 32x32x16 MFMA with register operands only: the sustained rate the split-product GEMMs are priced against.  `dep`
// accumulator tiles per wave: 1 = every MFMA depends on the previous one, 4 = four independent chains
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
template <int NACC>
__global__ __launch_bounds__(512) void mfma_bf16_loop(float* out, int iters, float seed) {
    f32x16 acc[NACC];
    for (int t = 0; t < NACC; ++t)
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(seed + e + threadIdx.x); b[e] = (__bf16)(seed - e); }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 32 / NACC; ++u)
#pragma unroll
            for (int t = 0; t < NACC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[t], 0, 0, 0);
    }
    float s = 0.f;
    for (int t = 0; t < NACC; ++t) s += acc[t][0] + acc[t][7];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
void run_bf16(float* out, hipEvent_t e0, hipEvent_t e1, int iters = 4000) {
    for (int wps = 1; wps <= 2; ++wps) {
        const int blocks = 256, threads = 256 * wps;
        mfma_bf16_loop<NACC><<<blocks, threads>>>(out, 10, 1.f);
        hipEventRecord(e0);
        mfma_bf16_loop<NACC><<<blocks, threads>>>(out, iters, 1.f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double n_mfma = 32.0 * iters * (threads / 64) * blocks;
        const double flops = 2.0 * 32 * 32 * 16 * n_mfma;
        printf("mfma_f32_32x32x16_bf16, %d accumulator tile(s) per wave, %d wave/SIMD: %.0f TFLOP/s (%.1f ns per MFMA per SIMD; 32 cycles at "
               "2.4 GHz = 13.3 ns)\n", NACC, wps, flops / (ms * 1e-3) / 1e12, ms * 1e6 / (32.0 * iters * wps));
    }
}

__global__ __launch_bounds__(256) void read_stream(const f32x4* __restrict__ in, size_t n, float* out) {
    f32x4 s = {0, 0, 0, 0};
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i + 3 * stride < n; i += 4 * stride) {
        f32x4 a = in[i], b = in[i + stride], c = in[i + 2 * stride], d = in[i + 3 * stride];
        s += (a + b) + (c + d);
    }
    for (; i < n; i += stride) s += in[i];
    if (s.x + s.y + s.z + s.w == 12345.678f) out[0] = 1.f;
}

int main() {
    float* out; hipMalloc(&out, 1 << 22);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wps = 1; wps <= 2; ++wps) {
        const int iters = 4000, blocks = 256, threads = 256 * wps;
        mfma_loop<<<blocks, threads>>>(out, 10, 1.f, 2.f);
        hipEventRecord(e0);
        mfma_loop<<<blocks, threads>>>(out, iters, 1.f, 2.f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double flops = 2.0 * 32 * 32 * 2 * 32.0 * iters * (threads / 64) * blocks;
        double cyc_per_mfma = ms * 1e-3 * 2.4e9 / (32.0 * iters * wps);
        printf("mfma_f32_32x32x2 %d wave/SIMD: %.1f TFLOP/s (%.1f cycles@2.4GHz per MFMA per SIMD => eff. clock %.2f GHz if 64 cyc)\n",
               wps, flops / (ms * 1e-3) / 1e12, cyc_per_mfma, 2.4 * 64.0 / cyc_per_mfma);
    }
    run_bf16<1>(out, e0, e1);
    run_bf16<4>(out, e0, e1);
    printf("sustained (about 50 ms per launch: long enough for the power management to settle):\n");
    run_bf16<4>(out, e0, e1, 110000);
    const size_t bytes = 1ull << 30;
    f32x4* buf; hipMalloc(&buf, bytes); hipMemset(buf, 1, bytes);
    for (int blocks : {2048, 4096, 8192}) {
        read_stream<<<blocks, 256>>>(buf, bytes / 16, out);
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) read_stream<<<blocks, 256>>>(buf, bytes / 16, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("read-only stream, %d blocks: %.2f TB/s\n", blocks, 5.0 * bytes / (ms * 1e-3) / 1e12);
    }
    // the same stream over buffers that fit the 256 MiB Infinity Cache (warm: the previous pass left them there)
    for (size_t mb : {32, 64, 128, 152, 200}) {
        const size_t nb = mb << 20;
        read_stream<<<4096, 256>>>(buf, nb / 16, out);
        read_stream<<<4096, 256>>>(buf, nb / 16, out);
        hipEventRecord(e0);
        for (int r = 0; r < 10; ++r) read_stream<<<4096, 256>>>(buf, nb / 16, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("read-only stream, %zu MB re-read (Infinity-Cache resident): %.2f TB/s, %.1f us per pass\n", mb,
               10.0 * nb / (ms * 1e-3) / 1e12, ms * 100.0);
    }
    return 0;
}

// Issue rates of the vector-ALU instruction kinds the several-queries score pass is made of (csrc/attn_scores.h), measured on the
// GPU box: cycles per wave64 instruction per SIMD for v_rcp_f32, v_exp_f32, v_pk_fma_f32, v_fma_f32 and the factored pass's own
// mix (2 v_pk_fma + 4 v_rcp + 2 v_pk_fma per 4 element-queries), with 1 / 2 / 4 waves per SIMD and 8 independent chains per wave.
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x2 = __attribute__((ext_vector_type(2))) float;

template <int KIND>
__global__ __launch_bounds__(1024) void loop(float* out, int iters, float seed) {
    float x[8];
    f32x2 y[8];
    for (int e = 0; e < 8; ++e) { x[e] = seed + e + threadIdx.x * 1e-3f; y[e] = f32x2{x[e], x[e] + 0.5f}; }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (KIND == 0) x[e] = __builtin_amdgcn_rcpf(x[e]);
                else if (KIND == 1) x[e] = __builtin_amdgcn_exp2f(x[e]);
                else if (KIND == 2) y[e] = __builtin_elementwise_fma(y[e], f32x2{1.0001f, 0.9999f}, f32x2{1e-3f, 1e-3f});
                else if (KIND == 3) x[e] = __builtin_fmaf(x[e], 1.0001f, 1e-3f);
                else {      // the factored pass's mix on two elements: pk_fma, rcp, rcp, pk_fma
                    const f32x2 d = __builtin_elementwise_fma(y[e], f32x2{1.0001f, 0.9999f}, f32x2{1.0f, 1.0f});
                    const f32x2 r = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
                    y[e] = __builtin_elementwise_fma(f32x2{-2.0f, -2.0f}, r, y[e]);
                }
            }
    }
    float s = 0.f;
    for (int e = 0; e < 8; ++e) s += x[e] + y[e].x + y[e].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
void run(const char* what, float* out, int instr_per_slot) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int wps = 1; wps <= 4; wps *= 2) {
        const int blocks = 256, threads = 256 * wps;        // one workgroup per CU, wps waves per SIMD
        loop<KIND><<<blocks, threads>>>(out, 10, 1.5f);
        hipEventRecord(e0);
        loop<KIND><<<blocks, threads>>>(out, iters, 1.5f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double per_simd = 32.0 * iters * wps * instr_per_slot;       // wave instructions issued per SIMD
        printf("%-34s %d wave(s)/SIMD: %.2f ns per wave instruction per SIMD = %.1f cycles at 2.4 GHz\n", what, wps, ms * 1e6 / per_simd,
               ms * 1e6 / per_simd * 2.4);
    }
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 1024 * sizeof(float));
    run<0>("v_rcp_f32", out, 1);
    run<1>("v_exp_f32", out, 1);
    run<2>("v_pk_fma_f32", out, 1);
    run<3>("v_fma_f32", out, 1);
    run<4>("mix: pk_fma, 2 rcp, pk_fma (4 instr)", out, 4);
    return 0;
}

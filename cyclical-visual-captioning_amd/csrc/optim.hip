// Optimizer step of the training path (trainer.py:116-122: clip_grad_norm_ -> optimizer.step(), Adam as main.py:171-191 builds it):
// global gradient norm, clip coefficient, Adam update and the clipped gradient's write-back in THREE launches over a table of
// parameter segments -- one pass over (p, g, m, v) instead of the library's norm reductions per bucket + a multiply over every
// gradient + the fused Adam's own pass.  Everything the update needs (norm, coefficient, step counters) stays on the device, so
// the step can be captured into a HIP graph.
//
// Arithmetic = torch.optim.Adam (amsgrad = False, maximize = False), per segment:
//     g   = grad * coef                         coef = min(1, max_norm / (inv * ||grad|| + 1e-6)) * inv   (inv = 1 / ranks: the
//                                               gradient arenas hold sums over ranks, cvc.distributed.GradReducer.clip_)
//     g  += weight_decay * p
//     m   = m + (g - m) * (1 - beta1)           v = beta2 * v + (1 - beta2) * g * g
//     p  -= (lr / (1 - beta1^t)) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps)          t = the segment's step count after this step
// Sums of squares are taken chunk by chunk and combined in chunk order: the norm is the same bits run to run.
#include "cvc_common.h"
#include <math.h>

namespace {

constexpr int OPT_WG = 256;
constexpr int OPT_CHUNK = 1 << 16;          // elements per workgroup

struct OptSeg {                             // mirrors cvc_optim_seg (include/cvc_hip.h)
    float* p; float* g; float* m; float* v;
    float* step;                            // this parameter's step count (float32, as torch keeps it on the device)
    long long n;
    float lr, weight_decay;
};

struct OptChunk { int seg; int pad; long long start; };

__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < OPT_WG / 64; ++i) t += red[i];
    }
    return t;                                // valid in thread 0
}

__global__ __launch_bounds__(OPT_WG) void optim_sumsq_kernel(const OptSeg* segs, const OptChunk* chunks, float* partial) {
    __shared__ float red[OPT_WG / 64];
    const OptChunk c = chunks[blockIdx.x];
    const OptSeg s = segs[c.seg];
    const long long end = c.start + OPT_CHUNK < s.n ? c.start + OPT_CHUNK : s.n;
    float acc = 0.f;
    const float* g = s.g;
    if ((((uintptr_t)g) & 15) == 0) {
        long long i = c.start + (long long)threadIdx.x * 4;
        for (; i + 4 <= end; i += OPT_WG * 4) {
            const f32x4 x = ld4(g + i);
            acc += (x.x * x.x + x.y * x.y) + (x.z * x.z + x.w * x.w);
        }
        for (; i < end; ++i) acc += (i < end) ? g[i] * g[i] : 0.f;          // (the tail of the last chunk: < 4 elements of one thread)
    } else {
        for (long long i = c.start + threadIdx.x; i < end; i += OPT_WG) acc += g[i] * g[i];
    }
    const float t = block_sum(acc, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

// one workgroup: partial sums in chunk order -> norm, clip coefficient; every segment's step count += 1
__global__ __launch_bounds__(OPT_WG) void optim_finalize_kernel(const OptSeg* segs, int nseg, const float* partial, int nchunk, float max_norm,
                                                                float inv, float* out /* [0] = norm of the averaged gradient, [1] = coef */,
                                                                const int* skip) {
    __shared__ float red[OPT_WG / 64];
    if (skip && *skip != 0) {                  // the step is void (a launch of it reported invalid outputs): no step count moves
        if (threadIdx.x == 0) { out[0] = 0.f; out[1] = 0.f; }
        return;
    }
    // fixed assignment (thread t sums partials t, t + 256, ...) and a fixed combine: deterministic
    float acc = 0.f;
    for (int i = threadIdx.x; i < nchunk; i += OPT_WG) acc += partial[i];
    const float t = block_sum(acc, red);
    if (threadIdx.x == 0) {
        const float total = sqrtf(t) * inv;
        out[0] = total;
        // a NaN norm poisons every gradient, as torch's clip_grad_norm_ does (clamp(NaN, max=1) = NaN; fminf would return 1)
        out[1] = max_norm > 0.f ? (total != total ? total : fminf(max_norm / (total + 1e-6f), 1.0f) * inv) : inv;
    }
    for (int s = threadIdx.x; s < nseg; s += OPT_WG) segs[s].step[0] += 1.0f;
}

__global__ __launch_bounds__(OPT_WG) void optim_adam_kernel(const OptSeg* segs, const OptChunk* chunks, const float* norm_coef, float beta1,
                                                            float beta2, float eps, int write_grad, const int* skip) {
    const OptChunk c = chunks[blockIdx.x];
    const OptSeg s = segs[c.seg];
    const long long end = c.start + OPT_CHUNK < s.n ? c.start + OPT_CHUNK : s.n;
    if (skip && *skip != 0) {
        // void step: p, m, v stay as they are.  write_grad == 2 promised the caller zeroed gradients (the next step's zero_grad()
        // is folded into this pass): kept -- the void step's gradients are garbage anyway
        if (write_grad == 2)
            for (long long i = c.start + threadIdx.x; i < end; i += OPT_WG) s.g[i] = 0.f;
        return;
    }
    const float coef = norm_coef[1];
    const float t = s.step[0];                                             // already incremented by the finalize launch
    // bias corrections as torch computes them (1 - beta^t in fp32 via pow)
    const float bc1 = 1.0f - powf(beta1, t), bc2 = 1.0f - powf(beta2, t);
    const float step_size = s.lr / bc1, inv_sqrt_bc2 = 1.0f / sqrtf(bc2);
    const float wd = s.weight_decay, omb1 = 1.0f - beta1, omb2 = 1.0f - beta2;
    auto upd = [&](float& p, float& g, float& m, float& v) __attribute__((always_inline)) {
        g *= coef;
        const float ge = wd != 0.f ? g + wd * p : g;
        m = m + (ge - m) * omb1;
        v = beta2 * v + omb2 * ge * ge;
        p -= step_size * (m / (sqrtf(v) * inv_sqrt_bc2 + eps));
    };
    const bool al = ((((uintptr_t)s.p) | ((uintptr_t)s.g) | ((uintptr_t)s.m) | ((uintptr_t)s.v)) & 15) == 0;
    if (al) {
        long long i = c.start + (long long)threadIdx.x * 4;
        for (; i + 4 <= end; i += OPT_WG * 4) {
            f32x4 p = ld4(s.p + i), g = ld4(s.g + i), m = ld4(s.m + i), v = ld4(s.v + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float pe = p[e], ge = g[e], me = m[e], ve = v[e];
                upd(pe, ge, me, ve);
                p[e] = pe; g[e] = ge; m[e] = me; v[e] = ve;
            }
            st4(s.p + i, p); st4(s.m + i, m); st4(s.v + i, v);
            if (write_grad == 1) st4(s.g + i, g);
            else if (write_grad == 2) st4(s.g + i, f32x4{0.f, 0.f, 0.f, 0.f});
        }
        for (; i < end; ++i) {
            float p = s.p[i], g = s.g[i], m = s.m[i], v = s.v[i];
            upd(p, g, m, v);
            s.p[i] = p; s.m[i] = m; s.v[i] = v;
            if (write_grad) s.g[i] = write_grad == 2 ? 0.f : g;
        }
    } else {
        for (long long i = c.start + threadIdx.x; i < end; i += OPT_WG) {
            float p = s.p[i], g = s.g[i], m = s.m[i], v = s.v[i];
            upd(p, g, m, v);
            s.p[i] = p; s.m[i] = m; s.v[i] = v;
            if (write_grad) s.g[i] = write_grad == 2 ? 0.f : g;
        }
    }
}

}  // namespace

extern "C" int cvc_optim_chunk_elems(void) { return OPT_CHUNK; }

extern "C" int cvc_adam_clip_step(const cvc_optim_seg* segs, int nseg, const cvc_optim_chunk* chunks, int nchunk, float max_norm,
                                  float inv_world, float beta1, float beta2, float eps, int write_grad, float* partial,
                                  float* norm_coef, const int* skip, cvc_stream_t stream) {
    static_assert(sizeof(cvc_optim_seg) == sizeof(OptSeg) && sizeof(cvc_optim_chunk) == sizeof(OptChunk), "table layouts");
    if (!segs || !chunks || !partial || !norm_coef || nseg < 1 || nchunk < 1 || inv_world <= 0.f || beta1 < 0.f || beta1 >= 1.f ||
        beta2 < 0.f || beta2 >= 1.f || eps < 0.f)
        return CVC_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    const OptSeg* s = reinterpret_cast<const OptSeg*>(segs);
    const OptChunk* c = reinterpret_cast<const OptChunk*>(chunks);
    hipLaunchKernelGGL(optim_sumsq_kernel, dim3(nchunk), dim3(OPT_WG), 0, st, s, c, partial);
    hipLaunchKernelGGL(optim_finalize_kernel, dim3(1), dim3(OPT_WG), 0, st, s, nseg, partial, nchunk, max_norm, inv_world, norm_coef, skip);
    hipLaunchKernelGGL(optim_adam_kernel, dim3(nchunk), dim3(OPT_WG), 0, st, s, c, norm_coef, beta1, beta2, eps, write_grad, skip);
    return cvc_launch_status();
}

// Counter-based dropout masks (nn.Dropout of the training pass: model/captioner.py:53-68, model/decoder_core.py:62, 109,
// model/backbone.py:55-57), generated INSIDE the kernels that consume them -- no mask tensors, no torch.bernoulli, no ATen dropout
// launches -- and reproducible on the host (cvc/synth.py::dropout_keep restates this file in numpy; the train-mode parity tests
// hand those masks to the CPU oracle).
//
//   keep(element) = hash(seed, step, site, element index) >= p * 2^32          value = keep ? 1 / (1 - p) : 0
//
// state: 4 words of DEVICE memory {seed_lo, seed_hi, step, 0}; the host advances `step` once per training step with a device-side
// add (so a HIP-graph replay of the step draws fresh masks).  site: which dropout of the pass (cvc/dropout.py numbers them).
#pragma once
#include <stdint.h>

struct DropSpec {              // by-value kernel argument
    const uint32_t* state;     // null = no dropout
    uint32_t site;
    uint32_t thresh;           // p * 2^32: an element is DROPPED when its hash is below
    float scale;               // 1 / (1 - p)
};

__host__ __device__ inline uint32_t cvc_drop_hash(uint32_t seed_lo, uint32_t seed_hi, uint32_t step, uint32_t site, uint32_t idx) {
    uint32_t x = idx * 0x9E3779B1u + site * 0x85EBCA77u + step * 0xC2B2AE3Du + seed_lo;
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    x += seed_hi;
    x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12; x *= 0x297A2D39u; x ^= x >> 15;
    return x;
}

#ifdef __HIPCC__
// the multiplier of element idx (0 or scale); s0..s2 = state words, read once per thread by the caller
__device__ __forceinline__ float cvc_drop_mult(const DropSpec& d, uint32_t s0, uint32_t s1, uint32_t s2, uint32_t idx) {
    return cvc_drop_hash(s0, s1, s2, d.site, idx) >= d.thresh ? d.scale : 0.f;
}
#endif

static inline DropSpec cvc_drop_spec(const uint32_t* state, uint32_t site, float p) {
    DropSpec d;
    d.state = (state != nullptr && p > 0.f) ? state : nullptr;
    d.site = site;
    const double t = (double)p * 4294967296.0;
    d.thresh = t >= 4294967295.0 ? 0xffffffffu : (uint32_t)t;
    d.scale = p < 1.f ? 1.f / (1.f - p) : 0.f;
    return d;
}

// Segment arithmetic of the grouped stream-K GEMM (gemm_gsk.hip), shared with the kernels that sum its partial tiles:
// the late LSTM gate kernel (gemm_packed.hip), the score pass's query load (attn_scores.h) and the word selection (vocab.hip).
//
// A launch walks a linear space of UNITS, one unit = (256-row tile, one 32-k chunk): the tiles of group 0 first (tile-major,
// chunk-minor), then group 1, ...  Workgroup w owns units [w U, (w + 1) U).  Its intersection with a tile is one SEGMENT of that
// tile: a partial product over a contiguous chunk range, stored as a [8 blocks][64 rows][32 gate rows] fp32 slab at
//   slab + ((tile * maxseg + seg) * 8 + block_in_tile) * 2048,     seg = w - first_wg(tile).
// A tile's segments are summed by its consumer in segment order, so results do not depend on which workgroup ran when.
#pragma once
#include "cvc_common.h"

using GskSegs = cvc_gsk_segs;   // { slab, unit0 (first unit of the group), nchunk (units per tile), U (units per workgroup), maxseg }

__host__ __device__ inline int gsk_first_wg(int unit0, int nchunk, int U, int tile) { return (unit0 + tile * nchunk) / U; }
__host__ __device__ inline int gsk_nseg(int unit0, int nchunk, int U, int tile) {
    return (unit0 + (tile + 1) * nchunk - 1) / U - (unit0 + tile * nchunk) / U + 1;
}
__device__ __forceinline__ int gsk_nseg(const GskSegs& g, int tile) { return gsk_nseg(g.unit0, g.nchunk, g.U, tile); }
// partial tile of (tile, seg, block j of the tile): [64 rows][32 gate rows]
__device__ __forceinline__ const float* gsk_part(const GskSegs& g, int tile, int seg, int j) {
    return g.slab + ((size_t)(tile * g.maxseg + seg) * 8 + j) * 2048;
}

// hashgrid_common.h -- device helpers shared by the stand-alone encoder kernels and the
// fused render kernels: hash, cell location, trilinear weights, table element loads.
//
// Reference behaviour: hashgrid/src/hashgrid_bg_kernel.cu:14-24 (hash), :27-38 (weights),
// :40-77 (weight derivatives), :124-130 (cell location, contracted space),
// hashgrid/src/hashgrid_kernel.cu:127-143 (cell location, world-space box).
#pragma once
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>

#include "common.h"

namespace scanerf {

__device__ __forceinline__ uint32_t grid_hash(int x, int y, int z, uint32_t mask)
{
    return ((uint32_t)x ^ ((uint32_t)y * 2654435761u) ^ ((uint32_t)z * 805459861u)) & mask;
}

// corner c = (dx<<2)|(dy<<1)|dz  (z fastest), the reference's 000,001,...,111 order
__device__ __forceinline__ void corner_indices(uint32_t idx[8], int bx, int by, int bz, uint32_t mask)
{
    uint32_t hx0 = (uint32_t)bx, hx1 = (uint32_t)(bx + 1);
    uint32_t hy0 = (uint32_t)by * 2654435761u, hy1 = (uint32_t)(by + 1) * 2654435761u;
    uint32_t hz0 = (uint32_t)bz * 805459861u, hz1 = (uint32_t)(bz + 1) * 805459861u;
    idx[0] = (hx0 ^ hy0 ^ hz0) & mask;
    idx[1] = (hx0 ^ hy0 ^ hz1) & mask;
    idx[2] = (hx0 ^ hy1 ^ hz0) & mask;
    idx[3] = (hx0 ^ hy1 ^ hz1) & mask;
    idx[4] = (hx1 ^ hy0 ^ hz0) & mask;
    idx[5] = (hx1 ^ hy0 ^ hz1) & mask;
    idx[6] = (hx1 ^ hy1 ^ hz0) & mask;
    idx[7] = (hx1 ^ hy1 ^ hz1) & mask;
}

__device__ __forceinline__ void trilinear_weights(float w[8], float tx, float ty, float tz)
{
    float ax = 1 - tx, ay = 1 - ty, az = 1 - tz;
    w[0] = ax * ay * az;
    w[1] = ax * ay * tz;
    w[2] = ax * ty * az;
    w[3] = ax * ty * tz;
    w[4] = tx * ay * az;
    w[5] = tx * ay * tz;
    w[6] = tx * ty * az;
    w[7] = tx * ty * tz;
}

// contracted-space ("bg") cell: p in [-2,2] -> [0,1] -> v = p01*(res-1)
// (no FMA contraction here: t = v - b must subtract the ROUNDED product, as separate mul / sub
// do; a fused p01*(res-1)-b changes the offset by up to half an ulp of v, ~1e-4 at fine levels)
__device__ __forceinline__ void locate_bg(float p, int res, int &b, float &t, float &scale)
{
#pragma clang fp contract(off)
    float p01 = (p + 2.0f) / 4.0f;
    float v = p01 * (float)(res - 1);
    b = (int)v;
    t = v - (float)b;
    scale = (float)(res - 1) / 4.0f;
}

// world-space box cell: clamp to the box, grid = size/(res-1)
__device__ __forceinline__ void locate_box(float p_in, int res, float corner, float size, int &b, float &t,
                                           float &scale)
{
#pragma clang fp contract(off)
    float p = fmaxf(corner, fminf(p_in, corner + size));
    float g = size / (float)(res - 1);
    b = (int)((p - corner) / g);
    float vmin = (float)b * g + corner;
    t = (p - vmin) / g;
    scale = 1.0f / g;
}

// ---- table element loads: one (f0,f1) entry --------------------------------------
template <int DT>
struct TableElem;
template <>
struct TableElem<SCANERF_F32> {
    static __device__ __forceinline__ float2 load(const void *base, uint32_t i)
    {
        return reinterpret_cast<const float2 *>(base)[i];
    }
    static constexpr int bytes = 8;
};
template <>
struct TableElem<SCANERF_F16> {
    static __device__ __forceinline__ float2 load(const void *base, uint32_t i)
    {
        __half2 h = reinterpret_cast<const __half2 *>(base)[i];
        return __half22float2(h);
    }
    static constexpr int bytes = 4;
};
template <>
struct TableElem<SCANERF_BF16> {
    static __device__ __forceinline__ float2 load(const void *base, uint32_t i)
    {
        uint32_t u = reinterpret_cast<const uint32_t *>(base)[i];
        return make_float2(__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u));
    }
    static constexpr int bytes = 4;
};

// ---- the 8 corner entries of one cell with fewer cache-line requests (half-precision tables) ------------------------
// Scattered loads on gfx950 cost one request per lane and line whatever their width (tools/gather_bench.hip: 265 G/s from
// L2, 55-80 G/s beyond it, the same for 8-byte and 16-byte loads).  idx(x+1) = idx(x) ^ ((x ^ (x+1)) & mask): for even x
// the two x-neighbours of a (y,z) corner are the entries i and i^1 -- one aligned two-entry chunk, one load; only for odd
// x does the neighbour live elsewhere: 6 requests per cell on average instead of 8.  Two exclusive paths write the same
// 8 results.  Half-precision tables: 1.24 -> 1.03 ms (bf16, configs[2]).  fp32 tables (16-byte chunk loads): round 2 measured
// 3.2 -> 4.3 ms in the then register-bound forward (spills); round 3's forward has registers to spare (no packed arithmetic,
// opaque LDS addressing: 0 spills) and takes them at 3.10 -> 3.03 ms (-DSCANERF_PAIRED_F32=1, csrc/Makefile) -- little,
// because the x-neighbours share their 64-byte line anyway: the kernel is bound by L2 MISSES (lines), not by requests.
// f[c] in corner_indices order (c = dx<<2 | dy<<1 | dz).
template <int DT>
__device__ __forceinline__ void gather_cell(const void *slice, const uint32_t idx[8], bool x_odd, float2 f[8])
{
    if constexpr (DT == SCANERF_F32) {
        // fp32 entries: the aligned pair (i & ~1, i | 1) is one 16-byte load.  Branch-free form: every lane loads the pair of
        // its x0 corner; lanes with odd x0 (the x1 corner lives elsewhere) add one 8-byte load, issued under EXEC so that only
        // those lanes cost a cache-line lookup: 6 lookups per cell on average instead of 8.
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 t = reinterpret_cast<const float4 *>(slice)[idx[q] >> 1];
            const bool hi = idx[q] & 1u;
            f[q] = hi ? make_float2(t.z, t.w) : make_float2(t.x, t.y);
            f[4 + q] = hi ? make_float2(t.x, t.y) : make_float2(t.z, t.w);
        }
        if (x_odd) {
#pragma unroll
            for (int q = 0; q < 4; ++q) f[4 + q] = reinterpret_cast<const float2 *>(slice)[idx[4 + q]];
        }
        return;
    }
    auto unpack = [](uint32_t u) {
        if (DT == SCANERF_F16) return __half22float2(*reinterpret_cast<const __half2 *>(&u));
        return make_float2(__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u));
    };
    if (x_odd) {
#pragma unroll
        for (int c = 0; c < 8; ++c) f[c] = TableElem<DT>::load(slice, idx[c]);
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint2 t = reinterpret_cast<const uint2 *>(slice)[idx[q] >> 1];
            const bool hi = idx[q] & 1u;
            f[q] = unpack(hi ? t.y : t.x);
            f[4 + q] = unpack(hi ? t.x : t.y);
        }
    }
}

}  // namespace scanerf

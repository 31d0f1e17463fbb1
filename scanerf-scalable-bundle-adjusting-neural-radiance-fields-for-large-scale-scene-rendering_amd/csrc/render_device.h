// render_device.h -- device code shared by the fused render kernels (forward: render.hip,
// backward: render_bwd.hip): decoder helpers on fp32 MFMA, hash encode of a tile, contraction.
// See render_common.h for the register / LDS layouts.
#pragma once
#include "hashgrid_common.h"
#include "render_common.h"

typedef float v16f __attribute__((ext_vector_type(16)));

#ifndef SCANERF_PAIRED_F32
#define SCANERF_PAIRED_F32 0   // 1: paired (16-byte) gathers for fp32 tables too (experiment; see gather_cell)
#endif

namespace scanerf {

// ------------------------------------------------------------------ device helpers
__device__ __forceinline__ float gauss_act(float x) { return __expf(x * x * -50.0f); }  // exp(-x^2/(2*0.1^2))
__device__ __forceinline__ float sigmoid_(float x) { return 1.0f / (1.0f + __expf(-x)); }
// the same two functions at hardware-instruction cost for the split-f16 kernels, whose time is VALU issue:
// exp2 with the constants folded (3 instructions instead of 4) and v_rcp_f32 (1 ulp) instead of an IEEE division
__device__ __forceinline__ float gauss_fast(float x) { return __builtin_amdgcn_exp2f(x * x * -72.13475204444817f); }
__device__ __forceinline__ float sigmoid_fast(float x)
{
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
}
__device__ __forceinline__ float softplus_(float x) { return x > 20.0f ? x : log1pf(__expf(x)); }
// softplus at hardware-instruction cost: max(x, 0) + log1p(exp(-|x|)) on v_exp_f32 / v_log_f32, with the series e - e^2 / 2 + e^3 / 3
// where 1 + e would lose e's digits (e < 2^-6: truncation 1e-6 relative to that term; above, 1 + e keeps e to 2^-18).  The library's
// log1pf is ~40 vector instructions per call.
__device__ __forceinline__ float softplus_fast(float x)
{
    const float e = __builtin_amdgcn_exp2f(-fabsf(x) * 1.4426950408889634f);
    const float l = e < 0x1p-6f ? e * (1.0f - e * (0.5f - e * 0.33333334f)) : __builtin_amdgcn_logf(1.0f + e) * 0.6931471805599453f;
    return fmaxf(x, 0.0f) + l;
}

__device__ __forceinline__ v16f load_bias(const float *lds, int layer, int blk, int h)
{
    const float4 *p = reinterpret_cast<const float4 *>(lds + PK_BIAS + ((layer * 2 + blk) * 2 + h) * 16);
    float4 a = p[0], b = p[1], c = p[2], d = p[3];
    v16f r = { a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w };
    return r;
}

// acc += W_img[blk] (32 x 2*NG*4... ) * B, with B supplied 4 steps at a time
#define MFMA4(acc, a4, b0, b1, b2, b3)                                          \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32((a4).x, (b0), acc, 0, 0, 0);     \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32((a4).y, (b1), acc, 0, 0, 0);     \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32((a4).z, (b2), acc, 0, 0, 0);     \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32((a4).w, (b3), acc, 0, 0, 0);

// one 32-row output block over a 16-register (32-input) B tile: 4 groups of 4 steps
__device__ __forceinline__ void mma_block16(v16f &acc, const float *img, int grp0, int lane, const v16f &b)
{
    const float *A = img + (lane + (lane >> 5)) * 4;
    const float4 a0 = *reinterpret_cast<const float4 *>(A + (grp0 + 0) * PK_GRP),
                 a1 = *reinterpret_cast<const float4 *>(A + (grp0 + 1) * PK_GRP),
                 a2 = *reinterpret_cast<const float4 *>(A + (grp0 + 2) * PK_GRP),
                 a3 = *reinterpret_cast<const float4 *>(A + (grp0 + 3) * PK_GRP);
    MFMA4(acc, a0, b[0], b[1], b[2], b[3])
    MFMA4(acc, a1, b[4], b[5], b[6], b[7])
    MFMA4(acc, a2, b[8], b[9], b[10], b[11])
    MFMA4(acc, a3, b[12], b[13], b[14], b[15])
    __builtin_amdgcn_sched_barrier(0);  // keep the scheduler from hoisting every layer's LDS reads
}

// Backward chain through the SAME image: acc[i][s] += sum_n W[n][i] dY[n][s] for the 16 output
// units n held in dy (output block nb), i = this lane's input unit of input block ib.
// One conflict-free ds_read_b32 per MFMA (see PK_GRP in render_common.h).
__device__ __forceinline__ void chain16(v16f &acc, const float *img, int ngrp, int lane, int ib, int nb, const v16f &dy)
{
    const int i5 = lane & 31, hp = lane >> 5;
    const int ri = 16 * ib + nmap_g(i5), hi = nmap_h(i5);
    const float *A = img + (nb * ngrp + (ri >> 2)) * PK_GRP + (4 * hp + 33 * hi) * 4 + (ri & 3);
#pragma unroll
    for (int g = 0; g < 16; ++g) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[nmap(g, 0) * 4], dy[g], acc, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
}

__device__ __forceinline__ v16f act16(const v16f &x)
{
    v16f r;
#pragma unroll
    for (int g = 0; g < 16; ++g) r[g] = gauss_act(x[g]);
    return r;
}

__device__ __forceinline__ float half_sum(float v)  // sum over the 32 lanes of a half
{
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off, 32);
    return v;
}

struct RayConst {
    float o[3], d[3], dnorm;
    float sh[16];
};

// SH deg-3 of the normalised direction (network.py:38-77, viewdirs/(norm+1e-8): :177)
__device__ __forceinline__ void ray_sh(const float d[3], float dnorm, float sh[16], float eps = 1e-8f)
{
    // training decoder: d/(|d|+1e-8) (network.py:177); render-time decoder: normalize(d), eps = 0 (decoder.h:201)
    const float inv = 1.0f / (dnorm + eps);
    const float x = d[0] * inv, y = d[1] * inv, z = d[2] * inv;
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    sh[0] = 0.28209479177387814f;
    sh[1] = 0.4886025119029199f * y;
    sh[2] = 0.4886025119029199f * z;
    sh[3] = 0.4886025119029199f * x;
    sh[4] = 1.0925484305920792f * xy;
    sh[5] = -1.0925484305920792f * yz;
    sh[6] = 0.31539156525252005f * (2.0f * zz - xx - yy);
    sh[7] = -1.0925484305920792f * xz;
    sh[8] = 0.5462742152960396f * (xx - yy);
    sh[9] = -0.5900435899266435f * y * (3 * xx - yy);
    sh[10] = 2.890611442640554f * xy * z;
    sh[11] = -0.4570457994644658f * y * (4 * zz - xx - yy);
    sh[12] = 0.3731763325901154f * z * (2 * zz - 3 * xx - 3 * yy);
    sh[13] = -0.4570457994644658f * x * (4 * zz - xx - yy);
    sh[14] = 1.445305721320277f * z * (xx - yy);
    sh[15] = -0.5900435899266435f * x * (xx - 3 * yy);
}

struct RenderArgs {
    const float *rays_o, *rays_d, *z_vals, *dists;
    const void *features;
    const int32_t *resolutions;
    const float *packed;      // PK_TOTAL floats
    const uint8_t *ray_valid; // optional
    float *out_ray, *weights;
    float *tile_T;            // optional [B, ceil(S/16)]: transmittance entering each 16-sample tile (for backward)
    float *xstash;            // optional [B*S][2][16]: encoder outputs per (sample, half-wave) (for backward)
    uint32_t *jstash;         // optional [B][ceil(S/32)][8][4][64] words (jst_pack): d(encoder outputs)/d(contracted position) per (ray, 32-sample
                              // tile, level j = 0..7 of the half-wave, component pair, forward lane = 32 h + (s & 31)); the six
                              // components = (feature 0: d/dx, d/dy, d/dz; feature 1: ...).  Lane-fastest so that every store of
                              // the wave is 256 contiguous bytes.  Half precision (the t16 backward that reads it multiplies f16
                              // gradient operands anyway): 1.6 instead of 3.2 GB written and read per 65 536 x 128 samples.  Lets the t16 backward form the pose gradient without gathering
                              // the table again
    int B, S, T;
    int contract_mode, infinity;
    float min_bbox[3], inv_size4[3];  // 4/bbox_size
    float bbox_size[3];
    int dbg;                  // timing experiments only (SCANERF_DEBUG_FWD): 1 = encode only, 2 = decode only
    uint32_t skip_levels;     // bit l: level l AND its half-wave partner l ^ 2 meet exactly-zero weights (coarse-to-fine mask,
                              // pair_masked_levels): no gathers, no scatter records for them
    // optional: the forward kernel also counts the scatter records the t16 backward will emit for these rays (scatter.hip
    // k_bin_count_rays' job: the hash indices are already in registers here).  counts [16 * NB][W], this workgroup's column
    uint32_t *plan_counts, *plan_maxbits, *plan_overflow;   // (launch maximum and overflow flag are zeroed here; plan_overflow[-1] = format)
    int plan_NB, plan_bucket_log, plan_W, plan_rec8;
};

// two f32 -> one word of two IEEE halves (round to nearest) and back
__device__ __forceinline__ uint32_t pack_f16x2(float a, float b)
{
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const h2 v = { (_Float16)a, (_Float16)b };
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float2 unpack_f16x2(uint32_t w)
{
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const h2 v = __builtin_bit_cast(h2, w);
    return make_float2((float)v[0], (float)v[1]);
}

// ---- Jacobian stash entry (RenderArgs::jstash): the six values d(feature f)/d(p_k) of one (sample, level) in 16 bytes --
// six 20-bit two's-complement significands under ONE 8-bit exponent (that of the largest of the six): every value to 2^-20 of
// the largest.  Round 3 stored them as three f16 pairs (12 bytes, 2^-11 each): the position path of the pose gradients was then
// 2e-4 of the largest ray gradient off, 40x everything else in the t16s backward; an f32 stash would double the 1.6 GB.
//   w0 = q0 | q1[11:0] << 20;  w1 = q1[19:12] | q2 << 8 | q3[3:0] << 28;  w2 = q3[19:4] | q4[15:0] << 16;
//   w3 = q4[19:16] | q5 << 4 | (E + 128) << 24;      v_i = q_i * 2^(E - 19),  max |v_i| < 2^E
__device__ __forceinline__ uint4 jst_pack(const float v[6])
{
    const float m = fmaxf(fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))), fmaxf(fabsf(v[4]), fabsf(v[5])));
    int E = __builtin_amdgcn_frexp_expf(m);           // m < 2^E (0 for m = 0)
    E = E < -100 ? -100 : (E > 100 ? 100 : E);          // (Jacobians of a table of finite features: far inside)
    uint32_t q[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        int qi = __float2int_rn(__builtin_amdgcn_ldexpf(v[i], 19 - E));
        qi = qi > 524287 ? 524287 : (qi < -524287 ? -524287 : qi);
        q[i] = (uint32_t)qi & 0xfffffu;
    }
    return make_uint4(q[0] | (q[1] << 20), (q[1] >> 12) | (q[2] << 8) | (q[3] << 28), (q[3] >> 4) | (q[4] << 16),
                      (q[4] >> 16) | (q[5] << 4) | ((uint32_t)(E + 128) << 24));
}
// -> the six significands as floats and the scale 2^(E - 19) they share
__device__ __forceinline__ void jst_unpack(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, float q[6], float &scale)
{
    q[0] = (float)(int)__builtin_amdgcn_sbfe((int)w0, 0, 20);
    q[1] = (float)(int)__builtin_amdgcn_sbfe((int)__builtin_amdgcn_alignbit(w1, w0, 20), 0, 20);
    q[2] = (float)(int)__builtin_amdgcn_sbfe((int)w1, 8, 20);
    q[3] = (float)(int)__builtin_amdgcn_sbfe((int)__builtin_amdgcn_alignbit(w2, w1, 28), 0, 20);
    q[4] = (float)(int)__builtin_amdgcn_sbfe((int)__builtin_amdgcn_alignbit(w3, w2, 16), 0, 20);
    q[5] = (float)(int)__builtin_amdgcn_sbfe((int)w3, 4, 20);
    scale = __builtin_amdgcn_ldexpf(1.0f, (int)(w3 >> 24) - 128 - 19);
}

// hash-encode 8 levels of one sample: register 2j+f of half h holds feature f of level
// 4(j>>1) + 2h + (j&1), i.e. input unit nmap(2j+f, h) = 2*level + f -- the same register<->unit
// map as every other layer, and the layout in which the backward pass produces dL/dx.
// GATHER_BATCH = levels whose 8 gathers each are in flight together: 2 keeps the forward kernel at
// two waves per SIMD (128-register budget); 8 issues all 64 gathers of the lane at once -- one
// memory latency instead of four -- for the one-wave-per-SIMD backward kernel.
// PAIRED: fetch x-neighbour pairs with one load where the hash puts them side by side (gather_cell; half-precision tables)
// hist (may be null): this workgroup's [16][NB] record counters in LDS -- one per (y,z) corner pair, two when the
// x-neighbours fall into different buckets, exactly scatter_common.h count_pairs; count = this lane's sample is a real one
// jrow (may be null): this lane's column of the tile's [8][4][64] block of RenderArgs::jstash (jst_pack)
template <int DT, int GATHER_BATCH = 2, bool PAIRED = false, bool COUNT = false, bool JST = false>
__device__ __forceinline__ void encode8(const RenderArgs &a, const int *lds_res, int h, const float p[3], v16f &x,
                                        uint32_t *hist = nullptr, bool count = false, uint32_t *jrow = nullptr)
{
    const uint32_t mask = (uint32_t)a.T - 1u;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int level = 4 * (j >> 1) + 2 * h + (j & 1);
        // coarse-to-fine: skip_levels is symmetric in the two half-waves' levels of a j (render_common.h pair_masked_levels), so
        // this is wave-uniform -- the table is not touched, no records are counted; the zero weights give the same contribution
        // for x = 0 as for the features
        if ((a.skip_levels >> (4 * (j >> 1) + (j & 1))) & 1u) {
            x[2 * j] = 0.0f;
            x[2 * j + 1] = 0.0f;
            if constexpr (JST) {
                uint32_t *jr = jrow + 4 * j * 64;
                jr[0] = jr[64] = jr[128] = jr[192] = 0u;
            }
            continue;
        }
        const int4 res = reinterpret_cast<const int4 *>(lds_res)[level];
        int b[3];
        float t[3], sc[3];
        locate_bg(p[0], res.x, b[0], t[0], sc[0]);
        locate_bg(p[1], res.y, b[1], t[1], sc[1]);
        locate_bg(p[2], res.z, b[2], t[2], sc[2]);
        uint32_t idx[8];
        float w[8];
        corner_indices(idx, b[0], b[1], b[2], mask);
        trilinear_weights(w, t[0], t[1], t[2]);
        if constexpr (COUNT) {  // (no per-lane branch: dead lanes add zero)
            uint32_t *hl = hist + level * a.plan_NB;
            const uint32_t one = count ? 1u : 0u;
#pragma unroll
            for (int q = 0; q < 4; ++q) atomicAdd(&hl[idx[q] >> a.plan_bucket_log], one);
            // (x-neighbours in different buckets -- practically never -- cost a second count; added as 0 otherwise, no branch)
            const uint32_t two = ((idx[0] ^ idx[4]) >> a.plan_bucket_log) != 0u ? one : 0u;
#pragma unroll
            for (int q = 0; q < 4; ++q) atomicAdd(&hl[idx[4 + q] >> a.plan_bucket_log], two);
        }
        const char *slice = (const char *)a.features + (size_t)level * a.T * TableElem<DT>::bytes;
        float2 f[8];
        if constexpr (PAIRED && (DT != SCANERF_F32 || SCANERF_PAIRED_F32)) {
            gather_cell<DT>(slice, idx, b[0] & 1, f);
        } else {
#pragma unroll
            for (int c = 0; c < 8; ++c) f[c] = TableElem<DT>::load(slice, idx[c]);
        }
        float ax = 0.0f, ay = 0.0f;
#ifndef ENC_INTERP
#define ENC_INTERP 0
#endif
#pragma unroll
        for (int c = 0; c < 8; ++c) {
#if ENC_INTERP == 1   // every weight one opaque register: no (w, w) pairs for packed multiply-adds
            asm volatile("" : "+v"(w[c]));
#endif
            ax = fmaf(w[c], f[c].x, ax);
            ay = fmaf(w[c], f[c].y, ay);
#if ENC_INTERP == 2   // the two sums kept apart: no packed multiply-adds at all
            asm volatile("" : "+v"(ax));
            asm volatile("" : "+v"(ay));
#endif
        }
        x[2 * j] = ax;
        x[2 * j + 1] = ay;
        if constexpr (JST) {
            // d(out)/d(p) = scale * sum_c f_c * dw_c/dt (csrc/hashgrid.hip k_embed_bwd's point gradient, before the contraction
            // with the upstream gradient): corner c = (dx << 2) | (dy << 1) | dz.  Formed as differences along each axis of the
            // bilinear interpolations on the two faces (6 face values per feature instead of 12 weight products held at once).
            const float tx = t[0], ty = t[1], tz = t[2];
            uint32_t *jr = jrow + 4 * j * 64;   // [4 words][64 lanes] per level (jst_pack)
            float jv[6];
#pragma unroll
            for (int ft = 0; ft < 2; ++ft) {
                float v[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) v[c] = ft ? f[c].y : f[c].x;
                // along z first: e[dx][dy] = lerp_z, dz[dx][dy] = difference
                const float e00 = fmaf(tz, v[1] - v[0], v[0]), e01 = fmaf(tz, v[3] - v[2], v[2]), e10 = fmaf(tz, v[5] - v[4], v[4]),
                            e11 = fmaf(tz, v[7] - v[6], v[6]);
                const float d00 = v[1] - v[0], d01 = v[3] - v[2], d10 = v[5] - v[4], d11 = v[7] - v[6];
                // d/dz = bilinear(x, y) of the z-differences
                const float dz0 = fmaf(ty, d01 - d00, d00), dz1 = fmaf(ty, d11 - d10, d10);
                const float gz = fmaf(tx, dz1 - dz0, dz0);
                // d/dy = lerp_x of (e?1 - e?0); d/dx = lerp_y of (e1? - e0?)
                const float gy = fmaf(tx, (e11 - e10) - (e01 - e00), e01 - e00);
                const float gx = fmaf(ty, (e11 - e01) - (e10 - e00), e10 - e00);
                jv[3 * ft + 0] = sc[0] * gx;
                jv[3 * ft + 1] = sc[1] * gy;
                jv[3 * ft + 2] = sc[2] * gz;
            }
#if defined(JST_DBG) && JST_DBG == 1   // timing experiments only: the Jacobians formed, not stored
            {
                uint32_t k0 = pack_f16x2(jv[0], jv[1]), k1 = pack_f16x2(jv[2], jv[3]), k2 = pack_f16x2(jv[4], jv[5]);
                asm volatile("" :: "v"(k0), "v"(k1), "v"(k2));
            }
#elif defined(JST_DBG) && JST_DBG == 2   // timing experiments only: stored, not formed
            jr[0] = __float_as_uint(f[0].x);
            jr[64] = __float_as_uint(f[1].x);
            jr[128] = __float_as_uint(f[2].x);
#else
            const uint4 jw = jst_pack(jv);
            jr[0] = jw.x;
            jr[64] = jw.y;
            jr[128] = jw.z;
            jr[192] = jw.w;
#endif
        }
        if ((j + 1) % GATHER_BATCH == 0) __builtin_amdgcn_sched_barrier(0);  // bound the gathers in flight per lane
    }
}

// (box and mode as plain arguments: the stand-alone scatter of scatter.hip places the samples of a render branch with it)
__device__ __forceinline__ void contract_point_box(const float min_bbox[3], const float bbox_size[3], int contract_mode,
                                                   const float o[3], const float d[3], float z, float p[3])
{
#pragma clang fp contract(off)  // torch evaluates these as separate ops (hashgrid/__init__.py:394-411,519)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float w = o[k] + z * d[k];
        p[k] = (w - min_bbox[k]) / bbox_size[k] * 4.0f - 2.0f;
    }
    if (contract_mode == 1) {
        float linf = fmaxf(fabsf(p[0]), fmaxf(fabsf(p[1]), fabsf(p[2])));
        float ratio = (2.0f - 1.0f / linf) / linf;
        p[0] *= ratio;
        p[1] *= ratio;
        p[2] *= ratio;
    }
}
__device__ __forceinline__ void contract_point(const RenderArgs &a, const float o[3], const float d[3], float z,
                                               float p[3])
{
    contract_point_box(a.min_bbox, a.bbox_size, a.contract_mode, o, d, z, p);
}

// per-sample decoder outputs (identical in both halves of the wave)
struct SampleOut {
    float sigma, dif[3], tint[3], spec[3];
};

// Decoder on one tile: x = 32 inputs per sample (16 registers x 2 halves) -> SampleOut.
// dinit = Dir layer-0 accumulator start (bias + SH part), constant per ray.
__device__ __forceinline__ SampleOut decode_tile(const float *lds, int lane, const v16f &x, const v16f dinit[2])
{
    const int h = lane >> 5;
    // Spatial_MLP.mlp.0 (32 -> 64) + Gaussian
    v16f a0 = load_bias(lds, 0, 0, h), a1 = load_bias(lds, 0, 1, h);
    mma_block16(a0, lds + PK_L0, 0, lane, x);
    mma_block16(a1, lds + PK_L0, 4, lane, x);
    a0 = act16(a0);
    a1 = act16(a1);
    // Spatial_MLP.mlp.2 (64 -> 64), linear
    v16f h0 = load_bias(lds, 1, 0, h), h1 = load_bias(lds, 1, 1, h);
    mma_block16(h0, lds + PK_L1, 0, lane, a0);
    mma_block16(h0, lds + PK_L1, 4, lane, a1);
    mma_block16(h1, lds + PK_L1, 8, lane, a0);
    mma_block16(h1, lds + PK_L1, 12, lane, a1);
    // heads on H[:32] = h0: per-lane partial dot over its 16 units, partner half adds the rest
    float hd[7] = { 0, 0, 0, 0, 0, 0, 0 };
    {
        const float4 *W = reinterpret_cast<const float4 *>(lds + PK_HEAD + h * 128);
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            float4 wa = W[2 * g], wb = W[2 * g + 1];
            float v = h0[g];
            hd[0] = fmaf(v, wa.x, hd[0]);
            hd[1] = fmaf(v, wa.y, hd[1]);
            hd[2] = fmaf(v, wa.z, hd[2]);
            hd[3] = fmaf(v, wa.w, hd[3]);
            hd[4] = fmaf(v, wb.x, hd[4]);
            hd[5] = fmaf(v, wb.y, hd[5]);
            hd[6] = fmaf(v, wb.z, hd[6]);
        }
    }
    SampleOut so;
    {
        const float *hb = lds + PK_HB;
#pragma unroll
        for (int c = 0; c < 7; ++c) hd[c] = hd[c] + __shfl_xor(hd[c], 32, 64) + hb[c];
        so.sigma = softplus_(hd[0]);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            so.dif[c] = sigmoid_(hd[1 + c]);
            so.tint[c] = sigmoid_(hd[4 + c]);
        }
    }
    // Directional_MLP.mlp.0 (48 -> 64): SH part + bias pre-accumulated in dinit
    v16f d0 = dinit[0], d1 = dinit[1];
    mma_block16(d0, lds + PK_D0H, 0, lane, h1);
    mma_block16(d1, lds + PK_D0H, 4, lane, h1);
    d0 = act16(d0);
    d1 = act16(d1);
    // Directional_MLP.mlp.2 (64 -> 64) + Gaussian
    v16f e0 = load_bias(lds, 3, 0, h), e1 = load_bias(lds, 3, 1, h);
    mma_block16(e0, lds + PK_D1, 0, lane, d0);
    mma_block16(e0, lds + PK_D1, 4, lane, d1);
    mma_block16(e1, lds + PK_D1, 8, lane, d0);
    mma_block16(e1, lds + PK_D1, 12, lane, d1);
    e0 = act16(e0);
    e1 = act16(e1);
    // Directional_MLP.mlp.4 (64 -> 3) + sigmoid, on the VALU
    float c3[3] = { 0, 0, 0 };
    {
        const float4 *W = reinterpret_cast<const float4 *>(lds + PK_D2 + h * 128);
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            float4 w0 = W[g], w1 = W[16 + g];
            c3[0] = fmaf(e0[g], w0.x, c3[0]);
            c3[1] = fmaf(e0[g], w0.y, c3[1]);
            c3[2] = fmaf(e0[g], w0.z, c3[2]);
            c3[0] = fmaf(e1[g], w1.x, c3[0]);
            c3[1] = fmaf(e1[g], w1.y, c3[1]);
            c3[2] = fmaf(e1[g], w1.z, c3[2]);
        }
        const float *hb = lds + PK_HB + 8;
#pragma unroll
        for (int c = 0; c < 3; ++c) so.spec[c] = sigmoid_(c3[c] + __shfl_xor(c3[c], 32, 64) + hb[c]);
    }
    return so;
}


// ---- render-time variants (hashgrid/src/rendering_kernel.cu): any tile's decoder image and
// table, read through generic pointers (the images of all tiles stay L2-resident) ----------------

// hash-encode 8 levels at p01 in [0,1]^3 (rendering_kernel.cu:79-114: v = p01*(res-1), no (p+2)/4 step);
// same register<->level map as encode8.  Lanes with active == false issue no loads.
// STRAIGHT: no branch anywhere -- inactive lanes gather too (any in-table address) and get zeros, the corners come through
// eight single loads whatever the parity of x.  The render-time kernels' gathers are bound by their round trips (with one lane
// in eight gathering the frame takes the same time), and only straight-line code lets the loads of GATHER_BATCH levels go out
// together.
template <int DT, int GATHER_BATCH = 2, bool STRAIGHT = false>
__device__ __forceinline__ void encode8_01(const void *table, const int32_t *res, int T, int h, const float p01[3],
                                           bool active, v16f &x)
{
    const uint32_t mask = (uint32_t)T - 1u;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int level = 4 * (j >> 1) + 2 * h + (j & 1);
        float ax = 0.0f, ay = 0.0f;
        if (STRAIGHT || active) {
#pragma clang fp contract(off)
            int b[3];
            float t[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float v = p01[k] * (float)(res[3 * level + k] - 1);
                b[k] = (int)v;
                t[k] = v - (float)b[k];
            }
            uint32_t idx[8];
            float w[8];
            corner_indices(idx, b[0], b[1], b[2], mask);
            trilinear_weights(w, t[0], t[1], t[2]);
            const char *slice = (const char *)table + (size_t)level * T * TableElem<DT>::bytes;
            float2 f[8];
            if constexpr (DT != SCANERF_F32 && !STRAIGHT) {
                gather_cell<DT>(slice, idx, b[0] & 1, f);
            } else {
#pragma unroll
                for (int c = 0; c < 8; ++c) f[c] = TableElem<DT>::load(slice, idx[c]);
            }
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                ax = fmaf(w[c], f[c].x, ax);
                ay = fmaf(w[c], f[c].y, ay);
            }
            if (STRAIGHT && !active) ax = ay = 0.0f;
        }
        x[2 * j] = ax;
        x[2 * j + 1] = ay;
        if ((j + 1) % GATHER_BATCH == 0) __builtin_amdgcn_sched_barrier(0);
    }
    SCANERF_LOAD_GUARD();
}

// decode_tile with the SH part of the directional layer computed inline (per-lane direction):
// img may point to global memory (one packed image per tile).
__device__ __forceinline__ SampleOut decode_tile_dir(const float *img, int lane, const v16f &x, const float d[3],
                                                     float dnorm, float eps)
{
    const int h = lane >> 5;
    float sh[16];
    ray_sh(d, dnorm, sh, eps);
    v16f dinit[2] = { load_bias(img, 2, 0, h), load_bias(img, 2, 1, h) };
    const float *A = img + PK_D0S + (lane + (lane >> 5)) * 4;
    const float4 a00 = *reinterpret_cast<const float4 *>(A), a01 = *reinterpret_cast<const float4 *>(A + PK_GRP),
                 a10 = *reinterpret_cast<const float4 *>(A + 2 * PK_GRP), a11 = *reinterpret_cast<const float4 *>(A + 3 * PK_GRP);
    float shb[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) shb[r] = h ? sh[2 * r + 1] : sh[2 * r];
    MFMA4(dinit[0], a00, shb[0], shb[1], shb[2], shb[3])
    MFMA4(dinit[0], a01, shb[4], shb[5], shb[6], shb[7])
    MFMA4(dinit[1], a10, shb[0], shb[1], shb[2], shb[3])
    MFMA4(dinit[1], a11, shb[4], shb[5], shb[6], shb[7])
    __builtin_amdgcn_sched_barrier(0);
    return decode_tile(img, lane, x, dinit);
}

}  // namespace scanerf

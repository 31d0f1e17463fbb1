// render_h3.h -- the decoder on f16 matrix cores with split operands ("h3" arithmetic).
//
// v_mfma_f32_32x32x16_f16 runs at 16x the rate of the f32-input MFMA on gfx950, but one f16 operand
// carries 11 significant bits.  Every operand is therefore split into two f16 parts
//     v = hi + lo,   hi = f16(v),   lo = f16(v - hi)            (22 significant bits together)
// and every product is evaluated as  lo_a*hi_b + hi_a*lo_b + hi_a*hi_b  (three MFMAs, f32 accumulate,
// small terms first); the dropped lo*lo term is 2^-22 relative.  Measured against an fp64 evaluation of
// network.ShallowMLP the outputs are as close as the fp32 evaluation is (a few 1e-6 relative; the bf16
// analogue is 30x worse and a single f16 product 1000x) at 3/16 of the fp32 MFMA time.
//
// Register / lane maps are those of render_common.h: lane l = (sample s = l & 31, half h = l >> 5),
// accumulator register g of a 32-unit block = unit nmap(g, h).  A 32x32x16 MFMA consumes per lane 8 values
// of the reduction index: k-slot 8h + j (j = 0..7).  Feeding registers 8t .. 8t+7 of a block as the B operand
// of k-step t puts unit  ku(t, h, j) = 16t + 8(j >> 2) + 4h + (j & 3)  at slot 8h + j, so the weight image
// stores W[n][ku(...)] at that slot and activations never move between lanes (MI355X guide, "An accumulator
// tile as the next MFMA's operand").
//
// Image: one sub-image per (layer, 32-row block b, k-step s, part hi|lo): 64 lanes x 16 B, the upper
// half-wave shifted by 64 B (H3_SUB = 1088 B), sub-images consecutive in the order [b][s][part].  The forward
// reads it with one ds_read_b128 per lane (conflict-free).  The backward products dX = W^T dY read the SAME
// image through ds_read_b64_tr_b16 (a 4-row x 16-column block delivered column-major): with this stride the
// 32 8-byte pieces a half-wave touches fall on 32 distinct bank pairs, so no transposed copy is kept.
#pragma once
#include "render_device.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2v __attribute__((ext_vector_type(2)));
typedef float f2v __attribute__((ext_vector_type(2)));

namespace scanerf {

__host__ __device__ constexpr int h3_ku(int t, int h, int j) { return 16 * t + 8 * (j >> 2) + 4 * h + (j & 3); }
// H3_OPAQUE_ADDR (render.hip and render_time.hip): lane offsets and the base of the f32 tail are made opaque to the optimiser.
// Seeing lane >> 5 as a 0/1 value it otherwise turns every `image offset + lane offset` into its own select of two constants, and
// the f32 tail lies past 64 KB, beyond the immediate of an LDS read: one address register per weight / bias read, ~35 of them
// live across the sample loop (render-time kernel 25 -> 4 spilled registers, frame 105 -> 99 ms; training forward 51 -> 0 spills
// together with -fno-slp-vectorize, 3.35 -> 3.10 ms).  Round 2 withdrew it from the training forward because that build wrote a
// wrong encoder output in ~6 % of cold first launches; round 3 traced the fault to packed-f32 weight pairs, not to
// this addressing.
#ifndef H3_OPAQUE_ADDR
#define H3_OPAQUE_ADDR 0
#endif
__device__ __forceinline__ int h3_lane_off(int lane)
{
    int r = (lane & 31) * 16 + (lane >> 5) * 576;
#if H3_OPAQUE_ADDR
    asm volatile("" : "+v"(r));
#endif
    return r;
}

// ---- operand split
struct HL {
    h8 hi, lo;
};
// registers 8t .. 8t+7 of an accumulator block -> B operand of k-step t
// H3_MIX_SPLIT (default, round 4): lo = (f16)(x - (float)hi) as ONE instruction per element (v_fma_mixlo_f16 / v_fma_mixhi_f16:
// the f16 hi part as it is, times -1.0, plus the f32 x, the exact f32 difference rounded to f16 into one half of the destination)
// instead of v_cvt_f32_f16 + v_sub_f32 per element and a v_cvt_pk_f16_f32 per pair: 12 instead of 24 vector instructions per
// operand -- the same roundings, the same bits (render_t16.h t16_split is the same form, with what it measured).  Inline asm,
// because the compiler does not select these forms; the `s_nop 1` tied to the lo registers stands in for the two wait states the
// hazard recogniser would put between a vector instruction it can see and a matrix instruction reading its result.
// (Round 2 tried v_fma_mix_f32 here, -4 % on the render-time frame, and withdrew it because the training step then differed
// between runs: that build still held packed-f32 instructions, the cause found in round 3 -- DESIGN.md 4.10.)
#ifndef H3_MIX_SPLIT
#define H3_MIX_SPLIT 1
#endif
__device__ __forceinline__ HL split8(const v16f &v, int t)
{
    HL o;
#if H3_MIX_SPLIT
    uint32_t lo32[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f2v x = { v[8 * t + 2 * q], v[8 * t + 2 * q + 1] };
        const h2v hi = __builtin_convertvector(x, h2v);
        const uint32_t hb = __builtin_bit_cast(uint32_t, hi);
        uint32_t lb;
        asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lb) : "v"(hb), "v"(x[0]));
        asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lb) : "v"(hb), "v"(x[1]));
        lo32[q] = lb;
        o.hi[2 * q] = hi[0];
        o.hi[2 * q + 1] = hi[1];
    }
    asm("s_nop 1" : "+v"(lo32[0]), "+v"(lo32[1]), "+v"(lo32[2]), "+v"(lo32[3]));
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const h2v lo = __builtin_bit_cast(h2v, lo32[q]);
        o.lo[2 * q] = lo[0];
        o.lo[2 * q + 1] = lo[1];
    }
#else
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f2v x = { v[8 * t + 2 * q], v[8 * t + 2 * q + 1] };
        const h2v hi = __builtin_convertvector(x, h2v);
        const f2v back = __builtin_convertvector(hi, f2v);
        const f2v r = { x[0] - back[0], x[1] - back[1] };
        const h2v lo = __builtin_convertvector(r, h2v);
        o.hi[2 * q] = hi[0];
        o.hi[2 * q + 1] = hi[1];
        o.lo[2 * q] = lo[0];
        o.lo[2 * q + 1] = lo[1];
    }
#endif
#if !(defined(H3_REGIONS) && H3_REGIONS) && SCANERF_GUARDS
    asm volatile("s_nop 1" : "+v"(o.hi), "+v"(o.lo));  // operand guard, see "operand hazard" below
#endif
    return o;
}
struct HL2 {
    HL t[2];
};
__device__ __forceinline__ HL2 split16(const v16f &v)
{
    HL2 o;
    o.t[0] = split8(v, 0);
    o.t[1] = split8(v, 1);
    return o;
}

// ---- "operand hazard" (rounds 1-2; guards compiled out, see common.h SCANERF_GUARDS).
// With the f16 MFMAs scheduled freely among the VALU code that produces their B operands, 3e-4 of the 32-sample tiles came out
// wrong in lanes 16-31, differently on every launch.  Rounds 1-2 answered with wait states where an operand is produced (split8)
// or with closed scheduling regions around every MFMA group (H3_REGIONS = 1).  Round 3: the differences need packed-f32
// instructions in the kernel (the splits' x - (float)hi had become v_pk_add_f32); compiled with -fno-slp-vectorize the kernels
// are bit-reproducible without either (tools/guard_probe.py), which is how they are built.
typedef HL A2;  // an A operand (weights): the same pair of parts
__device__ __forceinline__ A2 h3_lda(const char *sub)  // `sub` = address of this lane's 16 B of the hi part
{
    A2 a;
    a.hi = *reinterpret_cast<const h8 *>(sub);
    a.lo = *reinterpret_cast<const h8 *>(sub + H3_SUB);
    return a;
}
#if defined(H3_REGIONS) && H3_REGIONS
#define H3_REGION_BEGIN() __builtin_amdgcn_sched_barrier(0)
#define H3_REGION_END()          \
    asm volatile("s_nop 1");     \
    __builtin_amdgcn_sched_barrier(0)
#else
#define H3_REGION_BEGIN()
#define H3_REGION_END()
#endif
// acc += W * B with the three-term split (small terms first); call between H3_REGION_BEGIN / END
__device__ __forceinline__ void mma3(v16f &acc, const A2 &a, const HL &b)
{
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.lo, b.hi, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.hi, b.lo, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.hi, b.hi, acc, 0, 0, 0);
}
// Two row blocks of one layer over KS k-steps: u[b] += W[b] * B.  The two accumulators are independent, so their
// MFMAs alternate (no dependent-issue stall); one region per k-step, next k-step's A operands prefetched.
template <int KS>
__device__ __forceinline__ void h3_layer2(v16f u[2], const char *img, int base, int ksb, int lo, const HL *const B[KS])
{
    A2 a0 = h3_lda(img + base + ((0 * ksb + 0) * 2) * H3_SUB + lo), a1 = h3_lda(img + base + ((1 * ksb + 0) * 2) * H3_SUB + lo);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        A2 n0 = a0, n1 = a1;
        if (s + 1 < KS) {
            n0 = h3_lda(img + base + ((0 * ksb + s + 1) * 2) * H3_SUB + lo);
            n1 = h3_lda(img + base + ((1 * ksb + s + 1) * 2) * H3_SUB + lo);
        }
        H3_REGION_BEGIN();
        u[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0.lo, B[s]->hi, u[0], 0, 0, 0);
        u[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1.lo, B[s]->hi, u[1], 0, 0, 0);
        u[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0.hi, B[s]->lo, u[0], 0, 0, 0);
        u[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1.hi, B[s]->lo, u[1], 0, 0, 0);
        u[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0.hi, B[s]->hi, u[0], 0, 0, 0);
        u[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1.hi, B[s]->hi, u[1], 0, 0, 0);
        H3_REGION_END();
        a0 = n0;
        a1 = n1;
    }
}
// One row block over KS k-steps (heads, rgb layer)
template <int KS>
__device__ __forceinline__ void h3_layer1(v16f &u, const char *img, int base, int lo, const HL *const B[KS])
{
    A2 a = h3_lda(img + base + lo);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        A2 n = a;
        if (s + 1 < KS) n = h3_lda(img + base + ((s + 1) * 2) * H3_SUB + lo);
        H3_REGION_BEGIN();
        mma3(u, a, *B[s]);
        H3_REGION_END();
        a = n;
    }
}
// sub-image address of (layer base, k-steps per block, block b, k-step s) for this lane
__device__ __forceinline__ const char *h3_sub(const char *img, int base, int ks, int b, int s, int lane_off)
{
    return img + base + ((b * ks + s) * 2) * H3_SUB + lane_off;
}
__device__ __forceinline__ v16f h3_ld16(const char *img, int byte_off)  // 16 f32 accumulator start values
{
    const float4 *p = reinterpret_cast<const float4 *>(img + byte_off);
    const float4 a = p[0], b = p[1], c = p[2], d = p[3];
    return v16f{ a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w };
}
__device__ __forceinline__ v16f h3_bias(const char *img, int layer, int blk, int h)
{
#if H3_OPAQUE_ADDR
    int base = H3_BIAS + h * 64;  // (one opaque base register for the 32 reads of the f32 tail: see h3_lane_off)
    asm("" : "+v"(base));
    return h3_ld16(img + base, (layer * 2 + blk) * 128);
#else
    return h3_ld16(img, H3_BIAS + (((layer * 2 + blk) * 2 + h) * 16) * 4);
#endif
}

// ------------------------------------------------------------------ backward primitives
typedef short s4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s4v lds_s4v;

// ds_read_b64_tr_b16: per 16-lane group a block of 4 rows x 16 columns of 16-bit elements, delivered column-major.
// Lane 4q+p of the group supplies the address of row q, columns 4p..4p+3 (8 bytes); lane i receives column i,
// row q in element q.  EXEC must be all ones.
__device__ __forceinline__ h4 ds_tr4(const char *p)
{
    const s4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v *)(p));
    return __builtin_bit_cast(h4, v);
}
__device__ __forceinline__ h8 cat44(h4 a, h4 b) { return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7); }

// Lane-dependent address terms of the backward primitives, computed once per tile from an opaque lane index
// (8 registers): every LDS access below is then one base register plus an immediate offset.
struct H3Lane {
    int fwd;           // forward image: this lane's 16 B of a sub-image (h3_lane_off)
    int lT;            // transposed image reads (h3_lda_T)
    int pv, pb;        // staging writes: chunk (t, a) of block b at pb + (((8b + 4t + 2a) ^ pv) << 3)
    int g0[2], g1[2];  // staging reads of block b: samples 8h..8h+3 / 8h+4..8h+7 of k-step 0 (k-step 1: + 2048)
};
__device__ __forceinline__ H3Lane h3_lane(int lane)
{
    H3Lane L;
    const int h = lane >> 5, g16 = (lane >> 4) & 1, q = (lane >> 2) & 3, p = lane & 3, s = lane & 31;
    L.fwd = h3_lane_off(lane);
    L.lT = (4 * h + q) * 16 + (p & 1) * 576 + (p >> 1) * 8 + g16 * 2 * H3_SUB;
    L.pv = (((s >> 1) & 7) | (((s >> 1) & 1) << 3)) ^ h;
    L.pb = s * 128;
    const int gq = (4 * h + (q >> 1)) | (((q >> 1) & 1) << 3);  // g(s) of the first four samples (independent of the k-step)
    const int gv = (4 * g16 + p) ^ gq;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        L.g0[b] = (8 * h + q) * 128 + ((gv ^ (8 * b)) << 3);
        L.g1[b] = (8 * h + q + 4) * 128 + ((gv ^ (8 * b) ^ 2) << 3);
    }
    return L;
}

// A operand of a TRANSPOSED product dX = W^T dY from the forward image of a layer (ksb k-steps per row block):
// reduction over the layer's output units n of row block nb, k-step tq (slot 8h'+j' <-> n = 32nb + ku(tq, h', j'),
// i.e. the same slot<->unit map as B operands made from accumulator registers), output rows = input units
// 32ib + (lane & 31).  Two transposed reads per part; conflict-free (see the header comment).
__device__ __forceinline__ A2 h3_lda_T(const char *img, int base, int ksb, int nb, int tq, int ib, const H3Lane &L)
{
    const char *a = img + L.lT + (base + ((nb * ksb + 2 * ib) * 2) * H3_SUB + tq * 256);
    A2 r;
    r.hi = cat44(ds_tr4(a), ds_tr4(a + 8 * 16));
    r.lo = cat44(ds_tr4(a + H3_SUB), ds_tr4(a + H3_SUB + 8 * 16));
    return r;
}

// ---- per-wave staging image for products that reduce over SAMPLES (weight gradients dW = dY X^T): both operands
// need the unit on the lane and 8 samples per k-slot group, the transpose of the register layout.  A matrix part
// (hi or lo) of up to 64 units x 32 samples is kept as 8-byte chunks (sample s, unit quad uq) at
//     s*128 + ((uq ^ g(s)) * 8),   g(s) = ((s >> 1) & 7) | (((s >> 1) & 1) << 3)
// so that the writes (16 consecutive samples, one quad) and the transposed reads (4 consecutive samples x 8
// consecutive quads) both touch 32 distinct bank pairs.
constexpr int H3_STAGE_PART = 32 * 128;            // bytes per matrix part
constexpr int H3_STAGE_MAT = 2 * H3_STAGE_PART;    // hi + lo
__device__ __forceinline__ int h3_stage_off(int s, int uq) { return s * 128 + ((uq ^ (((s >> 1) & 7) | (((s >> 1) & 1) << 3))) << 3); }

// registers of one 32-unit block (lane = sample) -> chunks of units 32b .. 32b+31
__device__ __forceinline__ void h3_stage_put(char *mat, const H3Lane &L, int b, const HL2 &v)
{
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int off = L.pb + (((8 * b + 4 * t + 2 * a) ^ L.pv) << 3);
            const h8 &hi = v.t[t].hi, &lo = v.t[t].lo;
            *reinterpret_cast<h4 *>(mat + off) = h4{ hi[4 * a], hi[4 * a + 1], hi[4 * a + 2], hi[4 * a + 3] };
            *reinterpret_cast<h4 *>(mat + H3_STAGE_PART + off) = h4{ lo[4 * a], lo[4 * a + 1], lo[4 * a + 2], lo[4 * a + 3] };
        }
}
// operand (A or B alike) of k-step t for units 32b + (lane & 31): slot 8h + j <-> sample 16t + 8h + j;
// o0 / o1 = L.g0[b] / L.g1[b]
__device__ __forceinline__ HL h3_stage_get(const char *mat, int o0, int o1, int t)
{
    HL r;
    r.hi = cat44(ds_tr4(mat + o0 + 2048 * t), ds_tr4(mat + o1 + 2048 * t));
    r.lo = cat44(ds_tr4(mat + o0 + (H3_STAGE_PART + 2048 * t)), ds_tr4(mat + o1 + (H3_STAGE_PART + 2048 * t)));
    return r;
}
// sum of the 8 slots of an operand (for bias gradients): hi and lo parts, f32 accumulate
__device__ __forceinline__ float h3_sum8(const HL &v, float acc)
{
    const h2v one = { (_Float16)1.0f, (_Float16)1.0f };
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        acc = __builtin_amdgcn_fdot2(h2v{ v.lo[2 * q], v.lo[2 * q + 1] }, one, acc, false);
        acc = __builtin_amdgcn_fdot2(h2v{ v.hi[2 * q], v.hi[2 * q + 1] }, one, acc, false);
    }
    return acc;
}

// Dir layer-0 accumulator start of one ray: bias + W[:, 32:48] SH(dir)   (one k-step per block)
__device__ __forceinline__ void h3_dinit(const char *img, int lane, const float sh[16], v16f dinit[2])
{
    const int h = lane >> 5, lo = h3_lane_off(lane);
    v16f s16;  // registers 0..7 = this half's 8 SH values (slot 8h + j = SH[8h + j])
#pragma unroll
    for (int j = 0; j < 8; ++j) s16[j] = h ? sh[8 + j] : sh[j];
#pragma unroll
    for (int j = 8; j < 16; ++j) s16[j] = 0.0f;
    const HL b = split8(s16, 0);
    dinit[0] = h3_bias(img, 2, 0, h);
    dinit[1] = h3_bias(img, 2, 1, h);
    const A2 a0 = h3_lda(h3_sub(img, H3_D0, 3, 0, 2, lo)), a1 = h3_lda(h3_sub(img, H3_D0, 3, 1, 2, lo));
    H3_REGION_BEGIN();
    mma3(dinit[0], a0, b);
    mma3(dinit[1], a1, b);
    H3_REGION_END();
}

__device__ __forceinline__ v16f act16_fast(const v16f &x)
{
    v16f r;
#pragma unroll
    for (int g = 0; g < 16; ++g) r[g] = gauss_fast(x[g]);
    return r;
}

// Decoder forward on one 32-sample tile (same contract as decode_tile in render_device.h).
// DIRS: dinit is not used; the SH part of Directional_MLP.mlp.0 is computed from the lane's direction `dir` where that layer
// starts (same products and sums as h3_dinit + this function: bias, then the SH k-step, then the two H k-steps), so that the 32
// registers of dinit are not live through the first three layers (the kernels at three and four waves per SIMD).
template <bool DIRS = false>
__device__ __forceinline__ SampleOut decode_tile_h3(const char *img, int lane, const v16f &x, const v16f dinit[2], const float *dir = nullptr,
                                                    float eps = 0.0f)
{
    const int h = lane >> 5, lo = h3_lane_off(lane);
    // Spatial_MLP.mlp.0 (32 -> 64) + Gaussian
    HL2 a[2];
    {
        const HL2 xs = split16(x);
        v16f u[2] = { h3_bias(img, 0, 0, h), h3_bias(img, 0, 1, h) };
        const HL *const B[2] = { &xs.t[0], &xs.t[1] };
        h3_layer2<2>(u, img, H3_L0, 2, lo, B);
        a[0] = split16(act16_fast(u[0]));
        a[1] = split16(act16_fast(u[1]));
    }
    // Spatial_MLP.mlp.2 (64 -> 64), linear
    HL2 H[2];
    {
        v16f u[2] = { h3_bias(img, 1, 0, h), h3_bias(img, 1, 1, h) };
        const HL *const B[4] = { &a[0].t[0], &a[0].t[1], &a[1].t[0], &a[1].t[1] };
        h3_layer2<4>(u, img, H3_L1, 4, lo, B);
        H[0] = split16(u[0]);
        H[1] = split16(u[1]);
    }
    SampleOut so;
    {   // heads on H[:32]: one 32-row block, rows replicated so both halves hold all 7 outputs
        v16f u = h3_ld16(img, H3_HB);
        const HL *const B[2] = { &H[0].t[0], &H[0].t[1] };
        h3_layer1<2>(u, img, H3_HEAD, lo, B);
        so.sigma = softplus_fast(u[0]);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            so.dif[c] = sigmoid_fast(u[1 + c]);
            so.tint[c] = sigmoid_fast(u[4 + c]);
        }
    }
    // Directional_MLP.mlp.0 (48 -> 64): SH part + bias pre-accumulated in dinit
    HL2 c0[2];
    {
        v16f u[2];
        if constexpr (DIRS) {
            float sh[16];
            ray_sh(dir, sqrtf(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]), sh, eps);
            h3_dinit(img, lane, sh, u);
        } else {
            u[0] = dinit[0];
            u[1] = dinit[1];
        }
        const HL *const B[2] = { &H[1].t[0], &H[1].t[1] };
        h3_layer2<2>(u, img, H3_D0, 3, lo, B);
        c0[0] = split16(act16_fast(u[0]));
        c0[1] = split16(act16_fast(u[1]));
    }
    // Directional_MLP.mlp.2 (64 -> 64) + Gaussian
    HL2 c1[2];
    {
        v16f u[2] = { h3_bias(img, 3, 0, h), h3_bias(img, 3, 1, h) };
        const HL *const B[4] = { &c0[0].t[0], &c0[0].t[1], &c0[1].t[0], &c0[1].t[1] };
        h3_layer2<4>(u, img, H3_D1, 4, lo, B);
        c1[0] = split16(act16_fast(u[0]));
        c1[1] = split16(act16_fast(u[1]));
    }
    {   // Directional_MLP.mlp.4 (64 -> 3) + sigmoid
        v16f u = h3_ld16(img, H3_D2B);
        const HL *const B[4] = { &c1[0].t[0], &c1[0].t[1], &c1[1].t[0], &c1[1].t[1] };
        h3_layer1<4>(u, img, H3_D2, lo, B);
#pragma unroll
        for (int c = 0; c < 3; ++c) so.spec[c] = sigmoid_fast(u[c]);
    }
    return so;
}

}  // namespace scanerf

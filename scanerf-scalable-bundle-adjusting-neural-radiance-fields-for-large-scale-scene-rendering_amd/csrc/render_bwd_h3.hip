// render_bwd_h3.hip -- fused per-ray volume rendering, backward, on split-f16 matrix cores (gfx950).
//
// Same adjoint as render_bwd.hip (hashgrid/__init__.py:512-596 under autograd in the reference) with the decoder
// products in the h3 arithmetic of render_h3.h (three f16 MFMAs per term, f32 accumulate: 3/16 of the f32-MFMA time
// at f32-equivalent accuracy).  With the matrix work that cheap the kernel is organised around what remains:
//
//   * ONE WAVE PER RAY, its 32-sample tiles walked last -> first, so the compositing adjoint's suffix sums are a
//     register carry; the forward recompute and the activation gradients of a ray never leave its wave.
//   * WEIGHT GRADIENTS ARE OWNED BY BLOCK.  A wave that kept all of dW for its own tiles would hold 240 accumulator
//     registers next to ~330 registers of recompute state: measured, that spills ~900 registers and runs 2x slower
//     than the f32 kernel (one wave per SIMD exposes every scratch round trip).  Instead the 4 waves of a workgroup
//     walk their rays in lock step and each accumulates ONE 32x32 block of every layer over the 4 waves' tiles
//     (5 accumulators), reading the others' staged operands; two workgroup barriers per layer step.  Partials are
//     flushed once; k_reduce_dw adds them in a fixed order (deterministic).
//   * Products that reduce over units  (forward recompute H = W X, activation gradients dX = W^T dY) take their B
//     operand straight from accumulator registers; W^T comes from the SAME LDS image as W through transposed reads
//     (h3_lda_T), so one 69 KB image serves both directions.
//   * Products that reduce over samples (weight gradients dW = dY X^T) need the transposed register layout: the wave
//     writes the already split f16 operands to its 16 KB staging image and the owners read them back with
//     ds_read_b64_tr_b16 (h3_stage_put / h3_stage_get, conflict-free both ways); bias gradients are row sums of the
//     same operands (v_dot2_f32_f16).
//   * G'(u) = -100 u G(u) is kept from the recompute (one multiply) instead of being re-derived per use.
//   * GRADIENT RANGE.  Upstream gradients of a mean loss are ~1/(3B) and are multiplied by compositing weights down
//     to 1e-8: far below f16's normal range (6e-5).  Every workgroup therefore carries a power-of-two scale 2^K: the
//     pre-activation gradients of its 4 tiles are multiplied by it before they enter any f16 operand, so that the
//     largest one lies in [2^-8, 2^6]; K moves only when the tiles leave that window, and then all accumulators are
//     rescaled by the (exact) power of two.  Feature gradients are unscaled per tile, the weight-gradient
//     partials once at the flush.  MODE.FP16_OVFL clamps f32 -> f16 conversions at +-65504 instead of producing
//     infinities, as a safety net for pathological amplification through the layers.
#include <stdlib.h>

#include "render_bwd_common.h"
#include "render_h3.h"

using namespace scanerf;

namespace {

constexpr int kThreads = 256;
constexpr int kLdsRes = H3_BYTES;                          // resolutions [16][4] i32
constexpr int kLdsDinit = kLdsRes + 256;                   // 4 waves x [block 2][half 2][16] f32: Dir layer-0 start of the wave's ray
constexpr int kLdsSh = kLdsDinit + 4 * 256;                // 4 waves x SH[16] of the wave's ray
constexpr int kLdsMx = kLdsSh + 4 * 64;                    // 4 floats (padded to 64 B)
constexpr int kLdsStage = kLdsMx + 64;                     // 4 waves x {Y, X} staging matrices
constexpr int kLdsCursor = kLdsStage + 4 * 2 * H3_STAGE_MAT;  // record cursors (fused scatter producer only)


// A lane index the optimiser cannot trace back: LDS addresses derived from it are recomputed where they are used.
// Derived from the plain lane index they are loop invariants, and the ~200 of them this kernel needs are hoisted
// out of the tile loop and held in registers for its whole duration (measured: +40..120 live registers per step).
__device__ __forceinline__ int fresh(int lane)
{
    asm volatile("" : "+v"(lane));
    return lane;
}

// G(u) and G'(u) of a block: u <- G(u), d <- -100 u G(u)
__device__ __forceinline__ void act_and_deriv(v16f &u, v16f &d)
{
#pragma unroll
    for (int g = 0; g < 16; ++g) {
        const float q = gauss_fast(u[g]);
        d[g] = -100.0f * u[g] * q;
        u[g] = q;
    }
}
__device__ __forceinline__ v16f mul16(const v16f &a, const v16f &b)
{
    v16f r;
#pragma unroll
    for (int g = 0; g < 16; ++g) r[g] = a[g] * b[g];
    return r;
}

// dX (rows 32ib..32ib+31 of the layer's input) = W^T dY over the 64 output units (row blocks 0,1 x k-steps 0,1)
template <int NIB>
__device__ __forceinline__ void chain64(v16f dx[NIB], const char *img, int base, int ksb, const H3Lane &L, const HL2 dy[2])
{
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int tq = 0; tq < 2; ++tq) {
            HL a[NIB];
#pragma unroll
            for (int ib = 0; ib < NIB; ++ib) a[ib] = h3_lda_T(img, base, ksb, nb, tq, ib, L);
            H3_REGION_BEGIN();
#pragma unroll
            for (int ib = 0; ib < NIB; ++ib) dx[ib] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ib].lo, dy[nb].t[tq].hi, dx[ib], 0, 0, 0);
#pragma unroll
            for (int ib = 0; ib < NIB; ++ib) dx[ib] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ib].hi, dy[nb].t[tq].lo, dx[ib], 0, 0, 0);
#pragma unroll
            for (int ib = 0; ib < NIB; ++ib) dx[ib] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ib].hi, dy[nb].t[tq].hi, dx[ib], 0, 0, 0);
            H3_REGION_END();
        }
}
// dX (NIB row blocks) = W^T dY for a layer with ONE 32-row output block whose live rows sit in k-step 0 (heads, rgb)
template <int NIB>
__device__ __forceinline__ void chain_narrow(v16f dx[NIB], const char *img, int base, int ksb, const H3Lane &L, const HL &dy)
{
    HL a[NIB];
#pragma unroll
    for (int ib = 0; ib < NIB; ++ib) a[ib] = h3_lda_T(img, base, ksb, 0, 0, ib, L);
    H3_REGION_BEGIN();
#pragma unroll
    for (int ib = 0; ib < NIB; ++ib) mma3(dx[ib], a[ib], dy);
    H3_REGION_END();
}

// Weight-gradient block owned by this wave: acc += sum over slots [slot0, slot0+NS) of dY_slot[yb] X_slot[xb]^T, the
// operands read back (transposed) from the staging images of the NS waves' tiles; rowsum += row sums of dY (lane = unit,
// this half-wave's samples).  x_in_y: the X operand sits in the Y matrix (heads: H[:32] is parked beside the narrow block).
template <int NS, int ROWSUM_FROM = 0, int ROWSUM_TO = 2 * NS>
__device__ __forceinline__ void wgrad_block(v16f &acc, float &rowsum, const char *stage, int slot0, const H3Lane &L, int yb,
                                            bool x_in_y, int xb, int rs_shift = 0)
{
    // block selection is wave-uniform at run time: pick the lane terms once
    const int yo0 = yb ? L.g0[1] : L.g0[0], yo1 = yb ? L.g1[1] : L.g1[0];
    const int xo0 = (xb ? L.g0[1] : L.g0[0]) + (x_in_y ? 0 : H3_STAGE_MAT), xo1 = (xb ? L.g1[1] : L.g1[0]) + (x_in_y ? 0 : H3_STAGE_MAT);
    const char *m0 = stage + slot0 * 2 * H3_STAGE_MAT;
    HL a = h3_stage_get(m0, yo0, yo1, 0), b = h3_stage_get(m0, xo0, xo1, 0);
#pragma unroll
    for (int i = 0; i < 2 * NS; ++i) {
        HL an = a, bn = b;
        if (i + 1 < 2 * NS) {
            const char *m = m0 + ((i + 1) >> 1) * 2 * H3_STAGE_MAT;
            an = h3_stage_get(m, yo0, yo1, (i + 1) & 1);
            bn = h3_stage_get(m, xo0, xo1, (i + 1) & 1);
        }
        // row sums (bias gradients) only over steps [ROWSUM_FROM, ROWSUM_TO) + rs_shift: the two owners of a row
        // block share that work
        if (ROWSUM_TO > ROWSUM_FROM && i >= ROWSUM_FROM + rs_shift && i < ROWSUM_TO + rs_shift) rowsum = h3_sum8(a, rowsum);
        H3_REGION_BEGIN();
        mma3(acc, a, b);
        H3_REGION_END();
        a = an;
        b = bn;
    }
}

template <int DT>
__global__ void __launch_bounds__(kThreads, 1) k_render_bwd_h3(BwdArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    int *lres = reinterpret_cast<int *>(lds + kLdsRes);
    uint32_t *cursor = reinterpret_cast<uint32_t *>(lds + kLdsCursor);
    float *shbuf = reinterpret_cast<float *>(lds + kLdsSh);   // [4 waves][16]: SH of each wave's ray
    float *mxbuf = reinterpret_cast<float *>(lds + kLdsMx);   // [4]: each wave's largest |pre-activation gradient| of the tile
    {
        const float4 *src = reinterpret_cast<const float4 *>(a.f.packed + PK_TOTAL);
        float4 *dst = reinterpret_cast<float4 *>(lds);
        for (int i = threadIdx.x; i < H3_BYTES / 16; i += kThreads) dst[i] = src[i];
        if (threadIdx.x < 64) {
            const int lv = threadIdx.x >> 2, c = threadIdx.x & 3;
            lres[threadIdx.x] = c < 3 ? a.f.resolutions[3 * lv + c] : 0;
        }
        float4 *stz = reinterpret_cast<float4 *>(lds + kLdsStage);  // finite contents for the unused rows of narrow blocks
        for (int i = threadIdx.x; i < 4 * 2 * H3_STAGE_MAT / 16; i += kThreads) stz[i] = make_float4(0, 0, 0, 0);
        if (a.recs) {
            const int nbins = 16 * a.bins.NB;
            for (int i = threadIdx.x; i < nbins; i += kThreads)
                cursor[i] = a.bin_starts[i] + a.bin_rowprefix[(size_t)i * a.bins.W + blockIdx.x];
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, sl = lane & 31, h = lane >> 5;
    const char *stage = lds + kLdsStage;
    char *stY = lds + kLdsStage + wv * 2 * H3_STAGE_MAT, *stX = stY + H3_STAGE_MAT;
    const int S = a.f.S, ntiles = (S + 31) >> 5;
    const v16f zero16 = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    __builtin_amdgcn_s_setreg(1 | (23 << 6), 1);  // hwreg(HW_REG_MODE, 23, 1) = FP16_OVFL: f16 conversions saturate

    // ---- ownership of the weight-gradient blocks (the 4 waves' tiles are summed by the owner)
    const int rb = wv >> 1, cb = wv & 1;   // 64x64 layers: block (rb, cb) over all 4 slots
    const int s2 = 2 * cb;                 // 64x32 layers: row block rb over slots s2, s2+1
    v16f gW_D1 = zero16, gW_L1 = zero16, gW_D0H = zero16, gW_L0 = zero16;
    v16f gW_nar = zero16;                  // wave 0: heads (rows 0-6); waves 1, 2: rgb layer, input block wv-1 (rows 8-10)
    float gW_D0S[8];                       // unit 32rb + sl, SH index 8h + j, own slots
    float gB_D1 = 0.0f, gB_L1 = 0.0f, gB_D0 = 0.0f, gB_L0 = 0.0f;  // lane = unit 32rb + sl, this half's samples
    float gB_head[7], gB_d2[3];            // lane = sample, own tiles
    float gmax = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i) gW_D0S[i] = 0.0f;
#pragma unroll
    for (int i = 0; i < 7; ++i) gB_head[i] = 0.0f;
#pragma unroll
    for (int i = 0; i < 3; ++i) gB_d2[i] = 0.0f;
    int K = 0;                             // gradient scale 2^K of the workgroup (identical in its 4 waves)
    float sc = 1.0f, isc = 1.0f;

    const int ngroups_all = (a.f.B + 3) >> 2;
    for (int grp = blockIdx.x; grp < ngroups_all; grp += gridDim.x) {
        const int ray = 4 * grp + wv;
        const bool active = ray < a.f.B && !(a.f.ray_valid && !a.f.ray_valid[ray]);  // wave-uniform
        const int rayc = active ? ray : 0;  // inactive waves run on ray 0's geometry with zero inputs and gradients
        if (ray < a.f.B && !active && a.dfeat)  // invalid ray: zero feature gradients
            for (int s = lane >> 1; s < S; s += 32)
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    reinterpret_cast<float2 *>(a.dfeat)[(size_t)(2 * j + (lane & 1)) * a.f.B * S + (size_t)ray * S + s] =
                        make_float2(0, 0);
        float o[3], d[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            o[k] = a.f.rays_o[3 * rayc + k];
            d[k] = a.f.rays_d[3 * rayc + k];
        }
        const float dnorm = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        const float *go = a.grad_out + (size_t)rayc * SCANERF_RAY_OUT;
        const float *fo = a.f.out_ray + (size_t)rayc * SCANERF_RAY_OUT;
        {   // per ray: SH (published for the owners of the SH part) and the Dir layer-0 accumulator start, parked in LDS
            float sh[16];
            ray_sh(d, dnorm, sh);
            v16f dinit[2];
            h3_dinit(lds, lane, sh, dinit);
            if (sl == 0) {
                float4 *dp = reinterpret_cast<float4 *>(lds + kLdsDinit + wv * 256 + h * 64);
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        dp[b * 8 + q] = make_float4(dinit[b][4 * q], dinit[b][4 * q + 1], dinit[b][4 * q + 2], dinit[b][4 * q + 3]);
            }
            if (lane < 16) {
                float v = 0.0f;
#pragma unroll
                for (int j = 0; j < 16; ++j) v = lane == j ? sh[j] : v;
                shbuf[wv * 16 + lane] = v;
            }
        }
        float Rcarry = 0.0f;              // sum of a_j w_j over all later tiles of the ray
        float slot_rs[2] = { 0, 0 };      // owner: per own slot, sum over the ray of dL/d(dir layer-0 pre-activation), unit 32rb + sl
        __syncthreads();                  // shbuf / dinit visible; the previous group's staging reads are complete

        for (int tile = ntiles - 1; tile >= 0; --tile) {
            const int s = tile * 32 + sl;
            const bool live = s < S;
            const float z = live ? a.f.z_vals[(size_t)rayc * S + s] : 0.0f;
            const float dist_i = live ? a.f.dists[(size_t)rayc * S + s] : 0.0f;
            float delta = dist_i * dnorm;
            if (a.f.infinity && s == S - 1) delta = 1e10f;

            const H3Lane L = h3_lane(fresh(lane));  // this tile's lane address terms (not loop invariants: see fresh())
            const int hf = L.fwd >= 576;             // = lane >> 5, derived from the opaque index
            // ================= forward recompute =================
            v16f x;
            if (!active) {
                x = zero16;
            } else if (a.xstash) {
                const float4 *xs = reinterpret_cast<const float4 *>(a.xstash + ((size_t)ray * S + (live ? s : 0)) * 32 + 16 * h);
                const float4 q0 = xs[0], q1 = xs[1], q2 = xs[2], q3 = xs[3];
                x = v16f{ q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w };
            } else {
                float p[3];
                contract_point(a.f, o, d, z, p);
                encode8<DT, 2>(a.f, lres, h, p, x);
            }
            // warm the caches with the NEXT tile's stashed inputs (one line per lane): at one wave per SIMD the miss
            // at the top of every tile is otherwise fully exposed
            float warm = 0.0f;
            if (active && a.xstash && tile > 0 && live)
                warm = a.xstash[((size_t)ray * S + s - 32) * 32 + 16 * h];
            const HL2 xs2 = split16(x);
            // Only H, G'(v1) and the inputs x are held across the backward steps; the two other Gaussian layers are
            // recomputed where their gradients are formed (12 MFMAs each) -- holding them spills (see the header).
            HL2 Hs[2];
            v16f dv1f[2];
            {
                HL2 a0s[2];
                {
                    v16f u[2] = { h3_bias(lds, 0, 0, hf), h3_bias(lds, 0, 1, hf) };
                    const HL *const B[2] = { &xs2.t[0], &xs2.t[1] };
                    h3_layer2<2>(u, lds, H3_L0, 2, L.fwd, B);
                    a0s[0] = split16(act16_fast(u[0]));
                    a0s[1] = split16(act16_fast(u[1]));
                }
                v16f u[2] = { h3_bias(lds, 1, 0, hf), h3_bias(lds, 1, 1, hf) };
                const HL *const B[4] = { &a0s[0].t[0], &a0s[0].t[1], &a0s[1].t[0], &a0s[1].t[1] };
                h3_layer2<4>(u, lds, H3_L1, 4, L.fwd, B);
                Hs[0] = split16(u[0]);
                Hs[1] = split16(u[1]);
            }
            float sigma, dsig_dpre, dif[3], tint[3], spec[3];
            {
                v16f u = h3_ld16(lds, H3_HB);
                const HL *const B[2] = { &Hs[0].t[0], &Hs[0].t[1] };
                h3_layer1<2>(u, lds, H3_HEAD, L.fwd, B);
                sigma = softplus_(u[0]);
                dsig_dpre = u[0] > 20.0f ? 1.0f : sigmoid_fast(u[0]);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    dif[c] = sigmoid_fast(u[1 + c]);
                    tint[c] = sigmoid_fast(u[4 + c]);
                }
            }
            h3_stage_put(stY, L, 1, Hs[0]);  // X operand of the heads' weight gradient: upper half of this wave's Y image
            {
                HL2 c0s[2], c1s[2];
                {
                    v16f u[2] = { h3_ld16(lds, kLdsDinit + wv * 256 + (hf) * 64), h3_ld16(lds, kLdsDinit + wv * 256 + 128 + (hf) * 64) };
                    const HL *const B[2] = { &Hs[1].t[0], &Hs[1].t[1] };
                    h3_layer2<2>(u, lds, H3_D0, 3, L.fwd, B);
                    c0s[0] = split16(act16_fast(u[0]));
                    c0s[1] = split16(act16_fast(u[1]));
                }
                {
                    v16f u[2] = { h3_bias(lds, 3, 0, hf), h3_bias(lds, 3, 1, hf) };
                    const HL *const B[4] = { &c0s[0].t[0], &c0s[0].t[1], &c0s[1].t[0], &c0s[1].t[1] };
                    h3_layer2<4>(u, lds, H3_D1, 4, L.fwd, B);
                    act_and_deriv(u[0], dv1f[0]);
                    act_and_deriv(u[1], dv1f[1]);
                    c1s[0] = split16(u[0]);
                    c1s[1] = split16(u[1]);
                }
                v16f u = h3_ld16(lds, H3_D2B);
                const HL *const B[4] = { &c1s[0].t[0], &c1s[0].t[1], &c1s[1].t[0], &c1s[1].t[1] };
                h3_layer1<4>(u, lds, H3_D2, L.fwd, B);
#pragma unroll
                for (int c = 0; c < 3; ++c) spec[c] = sigmoid_fast(u[c]);
                h3_stage_put(stX, L, 0, c1s[0]);  // X operand of the rgb layer's weight gradient
                h3_stage_put(stX, L, 1, c1s[1]);
            }

            // ================= compositing: recompute and adjoint =================
            const float ex = live ? expf(-sigma * delta) : 1.0f;  // 1 - alpha
            const float alpha = 1.0f - ex;
            const float fi = 1.0f - alpha + 1e-6f;
            float incl = fi;
#pragma unroll
            for (int off = 1; off < 32; off <<= 1) {
                const float t = __shfl_up(incl, off, 32);
                if (sl >= off) incl *= t;
            }
            float excl = __shfl_up(incl, 1, 32);
            if (sl == 0) excl = 1.0f;
            const float Ti = a.tile_T[(size_t)rayc * ((S + 15) >> 4) + 2 * tile] * excl;
            const float w = alpha * Ti;
            float gD[3], gS[3], gTi[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float pre = fo[5 + c] + fo[8 + c];  // diffuse + specular before the clamp
                const float grgb = (pre >= 0.0f && pre <= 1.0f) ? go[c] : 0.0f;
                gD[c] = go[5 + c] + grgb;
                gS[c] = go[8 + c] + grgb;
                gTi[c] = go[11 + c];
            }
            const float gDepth = go[3], gTl = go[4], gW2 = go[14], Tl = fo[4];
            float ai = gDepth * z;
#pragma unroll
            for (int c = 0; c < 3; ++c) ai += gD[c] * dif[c] + gS[c] * tint[c] * spec[c] + gTi[c] * tint[c];
            const float aw = live ? ai * w : 0.0f;
            float rs = aw;  // inclusive suffix sum inside the tile
#pragma unroll
            for (int off = 1; off < 32; off <<= 1) {
                const float t = __shfl_down(rs, off, 32);
                if (sl + off < 32) rs += t;
            }
            const float suffix = Rcarry + rs - aw;
            Rcarry += __shfl(rs, 0, 32);
            float dalpha = Ti * ai - (suffix + ((s < S - 1) ? gTl * Tl : 0.0f)) / fi;
            if (!live) dalpha = 0.0f;
            const float dsigma = dalpha * delta * ex;
            if (a.g_dnorm && active) {
                const float dd = (a.f.infinity && s == S - 1) ? 0.0f : dist_i;  // the infinity sample's delta is a constant
                const float gd = half_sum(dalpha * sigma * ex * dd);
                if (lane == 0) a.g_dnorm[(size_t)ray * ntiles + tile] = gd;
            }
            float gh[7], gs3[3];  // gradients w.r.t. the head / rgb pre-activations
            gh[0] = dsigma * dsig_dpre;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                gh[1 + c] = w * gD[c] * dif[c] * (1.0f - dif[c]);
                gh[4 + c] = w * (gS[c] * spec[c] + gTi[c]) * tint[c] * (1.0f - tint[c]);
                gs3[c] = (w * gS[c] * tint[c] + gW2 * w * 2.0f * spec[c]) * spec[c] * (1.0f - spec[c]);
            }
            if (!active) {  // select, not multiply: the inputs of an inactive wave may be anything
#pragma unroll
                for (int c = 0; c < 7; ++c) gh[c] = 0.0f;
#pragma unroll
                for (int c = 0; c < 3; ++c) gs3[c] = 0.0f;
            }
            {   // this tile's largest |gradient|, published for the workgroup's scale
                float mx = 0.0f;
#pragma unroll
                for (int c = 0; c < 7; ++c) mx = fmaxf(mx, fabsf(gh[c]));
#pragma unroll
                for (int c = 0; c < 3; ++c) mx = fmaxf(mx, fabsf(gs3[c]));
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
                if (lane == 0) mxbuf[wv] = mx;
            }
            __syncthreads();  // ---- S: tile maxima visible (and the staged X operands of the narrow step)
            {   // gradient scale (see the header): keep the workgroup's largest |gradient| * 2^K in [2^-8, 2^6]
                const float mx = fmaxf(fmaxf(mxbuf[0], mxbuf[1]), fmaxf(mxbuf[2], mxbuf[3]));
                const float ms = mx * sc;
                if (mx > 0.0f && mx < 3.0e38f && (ms > 64.0f || ms < 0.00390625f)) {
                    int e;
                    frexpf(mx, &e);  // mx = f * 2^e, f in [0.5, 1)
                    int Kn = -e;
                    Kn = Kn > K + 100 ? K + 100 : (Kn < K - 100 ? K - 100 : Kn);  // rescale factor stays an f32 power of two
                    Kn = Kn > 100 ? 100 : (Kn < -100 ? -100 : Kn);
                    const float r = ldexpf(1.0f, Kn - K);
                    K = Kn;
                    sc = ldexpf(1.0f, K);
                    isc = ldexpf(1.0f, -K);
                    gW_D1 *= r; gW_L1 *= r; gW_D0H *= r; gW_L0 *= r; gW_nar *= r;
                    gB_D1 *= r; gB_L1 *= r; gB_D0 *= r; gB_L0 *= r; slot_rs[0] *= r; slot_rs[1] *= r;
#pragma unroll
                    for (int j = 0; j < 8; ++j) gW_D0S[j] *= r;
#pragma unroll
                    for (int c = 0; c < 7; ++c) gB_head[c] *= r;
#pragma unroll
                    for (int c = 0; c < 3; ++c) gB_d2[c] *= r;
                }
#pragma unroll
                for (int c = 0; c < 7; ++c) gh[c] *= sc;
#pragma unroll
                for (int c = 0; c < 3; ++c) gs3[c] *= sc;
            }
#pragma unroll
            for (int c = 0; c < 7; ++c) gB_head[c] += gh[c];
#pragma unroll
            for (int c = 0; c < 3; ++c) gB_d2[c] += gs3[c];

            // ================= narrow layers: rgb (64 -> 3) and heads (32 -> 7) =================
            // one staged 32-row block: rows 0-3 sigma,dif; 4-6 tint; 8-10 rgb
            HL nar;      // B operand of the transposed products: half 0 carries the rows, half 1 zeros (its rows are replicas)
            HL narrgb;
            {
                v16f t16 = zero16, r16 = zero16;
                if (h == 0) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) t16[c] = gh[c];
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        t16[4 + c] = gh[4 + c];  // slot j = 4..6 of half 0 <-> image rows 8..10
                        r16[c] = gs3[c];
                    }
                }
                nar = split8(t16, 0);
                narrgb = split8(r16, 0);
                v16f st = zero16;
#pragma unroll
                for (int c = 0; c < 4; ++c) st[c] = h == 0 ? gh[c] : (c < 3 ? gh[4 + c] : 0.0f);
                const HL stq = split8(st, 0);
                v16f st2 = zero16;
#pragma unroll
                for (int c = 0; c < 3; ++c) st2[c] = h == 0 ? gs3[c] : 0.0f;
                const HL stq2 = split8(st2, 0);
                const int o0 = h3_stage_off(sl, h), o1 = h3_stage_off(sl, 2 + h);
                *reinterpret_cast<h4 *>(stY + o0) = h4{ stq.hi[0], stq.hi[1], stq.hi[2], stq.hi[3] };
                *reinterpret_cast<h4 *>(stY + H3_STAGE_PART + o0) = h4{ stq.lo[0], stq.lo[1], stq.lo[2], stq.lo[3] };
                *reinterpret_cast<h4 *>(stY + o1) = h4{ stq2.hi[0], stq2.hi[1], stq2.hi[2], stq2.hi[3] };
                *reinterpret_cast<h4 *>(stY + H3_STAGE_PART + o1) = h4{ stq2.lo[0], stq2.lo[1], stq2.lo[2], stq2.lo[3] };
            }
            __syncthreads();  // ---- A1
            {
                float dummy = 0.0f;
                if (wv == 0) wgrad_block<4, 0, 0>(gW_nar, dummy, stage, 0, L, 0, true, 1);        // heads: x = H[:32]
                else if (wv < 3) wgrad_block<4, 0, 0>(gW_nar, dummy, stage, 0, L, 0, false, wv - 1);  // rgb: x = c1 block wv-1
            }
            // dv1 = (W_rgb^T gs3) * G'(v1)
            HL2 dys[2];
            {
                v16f dc[2] = { zero16, zero16 };
                chain_narrow<2>(dc, lds, H3_D2, 4, L, narrgb);
                dys[0] = split16(mul16(dc[0], dv1f[0]));
                dys[1] = split16(mul16(dc[1], dv1f[1]));
            }
            __syncthreads();  // ---- B1
            // ================= Directional_MLP.mlp.2 (64 -> 64) =================
            h3_stage_put(stY, L, 0, dys[0]);
            h3_stage_put(stY, L, 1, dys[1]);
            v16f dv0f[2];
            {   // c0 = G(v0), G'(v0) recomputed from H[32:]
                v16f u[2] = { h3_ld16(lds, kLdsDinit + wv * 256 + (hf) * 64), h3_ld16(lds, kLdsDinit + wv * 256 + 128 + (hf) * 64) };
                const HL *const B[2] = { &Hs[1].t[0], &Hs[1].t[1] };
                h3_layer2<2>(u, lds, H3_D0, 3, L.fwd, B);
                act_and_deriv(u[0], dv0f[0]);
                act_and_deriv(u[1], dv0f[1]);
                h3_stage_put(stX, L, 0, split16(u[0]));
                h3_stage_put(stX, L, 1, split16(u[1]));
            }
            __syncthreads();  // ---- A2
            wgrad_block<4, 0, 4>(gW_D1, gB_D1, stage, 0, L, rb, false, cb, 4 * cb);
            {
                v16f dc[2] = { zero16, zero16 };
                chain64<2>(dc, lds, H3_D1, 4, L, dys);
                dys[0] = split16(mul16(dc[0], dv0f[0]));   // dv0
                dys[1] = split16(mul16(dc[1], dv0f[1]));
            }
            __syncthreads();  // ---- B2
            // ================= Directional_MLP.mlp.0 (32 of its 48 inputs; the SH part per ray) =================
            h3_stage_put(stY, L, 0, dys[0]);
            h3_stage_put(stY, L, 1, dys[1]);
            h3_stage_put(stX, L, 0, Hs[1]);
            __syncthreads();  // ---- A3
#pragma unroll
            for (int i = 0; i < 2; ++i) {   // own slots one at a time: their row sums meet different rays' SH
                float rsum = 0.0f;
                wgrad_block<1>(gW_D0H, rsum, stage, s2 + i, L, rb, false, 0);
                gB_D0 += rsum;
                rsum += __shfl_xor(rsum, 32, 64);  // all 32 samples of the tile, lane = unit 32rb + sl
                slot_rs[i] += rsum;
                const float4 *shp = reinterpret_cast<const float4 *>(shbuf + (s2 + i) * 16 + 8 * h);
                const float4 s0 = shp[0], s1 = shp[1];
                gW_D0S[0] = fmaf(rsum, s0.x, gW_D0S[0]); gW_D0S[1] = fmaf(rsum, s0.y, gW_D0S[1]);
                gW_D0S[2] = fmaf(rsum, s0.z, gW_D0S[2]); gW_D0S[3] = fmaf(rsum, s0.w, gW_D0S[3]);
                gW_D0S[4] = fmaf(rsum, s1.x, gW_D0S[4]); gW_D0S[5] = fmaf(rsum, s1.y, gW_D0S[5]);
                gW_D0S[6] = fmaf(rsum, s1.z, gW_D0S[6]); gW_D0S[7] = fmaf(rsum, s1.w, gW_D0S[7]);
            }
            v16f dH[2] = { zero16, zero16 };
            {
                v16f dc[1] = { zero16 };
                chain64<1>(dc, lds, H3_D0, 3, L, dys);
                dH[1] = dc[0];
                v16f dh0[1] = { zero16 };
                chain_narrow<1>(dh0, lds, H3_HEAD, 2, L, nar);
                dH[0] = dh0[0];
            }
            dys[0] = split16(dH[0]);
            dys[1] = split16(dH[1]);
            __syncthreads();  // ---- B3
            // ================= Spatial_MLP.mlp.2 (64 -> 64, linear) =================
            h3_stage_put(stY, L, 0, dys[0]);
            h3_stage_put(stY, L, 1, dys[1]);
            v16f du0f[2];
            {   // a0 = G(u0), G'(u0) recomputed from x
                v16f u[2] = { h3_bias(lds, 0, 0, hf), h3_bias(lds, 0, 1, hf) };
                const HL *const B[2] = { &xs2.t[0], &xs2.t[1] };
                h3_layer2<2>(u, lds, H3_L0, 2, L.fwd, B);
                act_and_deriv(u[0], du0f[0]);
                act_and_deriv(u[1], du0f[1]);
                h3_stage_put(stX, L, 0, split16(u[0]));
                h3_stage_put(stX, L, 1, split16(u[1]));
            }
            __syncthreads();  // ---- A4
            wgrad_block<4, 0, 4>(gW_L1, gB_L1, stage, 0, L, rb, false, cb, 4 * cb);
            {
                v16f dc[2] = { zero16, zero16 };
                chain64<2>(dc, lds, H3_L1, 4, L, dys);
                dys[0] = split16(mul16(dc[0], du0f[0]));   // du0
                dys[1] = split16(mul16(dc[1], du0f[1]));
            }
            __syncthreads();  // ---- B4
            // ================= Spatial_MLP.mlp.0 (32 -> 64) =================
            h3_stage_put(stY, L, 0, dys[0]);
            h3_stage_put(stY, L, 1, dys[1]);
            h3_stage_put(stX, L, 0, xs2);
            __syncthreads();  // ---- A5
            wgrad_block<2>(gW_L0, gB_L0, stage, s2, L, rb, false, 0);
            v16f dxa[1] = { zero16 };
            chain64<1>(dxa, lds, H3_L0, 2, L, dys);
            const v16f dx = dxa[0] * isc;
            __syncthreads();  // ---- B5: this tile's staging reads are complete

            asm volatile("" ::"v"(warm));
            // ================= feature gradients =================
            // register 2j+f of half h = level 4(j>>1)+2h+(j&1), feature f
            if (live && active && a.dfeat) {
                const size_t n = (size_t)ray * S + s, NS = (size_t)a.f.B * S;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int level = 4 * (j >> 1) + 2 * h + (j & 1);
                    reinterpret_cast<float2 *>(a.dfeat)[(size_t)level * NS + n] = make_float2(dx[2 * j], dx[2 * j + 1]);
                }
            }
            if (active && a.recs) {  // fused scatter producer (scatter.hip): records into the ranges the plan reserved
                // one level per trip of a ROLLED loop (unrolled, the eight levels' index arithmetic is live at once and
                // spills); the gradients are parked in this wave's staging image, free since barrier B5
                float4 *park = reinterpret_cast<float4 *>(stY + lane * 64);
                park[0] = make_float4(dx[0], dx[1], dx[2], dx[3]);
                park[1] = make_float4(dx[4], dx[5], dx[6], dx[7]);
                park[2] = make_float4(dx[8], dx[9], dx[10], dx[11]);
                park[3] = make_float4(dx[12], dx[13], dx[14], dx[15]);
                if (live) {
                    float pe[3];
                    contract_point(a.f, o, d, z, pe);
                    const uint32_t mask = (uint32_t)a.f.T - 1u;
                    // software-pipelined: the cursor round trips of level j+1 are in flight while level j's records are stored.
                    // Two register sets used alternately (a copy from "next" to "current" would wait for the atomics at the
                    // top of every trip); LDS operations return in order, so each trip reads its gradient BEFORE issuing
                    // the next level's atomics.
                    Pairs prA, prB;
                    PairSlots slA, slB;
                    auto lvl = [&](int j) { return 4 * (j >> 1) + 2 * h + (j & 1); };
                    // the level's resolution comes from LDS too: it is fetched one step ahead, before the atomics in flight
                    auto res_of = [&](int j) { return *reinterpret_cast<const int4 *>(lres + 4 * lvl(j)); };
                    auto issue = [&](int j, const int4 &r, Pairs &pr, PairSlots &sl) {
                        const int32_t rr[3] = { r.x, r.y, r.z };
                        make_pairs(pe, rr, mask, pr);
                        reserve_pairs(pr, cursor + lvl(j) * a.bins.NB, a.bins.bucket_log, sl);
                    };
                    auto commit = [&](int j, const Pairs &pr, const PairSlots &sl, float2 gxy) {
                        gmax = fmaxf(gmax, fmaxf(fabsf(gxy.x), fabsf(gxy.y)));
                        commit_pairs(pr, sl, gxy.x, gxy.y, cursor + lvl(j) * a.bins.NB, a.bins.bucket_log, a.bins.capacity, a.recs,
                                     a.grad_features + (size_t)lvl(j) * a.f.T * 2);
                    };
                    const float2 *gpark = reinterpret_cast<const float2 *>(stY + lane * 64);
                    int4 rA = res_of(0), rB = res_of(1);
                    issue(0, rA, prA, slA);
#pragma unroll 1
                    for (int jj = 0; jj < 4; ++jj) {
                        const float2 g0 = gpark[2 * jj];
                        rA = res_of(jj < 3 ? 2 * jj + 2 : 0);
                        __builtin_amdgcn_sched_barrier(0);
                        issue(2 * jj + 1, rB, prB, slB);
                        __builtin_amdgcn_sched_barrier(0);
                        commit(2 * jj, prA, slA, g0);
                        const float2 g1 = gpark[2 * jj + 1];
                        rB = res_of(jj < 3 ? 2 * jj + 3 : 1);
                        __builtin_amdgcn_sched_barrier(0);
                        if (jj < 3) issue(2 * jj + 2, rA, prA, slA);
                        __builtin_amdgcn_sched_barrier(0);
                        commit(2 * jj + 1, prB, slB, g1);
                    }
                }
            }
            SCANERF_STORE_GUARD();  // dfeat / record stores: their data registers are about to be reused by matrix results
        }
        // ---- per ray group: pose-gradient row sums of the own slots' rays
        if (a.g_rowsum && h == 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int rs_ray = 4 * grp + s2 + i;
                if (rs_ray < a.f.B && !(a.f.ray_valid && !a.f.ray_valid[rs_ray])) {
                    a.g_rowsum[((size_t)rs_ray * 2 + 0) * 64 + 32 * rb + sl] = slot_rs[i] * isc;
                    a.g_rowsum[((size_t)rs_ray * 2 + 1) * 64 + 32 * rb + sl] = 0.0f;
                }
            }
        }
    }

    if (a.recs) {  // launch-wide max |dL/dfeature| for the fixed-point scale of the accumulate pass
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) gmax = fmaxf(gmax, __shfl_xor(gmax, off, 64));
        if (lane == 0 && gmax > 0.0f) atomicMax(a.maxbits, __float_as_uint(gmax));
    }
    // ---- flush this wave's partial sums in blob order (dw_partial is zero-filled: only owned entries are written)
    float *out = a.dw_partial + (size_t)(blockIdx.x * 4 + wv) * SCANERF_PARAMSIZE;
    const int k = sl;
    auto put_w = [&](const v16f &acc, int base, int nb, int kb) {
#pragma unroll
        for (int g = 0; g < 16; ++g) out[base + 64 + (32 * kb + k) * 64 + 32 * nb + nmap(g, h)] = acc[g] * isc;
    };
    put_w(gW_D1, BLOB_D1, rb, cb);
    put_w(gW_L1, BLOB_S1, rb, cb);
    put_w(gW_L0, BLOB_S0, rb, 0);
    put_w(gW_D0H, BLOB_D0, rb, 0);
#pragma unroll
    for (int j = 0; j < 8; ++j) out[BLOB_D0 + 64 + (32 + 8 * h + j) * 64 + 32 * rb + k] = gW_D0S[j] * isc;
    {
        float v;
        v = (gB_D1 + __shfl_xor(gB_D1, 32, 64)) * isc;
        if (h == 0) out[BLOB_D1 + 32 * rb + k] = v;
        v = (gB_L1 + __shfl_xor(gB_L1, 32, 64)) * isc;
        if (h == 0) out[BLOB_S1 + 32 * rb + k] = v;
        v = (gB_D0 + __shfl_xor(gB_D0, 32, 64)) * isc;
        if (h == 0) out[BLOB_D0 + 32 * rb + k] = v;
        v = (gB_L0 + __shfl_xor(gB_L0, 32, 64)) * isc;
        if (h == 0) out[BLOB_S0 + 32 * rb + k] = v;
    }
    if (wv == 0) {  // heads: rows 0-3 (half 0, registers 0-3) sigma,dif; rows 4-6 (half 1, registers 0-2) tint; column = H unit k
        if (h == 0) {
            out[BLOB_SIG + 1 + k] = gW_nar[0] * isc;
#pragma unroll
            for (int c = 0; c < 3; ++c) out[BLOB_DIF + 3 + k * 3 + c] = gW_nar[1 + c] * isc;
        } else {
#pragma unroll
            for (int c = 0; c < 3; ++c) out[BLOB_TINT + 3 + k * 3 + c] = gW_nar[c] * isc;
        }
    } else if (wv < 3 && h == 0) {  // rgb layer: rows 8-10 = half 0, registers 4-6; column = unit 32(wv-1) + k
#pragma unroll
        for (int c = 0; c < 3; ++c) out[BLOB_D2 + 3 + (32 * (wv - 1) + k) * 3 + c] = gW_nar[4 + c] * isc;
    }
#pragma unroll
    for (int c = 0; c < 7; ++c) {
        const float b = half_sum(gB_head[c]) * isc;
        if (lane == 0) {
            if (c == 0) out[BLOB_SIG] = b;
            else if (c < 4) out[BLOB_DIF + c - 1] = b;
            else out[BLOB_TINT + c - 4] = b;
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float b = half_sum(gB_d2[c]) * isc;
        if (lane == 0) out[BLOB_D2 + c] = b;
    }
}

}  // namespace

namespace scanerf {

int launch_render_bwd_h3(const BwdArgs &a, int feat_dtype, int blocks, size_t lds_extra, hipStream_t st)
{
    const size_t lds_bytes = (size_t)kLdsCursor + lds_extra;
    SCANERF_REQUIRE(lds_bytes <= 160 * 1024, "render_backward(h3): %zu B of LDS needed (table too large for the fused scatter)", lds_bytes);
#define SCANERF_LAUNCH_BWD(DT)                                                                                     \
    {                                                                                                              \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_render_bwd_h3<DT>),                   \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);            \
        SCANERF_REQUIRE(e == hipSuccess, "render_backward(h3): cannot reserve %zu B of LDS: %s", lds_bytes,         \
                        hipGetErrorString(e));                                                                     \
        hipLaunchKernelGGL((k_render_bwd_h3<DT>), dim3(blocks), dim3(kThreads), lds_bytes, st, a);                  \
    }
    if (feat_dtype == SCANERF_F32) SCANERF_LAUNCH_BWD(SCANERF_F32)
    else if (feat_dtype == SCANERF_F16) SCANERF_LAUNCH_BWD(SCANERF_F16)
    else SCANERF_LAUNCH_BWD(SCANERF_BF16)
#undef SCANERF_LAUNCH_BWD
    return 0;
}

}  // namespace scanerf

// render_bwd.hip -- fused per-ray volume rendering, backward (gfx950).
//
// Adjoint of render.hip (hashgrid/__init__.py:512-596 under torch autograd in the reference):
// given dL/d(out_ray) it produces
//   * dL/d(hash features of every sample), level-major [16][B*S][2]  (fed to scatter.hip),
//   * dL/d(decoder blob) as per-wave partial sums (reduced by k_reduce_dw, deterministic),
// recomputing the forward per 32-sample tile instead of storing per-sample activations.
//
// One WORKGROUP per ray: its 4 waves take 4 consecutive 32-sample tiles (S=128: the whole ray).
// Three kinds of MFMA work per tile:
//   forward recompute      H^T   = W   X^T      A = weight image (LDS), B = activations (regs)
//   activation gradients   dX^T  = W^T dY^T     A = the SAME image walked transposed (chain16)
//   weight gradients       dW    = dY  X^T      reduction over SAMPLES: both operands need the
//                                               unit on the lane and samples across the steps --
//                                               the transpose of the register layout.
// The transposes go through an LDS stage buffer, and that buffer doubles as the exchange that
// lets the waves SPLIT the weight-gradient blocks: every wave writes its tile's (dY, X) rows,
// then accumulates only the blocks it owns over all 4 tiles.  A wave therefore holds 64
// accumulator registers instead of the 192 a one-wave-per-ray design needs (which spilled
// ~500 registers per lane: 47 GB of scratch traffic per launch, profiles/r01_fused_v1*).
//
// Compositing backward needs sum_{j>i} a_j w_j: tile totals are exchanged through LDS and the
// tile groups are walked LAST to FIRST with the tile-entry transmittances saved by the forward,
// so suffix sums are formed directly (a prefix-minus-total form divides rounding noise by
// f_i = 1-alpha_i+1e-6 and loses opaque samples).
#include <stdlib.h>

#include "render_bwd_common.h"

using namespace scanerf;

namespace {

constexpr int kBwdThreads = 256;
constexpr int kScrStride = 36;                   // floats per stage row (32 samples + pad: conflict-free b128 reads)
constexpr int kSlotRows = 128;                   // per wave: rows 0..63 dY (or private scratch), 64..127 X
constexpr int kSlotFloats = kSlotRows * kScrStride;
constexpr int kBwdLdsFloats = PK_TOTAL + 64 + 4 * kSlotFloats + 16;

// registers (lane = sample s, half h, reg g = unit nmap(g,h))  ->  rows[unit][sample]
__device__ __forceinline__ void rows_put(float *rows, int lane, int rowbase, const v16f &v)
{
    const int sl = lane & 31, h = lane >> 5;
#pragma unroll
    for (int g = 0; g < 16; ++g) rows[(rowbase + nmap(g, h)) * kScrStride + sl] = v[g];
}
// rows -> registers (lane = unit n, half h, reg t = sample 16h + t): operand layout of an MFMA
// whose reduction index is the sample
__device__ __forceinline__ v16f rows_get(const float *rows, int lane, int rowbase)
{
    const float4 *p = reinterpret_cast<const float4 *>(rows + (rowbase + (lane & 31)) * kScrStride + 16 * (lane >> 5));
    float4 a = p[0], b = p[1], c = p[2], d = p[3];
    v16f r = { a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w };
    return r;
}
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// acc[n][k] += sum_s dY[n][s] X[k][s]
__device__ __forceinline__ void mma_ws(v16f &acc, const v16f &dy, const v16f &x)
{
#pragma unroll
    for (int t = 0; t < 16; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(dy[t], x[t], acc, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ float sum16(const v16f &v)
{
    float s = 0.0f;
#pragma unroll
    for (int t = 0; t < 16; ++t) s += v[t];
    return s;
}
// VALU weight gradient of a narrow layer: acc[c] += sum_s g[c][s] x[s], g rows at `grow` (broadcast reads)
template <int NC>
__device__ __forceinline__ void narrow_wgrad(float *acc, const float *grow, int h, const v16f &xo)
{
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const float4 *gp = reinterpret_cast<const float4 *>(grow + c * kScrStride + 16 * h);
        float a = 0.0f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 gq = gp[q];
            a = fmaf(gq.x, xo[4 * q + 0], a);
            a = fmaf(gq.y, xo[4 * q + 1], a);
            a = fmaf(gq.z, xo[4 * q + 2], a);
            a = fmaf(gq.w, xo[4 * q + 3], a);
        }
        acc[c] += a;
    }
}

template <int DT>
__global__ void __launch_bounds__(kBwdThreads, 1) k_render_bwd(BwdArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    int *lres = reinterpret_cast<int *>(lds + PK_TOTAL);
    float *stage = lds + PK_TOTAL + 64;
    float *totals = stage + 4 * kSlotFloats;  // [4] tile totals of a_j w_j
    uint32_t *cursor = reinterpret_cast<uint32_t *>(totals + 16);  // [16*NB] record cursors (fused producer only)
    float gmax = 0.0f;
    if (a.recs) {
        const int nbins = 16 * a.bins.NB;
        for (int i = threadIdx.x; i < nbins; i += kBwdThreads)
            cursor[i] = a.bin_starts[i] + a.bin_rowprefix[(size_t)i * a.bins.W + blockIdx.x];
    }
    {
        const float4 *src = reinterpret_cast<const float4 *>(a.f.packed);
        float4 *dst = reinterpret_cast<float4 *>(lds);
        for (int i = threadIdx.x; i < PK_TOTAL / 4; i += kBwdThreads) dst[i] = src[i];
        if (threadIdx.x < 64) {
            int lv = threadIdx.x >> 2, c = threadIdx.x & 3;
            lres[threadIdx.x] = c < 3 ? a.f.resolutions[3 * lv + c] : 0;
        }
    }
    __syncthreads();
    const int lane_k = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int S = a.f.S, ntiles = (S + 31) >> 5, ngroups = (ntiles + 3) >> 2;
    // ownership of the weight-gradient blocks: 64x64 layers -> block (rb, cb) over all 4 tiles;
    // 64x32 layers -> row block rb2 over a PAIR of tiles
    const int rb = wv >> 1, cb = wv & 1;        // D1 / L1
    const int rb2 = wv & 1, tp = wv >> 1;       // D0H / L0: tiles 2*tp, 2*tp+1

    const v16f zero16 = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    v16f gW_D1 = zero16, gW_L1 = zero16, gW_D0H = zero16, gW_L0 = zero16;
    float gW_D0S[8], gW_head[7], gW_D2[2][3], gB_head[7], gB_d2[3];
    float gB_D1 = 0.0f, gB_L1 = 0.0f, gB_D0 = 0.0f, gB_L0 = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i) gW_D0S[i] = 0.0f;
#pragma unroll
    for (int i = 0; i < 7; ++i) gW_head[i] = gB_head[i] = 0.0f;
#pragma unroll
    for (int i = 0; i < 3; ++i) gW_D2[0][i] = gW_D2[1][i] = gB_d2[i] = 0.0f;

    for (int ray = blockIdx.x; ray < a.f.B; ray += gridDim.x) {
        if (a.f.ray_valid && !a.f.ray_valid[ray]) {  // block-uniform
            // invalid rays contribute nothing; their feature gradients are zero
            if (a.dfeat)
            for (int s = threadIdx.x >> 1; s < S; s += kBwdThreads / 2)
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    reinterpret_cast<float2 *>(a.dfeat)[(size_t)(2 * j + (threadIdx.x & 1)) * a.f.B * S + (size_t)ray * S + s] =
                        make_float2(0, 0);
            continue;
        }
        float o[3], d[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            o[k] = a.f.rays_o[3 * ray + k];
            d[k] = a.f.rays_d[3 * ray + k];
        }
        const float dnorm = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        const float *go = a.grad_out + (size_t)ray * SCANERF_RAY_OUT;  // dL/d(out_ray) of this ray
        const float *fo = a.f.out_ray + (size_t)ray * SCANERF_RAY_OUT;   // the forward outputs
        float Rcarry = 0.0f;  // sum of a_j w_j over all later tile groups
        float ray_rowsum = 0.0f;  // this wave's rows (32*rb2 + lane) of sum_s dv0, over its tile pair, all groups

        for (int grp = ngroups - 1; grp >= 0; --grp) {
            // Lane-derived LDS / global addresses are recomputed per tile group: hoisted to kernel entry
            // they are hundreds of lane-constant registers that end up spilled and reloaded from scratch.
            int lane = lane_k;
            asm volatile("" : "+v"(lane));
            const int sl = lane & 31, h = lane >> 5;
            float *slot = stage + wv * kSlotFloats;  // this wave's rows
            const int tile = 4 * grp + wv;
            const int s = tile * 32 + sl;
            const bool live = (tile < ntiles) && (s < S);
            const float z = live ? a.f.z_vals[(size_t)ray * S + s] : 0.0f;
            float delta = live ? a.f.dists[(size_t)ray * S + s] * dnorm : 0.0f;
            if (a.f.infinity && s == S - 1) delta = 1e10f;
            float p[3];
            contract_point(a.f, o, d, z, p);

            // ================= P0: forward recompute of this wave's tile =================
            v16f x;
            if (a.xstash) {
                // the forward saved this lane's 16 encoder outputs: 64 contiguous bytes per lane
                const float4 *xs = reinterpret_cast<const float4 *>(a.xstash + ((size_t)ray * S + (live ? s : 0)) * 32 + 16 * h);
                const float4 q0 = xs[0], q1 = xs[1], q2 = xs[2], q3 = xs[3];
                x = v16f{ q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w };
            } else {
                encode8<DT>(a.f, lres, h, p, x);
            }
            v16f H[2] = { load_bias(lds, 1, 0, h), load_bias(lds, 1, 1, h) };
            {
                v16f u0a = load_bias(lds, 0, 0, h), u0b = load_bias(lds, 0, 1, h);
                mma_block16(u0a, lds + PK_L0, 0, lane, x);
                mma_block16(u0b, lds + PK_L0, 4, lane, x);
                v16f a0 = act16(u0a), a1 = act16(u0b);
                mma_block16(H[0], lds + PK_L1, 0, lane, a0);
                mma_block16(H[0], lds + PK_L1, 4, lane, a1);
                mma_block16(H[1], lds + PK_L1, 8, lane, a0);
                mma_block16(H[1], lds + PK_L1, 12, lane, a1);
            }
            float hd[7] = { 0, 0, 0, 0, 0, 0, 0 };
            {
                const float4 *W = reinterpret_cast<const float4 *>(lds + PK_HEAD + h * 128);
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    float4 wa = W[2 * g], wb = W[2 * g + 1];
                    float v = H[0][g];
                    hd[0] = fmaf(v, wa.x, hd[0]); hd[1] = fmaf(v, wa.y, hd[1]); hd[2] = fmaf(v, wa.z, hd[2]);
                    hd[3] = fmaf(v, wa.w, hd[3]); hd[4] = fmaf(v, wb.x, hd[4]); hd[5] = fmaf(v, wb.y, hd[5]);
                    hd[6] = fmaf(v, wb.z, hd[6]);
                }
                const float *hb = lds + PK_HB;
#pragma unroll
                for (int c = 0; c < 7; ++c) hd[c] = hd[c] + __shfl_xor(hd[c], 32, 64) + hb[c];
            }
            const float sigma = softplus_(hd[0]);
            const float dsig_dpre = hd[0] > 20.0f ? 1.0f : sigmoid_(hd[0]);
            float dif[3], tint[3], spec[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                dif[c] = sigmoid_(hd[1 + c]);
                tint[c] = sigmoid_(hd[4 + c]);
            }
            v16f v0[2] = { load_bias(lds, 2, 0, h), load_bias(lds, 2, 1, h) };
            {   // SH part of the 48 inputs
                float shl[16];
                ray_sh(d, dnorm, shl);
                float shb[8];  // MFMA B operand: step r, half h -> SH[2r+h]
#pragma unroll
                for (int r = 0; r < 8; ++r) shb[r] = h ? shl[2 * r + 1] : shl[2 * r];
                const float *A = lds + PK_D0S + (lane + (lane >> 5)) * 4;
                const float4 a00 = *reinterpret_cast<const float4 *>(A), a01 = *reinterpret_cast<const float4 *>(A + PK_GRP),
                             a10 = *reinterpret_cast<const float4 *>(A + 2 * PK_GRP),
                             a11 = *reinterpret_cast<const float4 *>(A + 3 * PK_GRP);
                MFMA4(v0[0], a00, shb[0], shb[1], shb[2], shb[3])
                MFMA4(v0[0], a01, shb[4], shb[5], shb[6], shb[7])
                MFMA4(v0[1], a10, shb[0], shb[1], shb[2], shb[3])
                MFMA4(v0[1], a11, shb[4], shb[5], shb[6], shb[7])
                __builtin_amdgcn_sched_barrier(0);
            }
            mma_block16(v0[0], lds + PK_D0H, 0, lane, H[1]);
            mma_block16(v0[1], lds + PK_D0H, 4, lane, H[1]);
            v16f v1[2] = { load_bias(lds, 3, 0, h), load_bias(lds, 3, 1, h) };
            {
                v16f c0 = act16(v0[0]), c1 = act16(v0[1]);
                mma_block16(v1[0], lds + PK_D1, 0, lane, c0);
                mma_block16(v1[0], lds + PK_D1, 4, lane, c1);
                mma_block16(v1[1], lds + PK_D1, 8, lane, c0);
                mma_block16(v1[1], lds + PK_D1, 12, lane, c1);
            }
            {
                float c3[3] = { 0, 0, 0 };
                const float4 *W = reinterpret_cast<const float4 *>(lds + PK_D2 + h * 128);
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    float4 w0 = W[g], w1 = W[16 + g];
                    const float e0 = gauss_act(v1[0][g]), e1 = gauss_act(v1[1][g]);
                    c3[0] = fmaf(e0, w0.x, c3[0]); c3[1] = fmaf(e0, w0.y, c3[1]); c3[2] = fmaf(e0, w0.z, c3[2]);
                    c3[0] = fmaf(e1, w1.x, c3[0]); c3[1] = fmaf(e1, w1.y, c3[1]); c3[2] = fmaf(e1, w1.z, c3[2]);
                }
                const float *hb = lds + PK_HB + 8;
#pragma unroll
                for (int c = 0; c < 3; ++c) spec[c] = sigmoid_(c3[c] + __shfl_xor(c3[c], 32, 64) + hb[c]);
            }
            // compositing recompute (identical arithmetic to the forward)
            const float ex = live ? expf(-sigma * delta) : 1.0f;  // 1 - alpha
            const float alpha = 1.0f - ex;
            const float fi = 1.0f - alpha + 1e-6f;
            float incl = fi;
#pragma unroll
            for (int off = 1; off < 32; off <<= 1) {
                float t = __shfl_up(incl, off, 32);
                if (sl >= off) incl *= t;
            }
            float excl = __shfl_up(incl, 1, 32);
            if (sl == 0) excl = 1.0f;
            const float Ti = (tile < ntiles ? a.tile_T[(size_t)ray * ((a.f.S + 15) >> 4) + 2 * tile] : 0.0f) * excl;
            const float w = alpha * Ti;
            // upstream gradients (uniform per ray; read through the scalar cache)
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
                float t = __shfl_down(rs, off, 32);
                if (sl + off < 32) rs += t;
            }
            if (lane == 0) totals[wv] = rs;  // lane 0: the whole tile
            __syncthreads();  // ---- B1: tile totals visible; every wave is done with last group's stage rows

            // ================= P1: compositing adjoint, narrow layers, D1 operands =================
            const float t0_ = totals[0], t1_ = totals[1], t2_ = totals[2], t3_ = totals[3];
            const float later = (wv < 1 ? t1_ : 0.0f) + (wv < 2 ? t2_ : 0.0f) + (wv < 3 ? t3_ : 0.0f);
            const float suffix = Rcarry + later + rs - aw;
            Rcarry += t0_ + t1_ + t2_ + t3_;
            float dalpha = Ti * ai - (suffix + ((s < S - 1) ? gTl * Tl : 0.0f)) / fi;
            if (!live) dalpha = 0.0f;
            const float dsigma = dalpha * delta * ex;
            if (a.g_dnorm && tile < ntiles) {
                // d(delta)/d|d| = dist (the infinity sample's delta is the constant 1e10)
                const float dist_i = (live && !(a.f.infinity && s == S - 1)) ? a.f.dists[(size_t)ray * S + s] : 0.0f;
                const float gd = half_sum(dalpha * sigma * ex * dist_i);
                if (lane == 0) a.g_dnorm[(size_t)ray * ntiles + tile] = gd;
            }
            float gh[7], gs3[3];  // gradients w.r.t. the head pre-activations
            gh[0] = dsigma * dsig_dpre;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                gh[1 + c] = w * gD[c] * dif[c] * (1.0f - dif[c]);
                gh[4 + c] = w * (gS[c] * spec[c] + gTi[c]) * tint[c] * (1.0f - tint[c]);
                gs3[c] = (w * gS[c] * tint[c] + gW2 * w * 2.0f * spec[c]) * spec[c] * (1.0f - spec[c]);
            }
#pragma unroll
            for (int c = 0; c < 7; ++c) gB_head[c] += gh[c];
#pragma unroll
            for (int c = 0; c < 3; ++c) gB_d2[c] += gs3[c];

            // Directional_MLP.mlp.4 (64 -> 3): dv1 = (W^T g) * G'(v1)
            v16f dv1[2];
            {
                const float4 *W = reinterpret_cast<const float4 *>(lds + PK_D2 + h * 128);
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    float4 w0 = W[g], w1 = W[16 + g];
                    const float dc0 = w0.x * gs3[0] + w0.y * gs3[1] + w0.z * gs3[2];
                    const float dc1 = w1.x * gs3[0] + w1.y * gs3[1] + w1.z * gs3[2];
                    const float q0 = opaque1(v1[0][g]), q1 = opaque1(v1[1][g]);
                    dv1[0][g] = dc0 * dgauss(q0, gauss_act(q0));
                    dv1[1][g] = dc1 * dgauss(q1, gauss_act(q1));
                }
            }
            // narrow-layer weight gradients on the VALU, own rows as private scratch:
            // rows 0..9 = head / rgb pre-activation gradients [c][sample]; rows 64.. = the layer inputs
            if (h == 0) {
#pragma unroll
                for (int c = 0; c < 7; ++c) slot[c * kScrStride + sl] = gh[c];
#pragma unroll
                for (int c = 0; c < 3; ++c) slot[(7 + c) * kScrStride + sl] = gs3[c];
            }
            {
                v16f c1a = act16(opaque16(v1[0])), c1b = act16(opaque16(v1[1]));
                rows_put(slot, lane, 64, c1a);
                rows_put(slot, lane, 96, c1b);
                wave_lds_fence();
                v16f xo = rows_get(slot, lane, 64);
                narrow_wgrad<3>(gW_D2[0], slot + 7 * kScrStride, h, xo);
                xo = rows_get(slot, lane, 96);
                narrow_wgrad<3>(gW_D2[1], slot + 7 * kScrStride, h, xo);
                wave_lds_fence();
                rows_put(slot, lane, 64, H[0]);
                wave_lds_fence();
                xo = rows_get(slot, lane, 64);
                narrow_wgrad<7>(gW_head, slot, h, xo);
                wave_lds_fence();
            }
            // stage D1: dY = dv1, X = c0 = G(v0)
            rows_put(slot, lane, 0, dv1[0]);
            rows_put(slot, lane, 32, dv1[1]);
            {
                v16f c0a = act16(opaque16(v0[0])), c0b = act16(opaque16(v0[1]));
                rows_put(slot, lane, 64, c0a);
                rows_put(slot, lane, 96, c0b);
            }
            __syncthreads();  // ---- B2
            // ================= P2: owned D1 block; chain to dv0 =================
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float *ts = stage + t * kSlotFloats;
                v16f dy = rows_get(ts, lane, 32 * rb), xx = rows_get(ts, lane, 64 + 32 * cb);
                if (cb == 0) gB_D1 += sum16(dy);
                mma_ws(gW_D1, dy, xx);
            }
            v16f dv0[2] = { zero16, zero16 };
            chain16(dv0[0], lds + PK_D1, 8, lane, 0, 0, dv1[0]);
            chain16(dv0[0], lds + PK_D1, 8, lane, 0, 1, dv1[1]);
            chain16(dv0[1], lds + PK_D1, 8, lane, 1, 0, dv1[0]);
            chain16(dv0[1], lds + PK_D1, 8, lane, 1, 1, dv1[1]);
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const float q0 = opaque1(v0[0][g]), q1 = opaque1(v0[1][g]);
                dv0[0][g] *= dgauss(q0, gauss_act(q0));
                dv0[1][g] *= dgauss(q1, gauss_act(q1));
            }
            __syncthreads();  // ---- B3: D1 operands consumed
            // stage D0: dY = dv0, X = H[32:64]
            rows_put(slot, lane, 0, dv0[0]);
            rows_put(slot, lane, 32, dv0[1]);
            rows_put(slot, lane, 64, H[1]);
            __syncthreads();  // ---- B4
            // ================= P4: owned D0H half-block (+ SH part, bias); chain to dH =================
            {
                float rsum = 0.0f;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const float *ts = stage + (2 * tp + t) * kSlotFloats;
                    v16f dy = rows_get(ts, lane, 32 * rb2), xx = rows_get(ts, lane, 64);
                    rsum += sum16(dy);
                    mma_ws(gW_D0H, dy, xx);
                }
                gB_D0 += rsum;
                rsum += __shfl_xor(rsum, 32, 64);  // both halves of the samples
                ray_rowsum += rsum;
                float shl[16];
                ray_sh(d, opaque1(dnorm), shl);
#pragma unroll
                for (int j = 0; j < 8; ++j) gW_D0S[j] = fmaf(rsum, h ? shl[8 + j] : shl[j], gW_D0S[j]);
            }
            v16f dH[2] = { zero16, zero16 };
            chain16(dH[1], lds + PK_D0H, 4, lane, 0, 0, dv0[0]);
            chain16(dH[1], lds + PK_D0H, 4, lane, 0, 1, dv0[1]);
            {   // heads (32 -> 1+3+3): dH[:32] = W^T g
                const float4 *W = reinterpret_cast<const float4 *>(lds + PK_HEAD + h * 128);
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    float4 wa = W[2 * g], wb = W[2 * g + 1];
                    dH[0][g] = wa.x * gh[0] + wa.y * gh[1] + wa.z * gh[2] + wa.w * gh[3] + wb.x * gh[4] + wb.y * gh[5] +
                               wb.z * gh[6];
                }
            }
            // u0 recomputed (32 MFMAs) instead of held across the directional stage
            v16f u0[2] = { load_bias(lds, 0, 0, h), load_bias(lds, 0, 1, h) };
            mma_block16(u0[0], lds + PK_L0, 0, lane, x);
            mma_block16(u0[1], lds + PK_L0, 4, lane, x);
            __syncthreads();  // ---- B5: D0 operands consumed
            // stage L1: dY = dH, X = a0 = G(u0)
            rows_put(slot, lane, 0, dH[0]);
            rows_put(slot, lane, 32, dH[1]);
            {
                v16f a0 = act16(opaque16(u0[0])), a1 = act16(opaque16(u0[1]));
                rows_put(slot, lane, 64, a0);
                rows_put(slot, lane, 96, a1);
            }
            __syncthreads();  // ---- B6
            // ================= P6: owned L1 block; chain to du0 =================
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float *ts = stage + t * kSlotFloats;
                v16f dy = rows_get(ts, lane, 32 * rb), xx = rows_get(ts, lane, 64 + 32 * cb);
                if (cb == 0) gB_L1 += sum16(dy);
                mma_ws(gW_L1, dy, xx);
            }
            v16f du0[2] = { zero16, zero16 };
            chain16(du0[0], lds + PK_L1, 8, lane, 0, 0, dH[0]);
            chain16(du0[0], lds + PK_L1, 8, lane, 0, 1, dH[1]);
            chain16(du0[1], lds + PK_L1, 8, lane, 1, 0, dH[0]);
            chain16(du0[1], lds + PK_L1, 8, lane, 1, 1, dH[1]);
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const float q0 = opaque1(u0[0][g]), q1 = opaque1(u0[1][g]);
                du0[0][g] *= dgauss(q0, gauss_act(q0));
                du0[1][g] *= dgauss(q1, gauss_act(q1));
            }
            __syncthreads();  // ---- B7: L1 operands consumed
            // stage L0: dY = du0, X = x (32 input rows)
            rows_put(slot, lane, 0, du0[0]);
            rows_put(slot, lane, 32, du0[1]);
            rows_put(slot, lane, 64, x);
            __syncthreads();  // ---- B8
            // ================= P8: owned L0 half-block; dx =================
            {
                float rsum = 0.0f;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const float *ts = stage + (2 * tp + t) * kSlotFloats;
                    v16f dy = rows_get(ts, lane, 32 * rb2), xx = rows_get(ts, lane, 64);
                    rsum += sum16(dy);
                    mma_ws(gW_L0, dy, xx);
                }
                gB_L0 += rsum;
            }
            v16f dx = zero16;
            chain16(dx, lds + PK_L0, 4, lane, 0, 0, du0[0]);
            chain16(dx, lds + PK_L0, 4, lane, 0, 1, du0[1]);
            // feature gradients, level-major (register 2j+f of half h = level 4(j>>1)+2h+(j&1))
            if (live && a.dfeat) {
                const size_t n = (size_t)ray * S + s, NS = (size_t)a.f.B * S;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int level = 4 * (j >> 1) + 2 * h + (j & 1);
                    reinterpret_cast<float2 *>(a.dfeat)[(size_t)level * NS + n] = make_float2(dx[2 * j], dx[2 * j + 1]);
                }
            }
            // fused scatter producer: this lane's 8 levels x 4 corner pairs -> 16-byte records appended to
            // the ranges k_bin_count_rays reserved for this workgroup (same point, same pairs)
            if (live && a.recs) {
                float pe[3];
                contract_point(a.f, o, d, opaque1(z), pe);
                const uint32_t mask = (uint32_t)a.f.T - 1u;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int level = 4 * (j >> 1) + 2 * h + (j & 1);
                    gmax = fmaxf(gmax, fmaxf(fabsf(dx[2 * j]), fabsf(dx[2 * j + 1])));
                    Pairs pr;
                    make_pairs(pe, lres + 4 * level, mask, pr);
                    emit_pairs(pr, dx[2 * j], dx[2 * j + 1], cursor + level * a.bins.NB, a.bins.bucket_log,
                               a.bins.capacity, a.recs, a.grad_features + (size_t)level * a.f.T * 2);
                }
            }
            // (the next group's B1 orders this group's stage reads before the next writes)
        }
        if (a.g_rowsum && (lane_k >> 5) == 0) a.g_rowsum[((size_t)ray * 2 + tp) * 64 + 32 * rb2 + (lane_k & 31)] = ray_rowsum;
    }

    if (a.recs) {  // launch-wide max |dL/dfeature| for the fixed-point scale of the accumulate pass
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) gmax = fmaxf(gmax, __shfl_xor(gmax, off, 64));
        if (lane_k == 0 && gmax > 0.0f) atomicMax(a.maxbits, __float_as_uint(gmax));
    }
    // ---- flush this wave's partial sums in blob order (dw_partial is zero-filled: only owned entries are written)
    float *out = a.dw_partial + (size_t)(blockIdx.x * 4 + wv) * SCANERF_PARAMSIZE;
    const int lane = lane_k, h = lane >> 5;
    const int k = lane & 31;
    auto put_w = [&](const v16f &acc, int base, int rbk, int cbk) {
#pragma unroll
        for (int g = 0; g < 16; ++g) out[base + 64 + (32 * cbk + k) * 64 + 32 * rbk + nmap(g, h)] = acc[g];
    };
    put_w(gW_D1, BLOB_D1, rb, cb);
    put_w(gW_L1, BLOB_S1, rb, cb);
    put_w(gW_D0H, BLOB_D0, rb2, 0);
    put_w(gW_L0, BLOB_S0, rb2, 0);
#pragma unroll
    for (int j = 0; j < 8; ++j) out[BLOB_D0 + 64 + (32 + 8 * h + j) * 64 + 32 * rb2 + k] = gW_D0S[j];
    {
        float v;
        v = gB_D1 + __shfl_xor(gB_D1, 32, 64);
        if (h == 0 && cb == 0) out[BLOB_D1 + 32 * rb + k] = v;
        v = gB_L1 + __shfl_xor(gB_L1, 32, 64);
        if (h == 0 && cb == 0) out[BLOB_S1 + 32 * rb + k] = v;
        v = gB_D0 + __shfl_xor(gB_D0, 32, 64);
        if (h == 0) out[BLOB_D0 + 32 * rb2 + k] = v;
        v = gB_L0 + __shfl_xor(gB_L0, 32, 64);
        if (h == 0) out[BLOB_S0 + 32 * rb2 + k] = v;
    }
#pragma unroll
    for (int c = 0; c < 7; ++c) {
        float v = gW_head[c] + __shfl_xor(gW_head[c], 32, 64);
        if (h == 0) {
            if (c == 0) out[BLOB_SIG + 1 + k] = v;
            else if (c < 4) out[BLOB_DIF + 3 + k * 3 + (c - 1)] = v;
            else out[BLOB_TINT + 3 + k * 3 + (c - 4)] = v;
        }
        float b = half_sum(gB_head[c]);
        if (lane == 0) {
            if (c == 0) out[BLOB_SIG] = b;
            else if (c < 4) out[BLOB_DIF + c - 1] = b;
            else out[BLOB_TINT + c - 4] = b;
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            float v = gW_D2[b][c] + __shfl_xor(gW_D2[b][c], 32, 64);
            if (h == 0) out[BLOB_D2 + 3 + (32 * b + k) * 3 + c] = v;
        }
        float bsum = half_sum(gB_d2[c]);
        if (lane == 0) out[BLOB_D2 + c] = bsum;
    }
}

// grad_blob[e] += sum over waves of the partials; first-layer weights carry the folded weight_feature.
// 64 columns x 16 row chunks per workgroup: every thread sums a fixed subset of the rows, the 16 partial sums are
// added in a fixed order -- deterministic, and enough workgroups (219) to stream the 57 MB of partials.
__global__ void __launch_bounds__(1024) k_reduce_dw(const float *__restrict__ partial, int nwaves,
                                                    const float *__restrict__ wf, float *__restrict__ grad_blob)
{
    __shared__ float part[16][64];
    const int c = threadIdx.x & 63, r = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + c;
    float s = 0.0f;
    if (e < SCANERF_PARAMSIZE)
        for (int w = r; w < nwaves; w += 16) s += partial[(size_t)w * SCANERF_PARAMSIZE + e];
    part[r][c] = s;
    __syncthreads();
    if (r == 0 && e < SCANERF_PARAMSIZE) {
        float t = 0.0f;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += part[q][c];
        if (e >= BLOB_S0 + 64 && e < BLOB_S1) t *= wf[(e - 64) / 64];
        grad_blob[e] += t;
    }
}

}  // namespace

namespace scanerf {
int launch_render_bwd_f32(const BwdArgs &a, int feat_dtype, int blocks, size_t lds_extra, hipStream_t st)
{
    const size_t lds_bytes = (size_t)kBwdLdsFloats * sizeof(float) + lds_extra;
#define SCANERF_LAUNCH_BWD(DT)                                                                                     \
    {                                                                                                              \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_render_bwd<DT>),                      \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);            \
        SCANERF_REQUIRE(e == hipSuccess, "render_backward: cannot reserve %zu B of LDS: %s", lds_bytes,             \
                        hipGetErrorString(e));                                                                     \
        hipLaunchKernelGGL((k_render_bwd<DT>), dim3(blocks), dim3(kBwdThreads), lds_bytes, st, a);                  \
    }
    if (feat_dtype == SCANERF_F32) SCANERF_LAUNCH_BWD(SCANERF_F32)
    else if (feat_dtype == SCANERF_F16) SCANERF_LAUNCH_BWD(SCANERF_F16)
    else SCANERF_LAUNCH_BWD(SCANERF_BF16)
#undef SCANERF_LAUNCH_BWD
    return 0;
}
}  // namespace scanerf

// ---------------------------------------------------------------------------- C ABI
// (SCANERF_BWD_GRID / SCANERF_FWD_GRID cap the persistent grids of the fused backward / forward: a tuning and experiment knob --
// e.g. both kernels resident at once on disjoint CUs, tools/overlap_probe.py; the scatter plan follows this function)
SCANERF_API int scanerf_render_backward_grid(int B)
{
    int cap = kNumCU;
    { const int v = tune_int("SCANERF_BWD_GRID", 0); if (v >= 1 && v < cap) cap = v; }
    return B > cap ? cap : (B < 1 ? 1 : B);
}

// dw_partial: [4 * scanerf_render_backward_grid(B)][13994] f32 scratch; grad_blob [13994] is accumulated into.
SCANERF_API int scanerf_render_backward(const float *rays_o, const float *rays_d, const float *z_vals, const float *dists,
                                        const void *features, int feat_dtype, const int32_t *resolutions,
                                        const float *workspace, const float *weight_feature,
                                        const scanerf_render_cfg *cfg, const uint8_t *ray_valid, const float *out_ray,
                                        const float *tile_T, const float *grad_out, const float *xstash, float *dfeat,
                                        float *dw_partial, float *grad_blob, float *g_dnorm, float *g_rowsum,
                                        const void *jstash, float *g_raypos,
                                        void *scatter_ws, size_t scatter_ws_bytes, float *grad_features, int B, int S,
                                        int T, scanerf_stream_t stream)
{
    SCANERF_REQUIRE(B >= 0 && S >= 1, "render_backward: B=%d S=%d", B, S);
    SCANERF_REQUIRE(T >= 2 && (T & (T - 1)) == 0, "render_backward: T=%d must be a power of two", T);
    SCANERF_REQUIRE(feat_dtype >= 0 && feat_dtype <= 2, "render_backward: feat_dtype=%d", feat_dtype);
    SCANERF_REQUIRE(cfg, "render_backward: cfg is null");
    if (B == 0) return 0;
    SCANERF_REQUIRE(rays_o && rays_d && z_vals && dists && features && resolutions && workspace && weight_feature &&
                        out_ray && tile_T && grad_out && dw_partial && grad_blob,
                    "render_backward: null pointer");
    SCANERF_REQUIRE(dfeat || scatter_ws, "render_backward: need dfeat or a planned scatter workspace");
    BwdArgs a;
    a.f.rays_o = rays_o; a.f.rays_d = rays_d; a.f.z_vals = z_vals; a.f.dists = dists;
    a.f.features = features; a.f.resolutions = resolutions; a.f.packed = workspace; a.f.ray_valid = ray_valid;
    a.f.out_ray = const_cast<float *>(out_ray); a.f.weights = nullptr; a.f.tile_T = nullptr; a.f.xstash = nullptr;
    a.f.skip_levels = 0;  // (the re-gathering kernels read every level; which levels get records is the plan's word in the workspace)
    a.f.B = B; a.f.S = S; a.f.T = T;
    a.f.contract_mode = cfg->contract_mode; a.f.infinity = cfg->infinity;
    for (int k = 0; k < 3; ++k) {
        a.f.min_bbox[k] = cfg->min_bbox[k];
        a.f.bbox_size[k] = cfg->bbox_size[k];
        a.f.inv_size4[k] = 4.0f / cfg->bbox_size[k];
    }
    a.f.dbg = tune_int("SCANERF_DEBUG_BWD", 0);  // timing experiments only (-DSCANERF_BWD_EXPERIMENTS builds)
    a.grad_out = grad_out; a.tile_T = tile_T; a.dfeat = dfeat; a.dw_partial = dw_partial; a.xstash = xstash;
    a.g_dnorm = g_dnorm; a.g_rowsum = g_rowsum; a.g_raypos = g_raypos;
    a.f.jstash = static_cast<uint32_t *>(const_cast<void *>(jstash));
    SCANERF_REQUIRE(cfg->arith >= SCANERF_ARITH_F32 && cfg->arith <= SCANERF_ARITH_T16S, "render_backward: arith=%d", cfg->arith);
    const bool h3 = cfg->arith == SCANERF_ARITH_H3, t16s = cfg->arith == SCANERF_ARITH_T16S, t16 = cfg->arith == SCANERF_ARITH_T16 || t16s;
    SCANERF_REQUIRE(!t16 || xstash, "render_backward: arith T16 / T16S needs the forward's x-stash (use SCANERF_ARITH_H3 to re-gather)");
    SCANERF_REQUIRE(!g_raypos || t16, "render_backward: g_raypos is produced by the t16 kernel only");
    const int blocks = scanerf_render_backward_grid(B);
    size_t lds_extra = 0;
    a.recs = nullptr;
    if (scatter_ws) {  // planned by scanerf_render_scatter_plan on the same (B, S, T), cfg and inputs
        SCANERF_REQUIRE(grad_features, "render_backward: grad_features is required with a scatter workspace");
        SCANERF_REQUIRE(scanerf_render_scatter_workspace_bytes(B, S, T) != 0,
                        "render_backward: fused scatter does not support B=%d S=%d T=%d", B, S, T);
        a.bins.bucket_log = fused_bucket_log(T);
        a.bins.N = B * S; a.bins.L = 16; a.bins.T = T;
        a.bins.NB = T >> a.bins.bucket_log;
        a.bins.W = blocks;
        a.bins.per_wg = 0;
        a.bins.rpg = t16 ? 8 : (h3 ? 4 : 1);
        a.bins.rec8 = fused_rec8(cfg->arith, a.bins.bucket_log);  // as the plan decided (scatter.hip: fused_geom)
        BinWorkspace w;
        SCANERF_REQUIRE(bin_workspace_carve(scatter_ws, scatter_ws_bytes, 16 * a.bins.NB, blocks, w),
                        "render_backward: scatter workspace too small (%zu B)", scatter_ws_bytes);
        a.bins.capacity = fused_coarse_capacity(w.capacity, B, S, a.bins.bucket_log);   // (never into a large table's fine area)
        a.bin_rowprefix = w.counts; a.bin_starts = w.starts; a.recs = w.recs; a.maxbits = w.maxbits;
        a.grad_features = grad_features;
        lds_extra = (size_t)16 * a.bins.NB * sizeof(uint32_t);
    }
    hipStream_t st = (hipStream_t)stream;
    // partial rows: one per wave (f32 / h3 kernels), one per workgroup (t16: every entry has one owner in the workgroup)
    const int prows = t16 ? blocks : blocks * 4;
    hipError_t me = hipMemsetAsync(dw_partial, 0, (size_t)prows * SCANERF_PARAMSIZE * sizeof(float), st);
    SCANERF_REQUIRE(me == hipSuccess, "render_backward: memset failed: %s", hipGetErrorString(me));
    if (int e = t16 ? launch_render_bwd_t16(a, feat_dtype, blocks, lds_extra, st, t16s)
              : h3 ? launch_render_bwd_h3(a, feat_dtype, blocks, lds_extra, st)
                   : launch_render_bwd_f32(a, feat_dtype, blocks, lds_extra, st))
        return e;
    if (int e = check_launch("render_backward")) return e;
    hipLaunchKernelGGL(k_reduce_dw, dim3(ceil_div(SCANERF_PARAMSIZE, 64)), dim3(1024), 0, st, dw_partial, prows,
                       weight_feature, grad_blob);
    return check_launch("render_backward(reduce)");
}

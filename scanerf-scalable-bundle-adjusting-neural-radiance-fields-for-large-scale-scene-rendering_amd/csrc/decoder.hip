// decoder.hip -- the sigma / colour decoder as a STAND-ALONE op on the matrix cores (gfx950), forward and backward.
//
// Reference: network.ShallowMLP.forward (network.py:172-190) under torch autograd -- what an unchanged
// HashGrid.render_batch_rays (hashgrid/__init__.py:545-548) calls between the hash encoder and the compositing:
//     x [N, 32 features + 3 view direction], weight_feature [32]  ->  sigma [N,1], diffuse [N,3], specular [N,3], tint [N,3]
// In the fused kernels (render.hip, render_bwd_t16.hip) this network only exists between an encoder and a compositing stage;
// a caller that keeps the reference's own op-by-op structure got torch's graph instead (~40 kernels over [N,64] intermediates:
// 140 of the 155 ms of such a step at 65 536 x 128 samples).  Here the same arithmetic as the fused kernels' -- split-f16
// operands, three products per term, f32 accumulate: f32-equivalent (render_h3.h) -- runs per SAMPLE:
//   forward   k_decoder_fwd_h3:  32-sample tiles on v_mfma_f32_32x32x16_f16, the h3 image of render_h3.h, 8 waves per
//             workgroup, two waves per SIMD (k_decoder_fwd_s16, 16-sample tiles at four waves per SIMD, for comparison);
//             per-sample view directions (the SH part of Directional_MLP.mlp.0 is one more k-step per tile instead of a
//             per-ray constant);
//   backward  k_decoder_bwd_s16: the t16s structure of render_bwd_t16.hip (16-sample tiles on v_mfma_f32_16x16x32_f16,
//             8 waves = two per SIMD, weight-gradient blocks owned by waves and summed over the 8 waves' tiles through LDS
//             staging, the workgroup's power-of-two gradient scale) with the compositing adjoint replaced by the incoming
//             per-sample gradients and the record emission by a plain [N,32] store; optionally dL/d(view direction) through
//             the degree-3 harmonics and the normalisation.
// Inputs and outputs are addressed by (pointer, row stride in floats), so the concatenated x [N,35] and its gradient are
// read / written in place (no slicing copies).
#include <stdlib.h>

#include "render_h3.h"
#include "render_t16.h"

using namespace scanerf;

namespace {

struct DecArgs {
    const float *feats;       // [N] rows of 32 floats, row stride ld_feats
    const float *dirs;        // [N] rows of 3 floats, row stride ld_dirs
    const float *packed;      // scanerf_pack_decoder's workspace
    int ld_feats, ld_dirs;
    long long N;
    // forward outputs
    float *sigma, *dif, *spec, *tint;   // [N], [N,3], [N,3], [N,3] contiguous
    // backward: incoming gradients (any may be null = zero) and outputs
    const float *g_sigma, *g_dif, *g_spec, *g_tint;
    float *d_feats;           // rows of 32 floats, row stride ld_dfeats
    float *d_dirs;            // rows of 3 floats, row stride ld_ddirs; may be null
    int ld_dfeats, ld_ddirs;
    float *dw_partial;        // [grid][SCANERF_PARAMSIZE], zero-filled by the host wrapper
};

// ------------------------------------------------------------------------------------------------ forward
constexpr int kFwdThreads = 512;

__global__ void __launch_bounds__(kFwdThreads, 2) k_decoder_fwd_h3(DecArgs a)
{
    __shared__ __attribute__((aligned(16))) char lds[H3_BYTES];
    {
        const float4 *src = reinterpret_cast<const float4 *>(a.packed + PK_TOTAL);
        float4 *dst = reinterpret_cast<float4 *>(lds);
        for (int i = threadIdx.x; i < H3_BYTES / 16; i += kFwdThreads) dst[i] = src[i];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, sl = lane & 31, h = lane >> 5;
    const long long ntiles = (a.N + 31) >> 5;
    const long long stride = (long long)gridDim.x * (kFwdThreads / 64);
    for (long long tile = (long long)blockIdx.x * (kFwdThreads / 64) + (threadIdx.x >> 6); tile < ntiles; tile += stride) {
        const long long n = tile * 32 + sl;
        const bool live = n < a.N;
        const long long nc = live ? n : a.N - 1;
        const float *row = a.feats + nc * a.ld_feats;
        // register g of half h = decoder input 16 (g >> 3) + 8 ((g >> 2) & 1) + 4 h + (g & 3)   (render_h3.h h3_ku)
        v16f x;
#pragma unroll
        for (int g = 0; g < 16; ++g) x[g] = row[16 * (g >> 3) + 8 * ((g >> 2) & 1) + 4 * h + (g & 3)];
        const float *dr = a.dirs + nc * a.ld_dirs;
        const float d[3] = { dr[0], dr[1], dr[2] };
        const float dnorm = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        v16f dinit[2];
        {
            float sh[16];
            ray_sh(d, dnorm, sh);
            h3_dinit(lds, lane, sh, dinit);   // B operand columns = samples: every sample its own direction
        }
        const SampleOut so = decode_tile_h3(lds, lane, x, dinit);
        if (live && h == 0) {
            a.sigma[n] = so.sigma;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                a.dif[3 * n + c] = so.dif[c];
                a.spec[3 * n + c] = so.spec[c];
                a.tint[3 * n + c] = so.tint[c];
            }
        }
    }
}

// ---- forward on 16-sample tiles (experiments build, SCANERF_DECODER_FWD_S16=1; comparison): decode_tile_s16 (render_t16.h) -- v_mfma_f32_16x16x32_f16
// on the t16s image, lane (c, q) = sample c, quarter q: the forward recompute of the backward kernel below, and the decoder of the
// render-time kernel k_pts_inference_t16.  94 registers -> four waves per SIMD, where the kernel is bound by vector-instruction
// issue (profiles/r05_decoder_fwd_counters.txt).  Measured per launch at 8.4e6 samples (rocprofv3 kernel trace, same box):
//   one launch between other kernels (the op inside a training step):  32-sample tiles 0.92 ms, 16-sample tiles 1.02 ms
//   13 launches back to back (sustained matrix load, lower clock):     32-sample tiles 1.22 ms, 16-sample tiles 1.13 ms
// so the op keeps the 32-sample-tile kernel and the long-running render-time kernels use this decoder.  The k-step grouping
// differs between the two (32 units per MFMA instead of 16): the f32 sums round differently in the last bits; both are the
// split-f16 evaluation.
constexpr int kFwd16Threads = 512;
#ifndef FWD16_WAVES
#define FWD16_WAVES 2
#endif
__global__ void __launch_bounds__(kFwd16Threads, FWD16_WAVES) k_decoder_fwd_s16(DecArgs a)
{
    __shared__ __attribute__((aligned(16))) char lds16[S16_BYTES];
    {
        const float4 *src = reinterpret_cast<const float4 *>(a.packed + WS_S16);
        float4 *dst = reinterpret_cast<float4 *>(lds16);
        for (int i = threadIdx.x; i < S16_BYTES / 16; i += kFwd16Threads) dst[i] = src[i];
    }
    __syncthreads();
    const char *lds = lds16;
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    const long long ntiles = (a.N + 15) >> 4;
    const long long stride = (long long)gridDim.x * (kFwd16Threads / 64);
    for (long long tile = (long long)blockIdx.x * (kFwd16Threads / 64) + (threadIdx.x >> 6); tile < ntiles; tile += stride) {
        const long long n = tile * 16 + c;
        const bool live = n < a.N;
        const long long nc = live ? n : a.N - 1;
        v4f xa, xb;
        {
            const float *row = a.feats + nc * a.ld_feats + 16 * (q & 1) + 4 * (q >> 1);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                xa[g] = row[g];
                xb[g] = row[8 + g];
            }
        }
        const float *dr = a.dirs + nc * a.ld_dirs;
        const float d[3] = { dr[0], dr[1], dr[2] };
        const SampleOut so = decode_tile_s16(lds, lane, xa, xb, d, 1e-8f);
        if (live && q == 0) {
            a.sigma[n] = so.sigma;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                a.dif[3 * n + k] = so.dif[k];
                a.spec[3 * n + k] = so.spec[k];
                a.tint[3 * n + k] = so.tint[k];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ backward
constexpr int kThreads = 512;
constexpr int kWaves = 8;
struct BL {   // LDS carve (the t16s images of render_common.h S16_*)
    static constexpr int kImg = S16_BYTES;
    static constexpr int kBias = S16_BIAS;
    static constexpr int kStageWave = 2 * T16_STAGE_WAVE;        // {Y, X, Y lo, X lo}
    static constexpr int kMx = kImg;                             // 8 floats (+ pad)
    static constexpr int kStage = kMx + 64;
    static constexpr int kBytes = kStage + kWaves * kStageWave;
    static_assert(kStage % 16 == 0, "LDS carve alignment");
};

__device__ __forceinline__ int fresh(int v)
{
    asm volatile("" : "+v"(v));
    return v;
}
__device__ __forceinline__ T16Lane fresh_lane(const T16Lane &L)
{
    T16Lane r;
    r.lo16 = fresh(L.lo16);
    r.w1 = fresh(L.w1);
    r.r1 = fresh(L.r1);
    r.r2 = fresh(L.r2);
    r.pos8 = fresh(L.pos8);
    r.trp = fresh(L.trp);
    return r;
}
template <int N>
__device__ __forceinline__ void zero4(v4f (&v)[N])
{
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = v4f{ 0, 0, 0, 0 };
}
__device__ __forceinline__ t16_h4 lo4(const t16_h8 &v) { return __builtin_shufflevector(v, v, 0, 1, 2, 3); }

// Weight-gradient blocks owned by this wave (render_bwd_t16.hip wgrad, SPLIT form): acc[i] += sum over the 4 tile pairs of
// dY[yb] X[xb0 + i]^T, both operands hi + lo, read back transposed from the pairs' staging images.
template <int NX, bool ROWSUM, int XSTRIDE = 1>
__device__ __forceinline__ void wgrad(v4f *acc, float &rowsum, const char *stage, const T16Lane &L, int yb, int x_mat_off, int xb0)
{
    constexpr int kWave = 2 * T16_STAGE_WAVE, kLo = 2 * T16_STAGE_MAT;
#pragma unroll
    for (int P = 0; P < 4; ++P) {
        const char *pm = stage + P * 2 * kWave;
        const t16_h8 a = t16_stage_get(pm, L, yb);
        t16_h8 b[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) b[i] = t16_stage_get(pm + x_mat_off, L, xb0 + i * XSTRIDE);
        if (ROWSUM) rowsum = t16_sum8(a, rowsum);
        const t16_h8 alo = t16_stage_get(pm + kLo, L, yb);
        if (ROWSUM) rowsum = t16_sum8(alo, rowsum);
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const t16_h8 blo = t16_stage_get(pm + x_mat_off + kLo, L, xb0 + i * XSTRIDE);
            acc[i] = t16_mfma(alo, b[i], acc[i]);
            acc[i] = t16_mfma(a, blo, acc[i]);
            acc[i] = t16_mfma(a, b[i], acc[i]);
        }
        __builtin_amdgcn_sched_barrier(0);   // (bounds the operands in flight)
    }
}
__device__ __forceinline__ void stage_put2(char *mat, const T16Lane &L, int b, const T16HL &v)
{
    t16_stage_put(mat, L, b, __builtin_shufflevector(v.hi, v.hi, 0, 1, 2, 3));
    t16_stage_put(mat, L, b + 1, __builtin_shufflevector(v.hi, v.hi, 4, 5, 6, 7));
    t16_stage_put(mat + 2 * T16_STAGE_MAT, L, b, __builtin_shufflevector(v.lo, v.lo, 0, 1, 2, 3));
    t16_stage_put(mat + 2 * T16_STAGE_MAT, L, b + 1, __builtin_shufflevector(v.lo, v.lo, 4, 5, 6, 7));
}
__device__ __forceinline__ v4f gauss_deriv(const v4f &u)   // G'(u) = -100 u G(u)
{
    v4f d;
#pragma unroll
    for (int g = 0; g < 4; ++g) d[g] = -100.0f * u[g] * gauss_fast(u[g]);
    return d;
}
__device__ __forceinline__ void stage_act(char *stX, const T16Lane &L, const v4f u[4])
{
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        v4f a0, a1;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            a0[g] = gauss_fast(u[2 * t][g]);
            a1[g] = gauss_fast(u[2 * t + 1][g]);
        }
        stage_put2(stX, L, 2 * t, t16_split(a0, a1));
    }
}
// the SH k-step (input k-step 1 of the D0 pairs) walked transposed: dSH = W_D0[:, 32:48]^T dv0.  Input blocks 2 and 3 of
// s16_chain's numbering: rows m of block 2 = SH[m] (m < 4), SH[8 + m - 4] (4 <= m < 8); of block 3 = SH[4 + m], SH[12 + m - 4];
// rows 8..15 meet zero weights.  So lane group q = 0 ends up with SH[0..3] / SH[4..7], q = 1 with SH[8..11] / SH[12..15].
__device__ __forceinline__ void chain_sh(v4f dsh[2], const char *img, int trp, const T16HL dY[2])
{
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int bi = 2; bi < 4; ++bi) {
            const char *a0 = img + T16_D0 + ((2 * t) * 2 + (bi >> 1)) * T16_PAIR + (bi & 1) * 512 + trp;
            const char *a1 = a0 + 2 * T16_PAIR;
            const t16_h8 ahi = __builtin_shufflevector(t16_tr4(a0), t16_tr4(a1), 0, 1, 2, 3, 4, 5, 6, 7);
            const t16_h8 alo = __builtin_shufflevector(t16_tr4(a0 + T16_SUB), t16_tr4(a1 + T16_SUB), 0, 1, 2, 3, 4, 5, 6, 7);
            dsh[bi - 2] = t16_mfma(alo, dY[t].hi, dsh[bi - 2]);
            dsh[bi - 2] = t16_mfma(ahi, dY[t].lo, dsh[bi - 2]);
            dsh[bi - 2] = t16_mfma(ahi, dY[t].hi, dsh[bi - 2]);
        }
}

template <bool DIRGRAD>
__global__ void __launch_bounds__(kThreads, 2) k_decoder_bwd_s16(DecArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    float *mxbuf = reinterpret_cast<float *>(lds + BL::kMx);
    {
        const float4 *src = reinterpret_cast<const float4 *>(a.packed + WS_S16);
        float4 *dst = reinterpret_cast<float4 *>(lds);
        for (int i = threadIdx.x; i < BL::kImg / 16; i += kThreads) dst[i] = src[i];
        float4 *stz = reinterpret_cast<float4 *>(lds + BL::kStage);   // finite contents wherever a step leaves a block unwritten
        for (int i = threadIdx.x; i < kWaves * BL::kStageWave / 16; i += kThreads) stz[i] = make_float4(0, 0, 0, 0);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char *stage = lds + BL::kStage;
    char *stY = lds + BL::kStage + wv * BL::kStageWave, *stX = stY + T16_STAGE_MAT;
    __builtin_amdgcn_s_setreg(1 | (23 << 6), 1);  // hwreg(HW_REG_MODE, 23, 1) = FP16_OVFL: f16 conversions saturate

    // ---- ownership of the weight-gradient blocks (as render_bwd_t16.hip)
    const int rb = wv >> 1, cb = wv & 1;
    v4f gW_D1[2], gW_L1[2], gW_D0[2], gW_L0[1], gW_nar[1];
    zero4(gW_D1); zero4(gW_L1); zero4(gW_D0); zero4(gW_L0); zero4(gW_nar);
    float gB_D1 = 0.0f, gB_L1 = 0.0f, gB_D0 = 0.0f, gB_L0 = 0.0f, gB_nar = 0.0f;
    int K = 0;                             // gradient scale 2^K of the workgroup (identical in its 8 waves)
    float sc = 1.0f, isc = 1.0f;

    const long long ntiles = (a.N + 15) >> 4;
    const long long ngroups = (ntiles + kWaves - 1) / kWaves;
    for (long long grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const int ln = fresh(lane);
        const int c = ln & 15, q = ln >> 4;
        T16Lane L = t16_lane(ln, BL::kStageWave);
        const long long n = (grp * kWaves + wv) * 16 + c;
        const bool live = n < a.N;
        const long long nc = live ? n : a.N - 1;
        // ---- inputs: this lane's 8 decoder inputs (positions 8q .. 8q+7 of render_t16.h = inputs 16 (q & 1) + 8 e + 4 (q >> 1) + g)
        v4f xa, xb;
        {
            const float *row = a.feats + nc * a.ld_feats + 16 * (q & 1) + 4 * (q >> 1);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                xa[g] = row[g];
                xb[g] = row[8 + g];
            }
        }
        float d[3];
        {
            const float *dr = a.dirs + nc * a.ld_dirs;
            d[0] = dr[0]; d[1] = dr[1]; d[2] = dr[2];
        }
        float gin[10];   // dL/d(sigma, dif xyz, tint xyz, spec xyz) of this lane's sample
        gin[0] = (a.g_sigma && live) ? a.g_sigma[n] : 0.0f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            gin[1 + k] = (a.g_dif && live) ? a.g_dif[3 * n + k] : 0.0f;
            gin[4 + k] = (a.g_tint && live) ? a.g_tint[3 * n + k] : 0.0f;
            gin[7 + k] = (a.g_spec && live) ? a.g_spec[3 * n + k] : 0.0f;
        }

        // ================= forward recompute (render_bwd_t16.hip, SPLIT) =================
        v4f ku0[4], khh[4], kv0[4], kv1[4];
        float dsig_dpre, dif[3], tint[3], spec[3];
        v4f shmine;   // SH[4q .. 4q+3] of this lane's sample: the SH "units" it stages for the D0 weight gradient
        {
            T16HL HB[2];
            {
                const T16HL xB = t16_split(xa, xb);
                v4f act[4];
#pragma unroll
                for (int b = 0; b < 4; ++b) act[b] = t16_bias(lds, 0, b, q, BL::kBias);
                s16_layer<4, 1>(act, lds, T16_L0, L.pos8, &xB);
#pragma unroll
                for (int b = 0; b < 4; ++b)
#pragma unroll
                    for (int g = 0; g < 4; ++g) act[b][g] = gauss_fast(act[b][g]);
                const T16HL aB[2] = { t16_split(act[0], act[1]), t16_split(act[2], act[3]) };
#pragma unroll
                for (int b = 0; b < 4; ++b) khh[b] = t16_bias(lds, 1, b, q, BL::kBias);
                s16_layer<4, 2>(khh, lds, T16_L1, L.pos8, aB);
                HB[0] = t16_split(khh[0], khh[1]);
                HB[1] = t16_split(khh[2], khh[3]);
            }
            {   // heads on H[:32]
                v4f hd[2] = { t16_ld4(lds, BL::kBias + 256 * 4), t16_ld4(lds, BL::kBias + 260 * 4) };
                s16_layer<2, 1>(hd, lds, T16_HEAD, L.pos8, &HB[0]);
                dsig_dpre = hd[0][0] > 20.0f ? 1.0f : sigmoid_fast(hd[0][0]);
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    dif[k] = sigmoid_fast(hd[0][1 + k]);
                    tint[k] = sigmoid_fast(hd[1][k]);
                }
            }
            T16HL cB[2];
            {   // Directional_MLP.mlp.0: bias + SH part (k-step 1: slot (q, j) = SH[8q + j] for q < 2, this SAMPLE's direction) + H[32:64]
                T16HL shB;
                {
                    const float dnorm = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
                    float sh[16];
                    ray_sh(d, dnorm, sh);
                    v4f s0, s1;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        s0[j] = q == 0 ? sh[j] : (q == 1 ? sh[8 + j] : 0.0f);
                        s1[j] = q == 0 ? sh[4 + j] : (q == 1 ? sh[12 + j] : 0.0f);
                        shmine[j] = q == 0 ? sh[j] : (q == 1 ? sh[4 + j] : (q == 2 ? sh[8 + j] : sh[12 + j]));
                    }
                    shB = t16_split(s0, s1);
                }
#pragma unroll
                for (int b = 0; b < 4; ++b) kv0[b] = t16_bias(lds, 2, b, q, BL::kBias);
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const char *p1 = lds + T16_D0 + (b * 2 + 1) * T16_PAIR + L.pos8, *p0 = lds + T16_D0 + (b * 2) * T16_PAIR + L.pos8;
                    const t16_h8 shi = s16_lda(p1), slo = s16_lda(p1 + T16_SUB);
                    const t16_h8 ahi = s16_lda(p0), alo = s16_lda(p0 + T16_SUB);
                    kv0[b] = t16_mfma(slo, shB.hi, kv0[b]);
                    kv0[b] = t16_mfma(shi, shB.lo, kv0[b]);
                    kv0[b] = t16_mfma(shi, shB.hi, kv0[b]);
                    kv0[b] = t16_mfma(alo, HB[1].hi, kv0[b]);
                    kv0[b] = t16_mfma(ahi, HB[1].lo, kv0[b]);
                    kv0[b] = t16_mfma(ahi, HB[1].hi, kv0[b]);
                }
                v4f act[4];
#pragma unroll
                for (int b = 0; b < 4; ++b)
#pragma unroll
                    for (int g = 0; g < 4; ++g) act[b][g] = gauss_fast(kv0[b][g]);
                cB[0] = t16_split(act[0], act[1]);
                cB[1] = t16_split(act[2], act[3]);
            }
            {
#pragma unroll
                for (int b = 0; b < 4; ++b) kv1[b] = t16_bias(lds, 3, b, q, BL::kBias);
                s16_layer<4, 2>(kv1, lds, T16_D1, L.pos8, cB);
                v4f act[4];
#pragma unroll
                for (int b = 0; b < 4; ++b)
#pragma unroll
                    for (int g = 0; g < 4; ++g) act[b][g] = gauss_fast(kv1[b][g]);
                cB[0] = t16_split(act[0], act[1]);
                cB[1] = t16_split(act[2], act[3]);
            }
            {
                v4f r[1] = { t16_ld4(lds, BL::kBias + 264 * 4) };
                s16_layer<1, 2>(r, lds, T16_D2, L.pos8, cB);
#pragma unroll
                for (int k = 0; k < 3; ++k) spec[k] = sigmoid_fast(r[0][k]);
            }
        }

        // ================= gradients w.r.t. the head / rgb pre-activations =================
        float gh[7], gs3[3];
        gh[0] = gin[0] * dsig_dpre;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            gh[1 + k] = gin[1 + k] * dif[k] * (1.0f - dif[k]);
            gh[4 + k] = gin[4 + k] * tint[k] * (1.0f - tint[k]);
            gs3[k] = gin[7 + k] * spec[k] * (1.0f - spec[k]);
        }
        {   // this tile's largest |gradient|, published for the workgroup's scale
            float mx = 0.0f;
#pragma unroll
            for (int k = 0; k < 7; ++k) mx = fmaxf(mx, fabsf(gh[k]));
#pragma unroll
            for (int k = 0; k < 3; ++k) mx = fmaxf(mx, fabsf(gs3[k]));
            mx = fmaxf(mx, row_ror<8>(mx));
            mx = fmaxf(mx, row_ror<4>(mx));
            mx = fmaxf(mx, row_ror<2>(mx));
            mx = fmaxf(mx, row_ror<1>(mx));
            if (lane == 0) mxbuf[wv] = mx;
        }
        __syncthreads();  // ---- S: tile maxima visible; every wave is done with the previous tile's staged operands
        {
            const float4 m0 = reinterpret_cast<const float4 *>(mxbuf)[0], m1 = reinterpret_cast<const float4 *>(mxbuf)[1];
            const float mx = fmaxf(fmaxf(fmaxf(m0.x, m0.y), fmaxf(m0.z, m0.w)), fmaxf(fmaxf(m1.x, m1.y), fmaxf(m1.z, m1.w)));
            const float ms = mx * sc;
            if (mx > 0.0f && mx < 3.0e38f && (ms >= 64.0f || ms < 4.0f)) {
                int e;
                frexpf(mx, &e);
                int Kn = 6 - e;  // mx * 2^K in [32, 64)
                Kn = Kn > K + 100 ? K + 100 : (Kn < K - 100 ? K - 100 : Kn);
                Kn = Kn > 100 ? 100 : (Kn < -100 ? -100 : Kn);
                const float r = ldexpf(1.0f, Kn - K);
                K = Kn;
                sc = ldexpf(1.0f, K);
                isc = ldexpf(1.0f, -K);
#pragma unroll
                for (int i = 0; i < 2; ++i) { gW_D1[i] *= r; gW_L1[i] *= r; }
                gW_D0[0] *= r; gW_D0[1] *= r; gW_L0[0] *= r; gW_nar[0] *= r;
                gB_D1 *= r; gB_L1 *= r; gB_D0 *= r; gB_L0 *= r; gB_nar *= r;
            }
#pragma unroll
            for (int k = 0; k < 7; ++k) gh[k] *= sc;
#pragma unroll
            for (int k = 0; k < 3; ++k) gs3[k] *= sc;
        }
        v4f dx[2];
        zero4(dx);
        constexpr int kLo = 2 * T16_STAGE_MAT;
        const v4f zero = { 0, 0, 0, 0 };
        // ================= narrow layers: heads (32 -> 7) and rgb (64 -> 3) =================
        L = fresh_lane(L);
        T16HL narS;
        {
            v4f nar;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float r1 = g < 3 ? gh[4 + g] : 0.0f, r2 = g < 3 ? gs3[g] : 0.0f;
                nar[g] = q == 0 ? gh[g] : (q == 1 ? r1 : (q == 2 ? r2 : 0.0f));
            }
            narS = t16_split(nar, zero);
            t16_stage_put(stY, L, 0, lo4(narS.hi));
            t16_stage_put(stY + kLo, L, 0, lo4(narS.lo));
            stage_put2(stY, L, 2, t16_split(khh[0], khh[1]));   // X operand of the heads' weight gradient: H[:32] in blocks 2, 3 of Y
            stage_act(stX, L, kv1);                              // X operand of the rgb layer's weight gradient: c1 = G(v1)
        }
        __syncthreads();  // ---- A1
        if (wv == 0) wgrad<1, true>(gW_nar, gB_nar, stage, L, 0, 0, 2);
        else if (wv == 1) { float dummy = 0.0f; wgrad<1, false>(gW_nar, dummy, stage, L, 0, 0, 3); }
        else if (wv < 6) { float dummy = 0.0f; wgrad<1, false>(gW_nar, dummy, stage, L, 0, T16_STAGE_MAT, wv - 2); }
        T16HL dyS[2];
        {   // dv1 = (W_rgb^T gs3) * G'(v1)
            v4f dc[4];
            zero4(dc);
            s16_chain_narrow<4>(dc, lds, S16T_D2, L.lo16, narS);
#pragma unroll
            for (int b = 0; b < 4; ++b) dc[b] *= gauss_deriv(kv1[b]);
            dyS[0] = t16_split(dc[0], dc[1]);
            dyS[1] = t16_split(dc[2], dc[3]);
        }
        __syncthreads();  // ---- B1
        // ================= Directional_MLP.mlp.2 (64 -> 64) =================
        L = fresh_lane(L);
        stage_put2(stY, L, 0, dyS[0]);
        stage_put2(stY, L, 2, dyS[1]);
        stage_act(stX, L, kv0);
        __syncthreads();  // ---- A2
        if (cb == 0) wgrad<2, true>(gW_D1, gB_D1, stage, L, rb, T16_STAGE_MAT, 0);
        else { float dummy = 0.0f; wgrad<2, false>(gW_D1, dummy, stage, L, rb, T16_STAGE_MAT, 2); }
        {
            v4f dc[4];
            zero4(dc);
            s16_chain<4, 2, 2>(dc, lds, T16_D1, L.trp, dyS);
#pragma unroll
            for (int b = 0; b < 4; ++b) dc[b] *= gauss_deriv(kv0[b]);   // dv0
            dyS[0] = t16_split(dc[0], dc[1]);
            dyS[1] = t16_split(dc[2], dc[3]);
        }
        __syncthreads();  // ---- B2
        // ================= Directional_MLP.mlp.0 (H[32:64] and the 16 harmonics of the sample's direction) =================
        L = fresh_lane(L);
        stage_put2(stY, L, 0, dyS[0]);
        stage_put2(stY, L, 2, dyS[1]);
        stage_put2(stX, L, 0, t16_split(khh[2], khh[3]));
        {   // the SH part of the layer's input: 16 more "units" (block 2 of X), this sample's own
            const T16HL shS = t16_split(shmine, zero);
            t16_stage_put(stX, L, 2, lo4(shS.hi));
            t16_stage_put(stX + kLo, L, 2, lo4(shS.lo));
        }
        __syncthreads();  // ---- A3
        if (cb == 0) wgrad<2, true, 2>(gW_D0, gB_D0, stage, L, rb, T16_STAGE_MAT, 0);   // x = H[32:48] and SH
        else { float dummy = 0.0f; wgrad<1, false>(gW_D0, dummy, stage, L, rb, T16_STAGE_MAT, 1); }   // x = H[48:64]
        if (DIRGRAD) {
            // dL/dSH = W_D0[:, 32:48]^T dv0 -> through the harmonics and the normalisation -> dL/d(direction) of this sample
            v4f dsh[2];
            zero4(dsh);
            chain_sh(dsh, lds, L.trp, dyS);
            // lane group q = 0: gsh[0..3] = dsh[0], gsh[4..7] = dsh[1]; q = 1: gsh[8..11] = dsh[0], gsh[12..15] = dsh[1]
            const float dn = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]), inv = 1.0f / (dn + 1e-8f);
            const float x = d[0] * inv, y = d[1] * inv, z = d[2] * inv;
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            constexpr float C1 = 0.4886025119029199f, C20 = 1.0925484305920792f, C21 = -1.0925484305920792f, C22 = 0.31539156525252005f,
                            C23 = -1.0925484305920792f, C24 = 0.5462742152960396f, C30 = -0.5900435899266435f, C31 = 2.890611442640554f,
                            C32 = -0.4570457994644658f, C33 = 0.3731763325901154f, C34 = -0.4570457994644658f, C35 = 1.445305721320277f,
                            C36 = -0.5900435899266435f;
            float gu[3];
            {
                const v4f &A = dsh[0], &B = dsh[1];
                // q == 0: A = gsh[0..3], B = gsh[4..7]
                const float gx0 = A[3] * C1 + B[0] * C20 * y - B[2] * 2.0f * C22 * x + B[3] * C23 * z;
                const float gy0 = A[1] * C1 + B[0] * C20 * x + B[1] * C21 * z - B[2] * 2.0f * C22 * y;
                const float gz0 = A[2] * C1 + B[1] * C21 * y + B[2] * 4.0f * C22 * z + B[3] * C23 * x;
                // q == 1: A = gsh[8..11], B = gsh[12..15]
                const float gx1 = A[0] * 2.0f * C24 * x + A[1] * 6.0f * C30 * xy + A[2] * C31 * yz - A[3] * 2.0f * C32 * xy - B[0] * 6.0f * C33 * xz +
                                  B[1] * C34 * (4.0f * zz - 3.0f * xx - yy) + B[2] * 2.0f * C35 * xz + B[3] * C36 * (3.0f * xx - 3.0f * yy);
                const float gy1 = -A[0] * 2.0f * C24 * y + A[1] * C30 * (3.0f * xx - 3.0f * yy) + A[2] * C31 * xz + A[3] * C32 * (4.0f * zz - xx - 3.0f * yy) -
                                  B[0] * 6.0f * C33 * yz - B[1] * 2.0f * C34 * xy - B[2] * 2.0f * C35 * yz - B[3] * 6.0f * C36 * xy;
                const float gz1 = A[2] * C31 * xy + A[3] * 8.0f * C32 * yz + B[0] * C33 * (6.0f * zz - 3.0f * xx - 3.0f * yy) + B[1] * 8.0f * C34 * xz +
                                  B[2] * C35 * (xx - yy);
                gu[0] = q == 0 ? gx0 : (q == 1 ? gx1 : 0.0f);
                gu[1] = q == 0 ? gy0 : (q == 1 ? gy1 : 0.0f);
                gu[2] = q == 0 ? gz0 : (q == 1 ? gz1 : 0.0f);
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) gu[k] += __shfl_xor(gu[k], 16, 64);   // lanes q = 0 and q = 1 of a sample
            // u = d / (|d| + eps): g_d = g_u inv - (g_u . d) inv^2 d / |d|
            const float dot = gu[0] * d[0] + gu[1] * d[1] + gu[2] * d[2];
            const float coef = dn > 0.0f ? -dot * inv * inv / dn : 0.0f;
            if (live && q == 0) {
                float *o = a.d_dirs + n * a.ld_ddirs;
#pragma unroll
                for (int k = 0; k < 3; ++k) o[k] = (gu[k] * inv + coef * d[k]) * isc;
            }
        }
        {
            v4f dH[4];
            zero4(dH);
            s16_chain<2, 2, 2>(&dH[2], lds, T16_D0, L.trp, dyS);        // dH[32:64] = W_D0[:, :32]^T dv0
            s16_chain_narrow<2>(&dH[0], lds, S16T_HEAD, L.lo16, narS);  // dH[0:32] = heads^T gh
            dyS[0] = t16_split(dH[0], dH[1]);
            dyS[1] = t16_split(dH[2], dH[3]);
        }
        __syncthreads();  // ---- B3
        // ================= Spatial_MLP.mlp.2 (64 -> 64, linear) =================
        L = fresh_lane(L);
        stage_put2(stY, L, 0, dyS[0]);
        stage_put2(stY, L, 2, dyS[1]);
        {   // u0 = W0 x + b0 again
            const T16HL xB = t16_split(xa, xb);
#pragma unroll
            for (int b = 0; b < 4; ++b) ku0[b] = t16_bias(lds, 0, b, q, BL::kBias);
            s16_layer<4, 1>(ku0, lds, T16_L0, L.pos8, &xB);
        }
        stage_act(stX, L, ku0);
        __syncthreads();  // ---- A4
        if (cb == 0) wgrad<2, true>(gW_L1, gB_L1, stage, L, rb, T16_STAGE_MAT, 0);
        else { float dummy = 0.0f; wgrad<2, false>(gW_L1, dummy, stage, L, rb, T16_STAGE_MAT, 2); }
        {
            v4f dc[4];
            zero4(dc);
            s16_chain<4, 2, 2>(dc, lds, T16_L1, L.trp, dyS);
#pragma unroll
            for (int b = 0; b < 4; ++b) dc[b] *= gauss_deriv(ku0[b]);   // du0
            dyS[0] = t16_split(dc[0], dc[1]);
            dyS[1] = t16_split(dc[2], dc[3]);
        }
        __syncthreads();  // ---- B4
        // ================= Spatial_MLP.mlp.0 (32 -> 64) =================
        L = fresh_lane(L);
        stage_put2(stY, L, 0, dyS[0]);
        stage_put2(stY, L, 2, dyS[1]);
        stage_put2(stX, L, 0, t16_split(xa, xb));
        __syncthreads();  // ---- A5
        if (cb == 0) wgrad<1, true>(gW_L0, gB_L0, stage, L, rb, T16_STAGE_MAT, 0);
        else { float dummy = 0.0f; wgrad<1, false>(gW_L0, dummy, stage, L, rb, T16_STAGE_MAT, 1); }
        s16_chain<2, 2, 1>(dx, lds, T16_L0, L.trp, dyS);
        if (live) {   // dL/d(decoder inputs) of this lane's 8 positions (the layer-0 image carries weight_feature: no factor left)
            float *o = a.d_feats + n * a.ld_dfeats + 16 * (q & 1) + 4 * (q >> 1);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                o[g] = dx[0][g] * isc;
                o[8 + g] = dx[1][g] * isc;
            }
        }
    }

    // ---- flush: the workgroup's partial sums in blob order, ONE row per workgroup (every entry has exactly one owner)
    const int c = lane & 15, q = lane >> 4;
    float *out = a.dw_partial + (size_t)blockIdx.x * SCANERF_PARAMSIZE;
    auto put64 = [&](const v4f &acc, int base, int cblk) {   // 64-output layer: acc = dW[n = 16rb + 4q + g][k = 16 cblk + c]
#pragma unroll
        for (int g = 0; g < 4; ++g) out[base + 64 + (16 * cblk + c) * 64 + 16 * rb + 4 * q + g] = acc[g] * isc;
    };
    put64(gW_D1[0], BLOB_D1, 2 * cb);
    put64(gW_D1[1], BLOB_D1, 2 * cb + 1);
    put64(gW_L1[0], BLOB_S1, 2 * cb);
    put64(gW_L1[1], BLOB_S1, 2 * cb + 1);
    put64(gW_D0[0], BLOB_D0, cb);
    if (cb == 0) put64(gW_D0[1], BLOB_D0, 2);   // SH part: input index 32 + c
    {
        const int kin = t16_pos_to_input(t16_l0_row_to_pos(cb, c));
#pragma unroll
        for (int g = 0; g < 4; ++g) out[BLOB_S0 + 64 + kin * 64 + 16 * rb + 4 * q + g] = gW_L0[0][g] * isc;
    }
    auto rowtotal = [&](float v) {  // sum over the 4 lane groups: lane = unit
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        return v;
    };
    if (cb == 0) {
        float v;
        v = rowtotal(gB_D1) * isc;
        if (q == 0) out[BLOB_D1 + 16 * rb + c] = v;
        v = rowtotal(gB_L1) * isc;
        if (q == 0) out[BLOB_S1 + 16 * rb + c] = v;
        v = rowtotal(gB_L0) * isc;
        if (q == 0) out[BLOB_S0 + 16 * rb + c] = v;
        v = rowtotal(gB_D0) * isc;
        if (q == 0) out[BLOB_D0 + 16 * rb + c] = v;
    }
    if (wv < 2) {  // heads: acc = d[row 4q + g][H unit 16 wv + c]; rows 0 sigma, 1-3 dif, 4-6 tint
        const int k = 16 * wv + c;
        if (q == 0) {
            out[BLOB_SIG + 1 + k] = gW_nar[0][0] * isc;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) out[BLOB_DIF + 3 + k * 3 + ch] = gW_nar[0][1 + ch] * isc;
        } else if (q == 1) {
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) out[BLOB_TINT + 3 + k * 3 + ch] = gW_nar[0][ch] * isc;
        }
    } else if (wv < 6 && q == 2) {  // rgb layer: rows 8-10, column = c1 unit 16(wv-2) + c
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) out[BLOB_D2 + 3 + (16 * (wv - 2) + c) * 3 + ch] = gW_nar[0][ch] * isc;
    }
    if (wv == 0) {  // narrow biases = row sums of the narrow block
        const float v = rowtotal(gB_nar) * isc;
        if (q == 0) {
            if (c == 0) out[BLOB_SIG] = v;
            else if (c < 4) out[BLOB_DIF + c - 1] = v;
            else if (c < 7) out[BLOB_TINT + c - 4] = v;
            else if (c >= 8 && c < 11) out[BLOB_D2 + c - 8] = v;
        }
    }
}

// grad_blob[e] += sum over the workgroups' partial rows (fixed order: deterministic); first-layer weights carry the folded
// weight_feature (render_bwd.hip k_reduce_dw, same form)
__global__ void __launch_bounds__(1024) k_decoder_reduce_dw(const float *__restrict__ partial, int nrows, const float *__restrict__ wf,
                                                            float *__restrict__ grad_blob)
{
    __shared__ float part[16][64];
    const int c = threadIdx.x & 63, r = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + c;
    float s = 0.0f;
    if (e < SCANERF_PARAMSIZE)
        for (int w = r; w < nrows; w += 16) s += partial[(size_t)w * SCANERF_PARAMSIZE + e];
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

int decoder_grid(long long N)
{
    const long long groups = (((N + 15) >> 4) + kWaves - 1) / kWaves;
    return (int)(groups < 1 ? 1 : (groups > kNumCU ? kNumCU : groups));
}

}  // namespace

// ---------------------------------------------------------------------------- C ABI
SCANERF_API int scanerf_decoder_forward(const float *feats, int ld_feats, const float *dirs, int ld_dirs, const float *workspace,
                                        float *sigma, float *diffuse, float *specular, float *tint, long long N,
                                        scanerf_stream_t stream)
{
    SCANERF_REQUIRE(N >= 0, "decoder_forward: N=%lld", N);
    if (N == 0) return 0;
    SCANERF_REQUIRE(feats && dirs && workspace && sigma && diffuse && specular && tint, "decoder_forward: null pointer");
    SCANERF_REQUIRE(ld_feats >= 32 && ld_dirs >= 3, "decoder_forward: row strides %d / %d (floats) are too small", ld_feats, ld_dirs);
    SCANERF_REQUIRE(((uintptr_t)workspace & 15) == 0, "decoder_forward: workspace must be 16-byte aligned");
    DecArgs a = {};
    a.feats = feats; a.ld_feats = ld_feats; a.dirs = dirs; a.ld_dirs = ld_dirs; a.packed = workspace; a.N = N;
    a.sigma = sigma; a.dif = diffuse; a.spec = specular; a.tint = tint;
#ifdef SCANERF_EXPERIMENTS
    if (tune_int("SCANERF_DECODER_FWD_S16", 0)) {   // 16-sample tiles at four waves per SIMD (see k_decoder_fwd_s16; experiments build)
        const long long nt16 = (N + 15) >> 4;
        long long b16 = (nt16 + 7) / 8;
        if (b16 > 2 * kNumCU) b16 = 2 * kNumCU;   // two resident 512-thread workgroups per CU (77 KB of LDS each), persistent
        hipLaunchKernelGGL(k_decoder_fwd_s16, dim3((int)b16), dim3(kFwd16Threads), 0, (hipStream_t)stream, a);
        return check_launch("decoder_forward(s16)");
    }
#endif
    const long long ntiles = (N + 31) >> 5;
    long long blocks = (ntiles + 7) / 8;
    if (blocks > 2 * kNumCU) blocks = 2 * kNumCU;
    hipLaunchKernelGGL(k_decoder_fwd_h3, dim3((int)blocks), dim3(kFwdThreads), 0, (hipStream_t)stream, a);
    return check_launch("decoder_forward");
}

SCANERF_API int scanerf_decoder_backward_grid(long long N) { return decoder_grid(N); }

SCANERF_API int scanerf_decoder_backward(const float *feats, int ld_feats, const float *dirs, int ld_dirs, const float *workspace,
                                         const float *weight_feature, const float *g_sigma, const float *g_diffuse,
                                         const float *g_specular, const float *g_tint, float *d_feats, int ld_dfeats,
                                         float *d_dirs, int ld_ddirs, float *dw_partial, float *grad_blob, long long N,
                                         scanerf_stream_t stream)
{
    SCANERF_REQUIRE(N >= 0, "decoder_backward: N=%lld", N);
    if (N == 0) return 0;
    SCANERF_REQUIRE(feats && dirs && workspace && weight_feature && d_feats && dw_partial && grad_blob, "decoder_backward: null pointer");
    SCANERF_REQUIRE(ld_feats >= 32 && ld_dirs >= 3 && ld_dfeats >= 32 && (!d_dirs || ld_ddirs >= 3),
                    "decoder_backward: row strides %d / %d / %d / %d (floats) are too small", ld_feats, ld_dirs, ld_dfeats, ld_ddirs);
    SCANERF_REQUIRE(((uintptr_t)workspace & 15) == 0, "decoder_backward: workspace must be 16-byte aligned");
    DecArgs a = {};
    a.feats = feats; a.ld_feats = ld_feats; a.dirs = dirs; a.ld_dirs = ld_dirs; a.packed = workspace; a.N = N;
    a.g_sigma = g_sigma; a.g_dif = g_diffuse; a.g_spec = g_specular; a.g_tint = g_tint;
    a.d_feats = d_feats; a.ld_dfeats = ld_dfeats; a.d_dirs = d_dirs; a.ld_ddirs = ld_ddirs; a.dw_partial = dw_partial;
    const int blocks = decoder_grid(N);
    hipStream_t st = (hipStream_t)stream;
    hipError_t me = hipMemsetAsync(dw_partial, 0, (size_t)blocks * SCANERF_PARAMSIZE * sizeof(float), st);
    SCANERF_REQUIRE(me == hipSuccess, "decoder_backward: memset failed: %s", hipGetErrorString(me));
    const size_t lds_bytes = BL::kBytes;
#define SCANERF_LAUNCH_DEC(DG)                                                                                                  \
    {                                                                                                                          \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_decoder_bwd_s16<DG>),                             \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);                        \
        SCANERF_REQUIRE(e == hipSuccess, "decoder_backward: cannot reserve %zu B of LDS: %s", lds_bytes, hipGetErrorString(e)); \
        hipLaunchKernelGGL((k_decoder_bwd_s16<DG>), dim3(blocks), dim3(kThreads), lds_bytes, st, a);                           \
    }
    if (d_dirs) SCANERF_LAUNCH_DEC(true)
    else SCANERF_LAUNCH_DEC(false)
#undef SCANERF_LAUNCH_DEC
    if (int e = check_launch("decoder_backward")) return e;
    hipLaunchKernelGGL(k_decoder_reduce_dw, dim3(ceil_div(SCANERF_PARAMSIZE, 64)), dim3(1024), 0, st, dw_partial, blocks,
                       weight_feature, grad_blob);
    return check_launch("decoder_backward(reduce)");
}

// render_bwd.hip -- fused per-ray volume rendering, backward (gfx950).
//
// Adjoint of render.hip (hashgrid/__init__.py:512-596 under torch autograd in the reference):
// given dL/d(out_ray) it produces
//   * dL/d(hash features of every sample), level-major [16][B*S][2]  (fed to scatter.hip),
//   * dL/d(decoder blob) as per-wave partial sums (reduced by k_reduce_dw, deterministic),
// recomputing the forward per 32-sample tile instead of storing per-sample activations.
//
// One wave per ray, 4 waves per workgroup, ONE wave per SIMD: the wave owns the whole
// 512-register file -- ~235 registers hold its private weight-gradient accumulators
// (MFMA C/D operands), the rest the tile's activations.  Three kinds of MFMA work per tile:
//   forward recompute      H^T   = W   X^T      A = weights (LDS),    B = activations (regs)
//   activation gradients   dX^T  = W^T dY^T     A = W^T image (LDS),  B = dY (regs)
//   weight gradients       dW    = dY  X^T      reduction over SAMPLES: both operands need the
//                                               unit on the lane and samples across steps, the
//                                               transpose of the register layout -> one trip
//                                               through a wave-private 9 KB LDS scratch each.
// Compositing backward walks the tiles LAST to FIRST with the tile-entry transmittances saved
// by the forward, so suffix sums are formed directly (a prefix-minus-total form divides
// rounding noise by f_i = 1-alpha_i+1e-6 and loses opaque samples).
#include "render_device.h"

using namespace scanerf;

namespace {

constexpr int kBwdThreads = 256;
constexpr int kScrStride = 36;                 // floats per scratch row (32 samples + pad: conflict-free b128 reads)
constexpr int kScrFloats = 64 * kScrStride;    // per wave
constexpr int kBwdLdsFloats = PK_TOTAL + PKT_TOTAL + 64 + 4 * kScrFloats;

struct BwdArgs {
    RenderArgs f;              // forward inputs (out_ray = forward outputs, read-only here)
    const float *grad_out;     // [B,16] dL/d(out_ray)
    const float *tile_T;       // [B, ntiles] from the forward
    const float *packed_t;     // PKT_TOTAL floats (transposed images)
    float *dfeat;              // [16][B*S][2]
    float *dw_partial;         // [nwaves][SCANERF_PARAMSIZE]
};

__device__ __forceinline__ float dgauss(float u, float a) { return -100.0f * u * a; }  // d/du exp(-50 u^2)

// ---- wave-private transposes through LDS ------------------------------------------------------
// registers (lane = sample s, half h, reg g = unit nmap(g,h))  ->  scratch[row = unit][col = sample]
__device__ __forceinline__ void scr_put(float *scr, int lane, int rowbase, const v16f &v)
{
    const int sl = lane & 31, h = lane >> 5;
#pragma unroll
    for (int g = 0; g < 16; ++g) scr[(rowbase + nmap(g, h)) * kScrStride + sl] = v[g];
}
// scratch -> registers (lane = unit n, half h, reg t = sample 16h + t): the operand layout of an
// MFMA whose reduction index is the sample
__device__ __forceinline__ v16f scr_get(const float *scr, int lane, int rowbase)
{
    const float4 *p = reinterpret_cast<const float4 *>(scr + (rowbase + (lane & 31)) * kScrStride + 16 * (lane >> 5));
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

// acc[n][k] += sum_s dY[n][s] X[k][s]   (both operands in the sample-on-steps layout)
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

template <int DT>
__global__ void __launch_bounds__(kBwdThreads, 1) k_render_bwd(BwdArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *ldt = lds + PK_TOTAL;                       // transposed images
    int *lres = reinterpret_cast<int *>(lds + PK_TOTAL + PKT_TOTAL);
    {
        const float4 *src = reinterpret_cast<const float4 *>(a.f.packed);
        const float4 *srct = reinterpret_cast<const float4 *>(a.packed_t);
        float4 *dst = reinterpret_cast<float4 *>(lds);
        for (int i = threadIdx.x; i < PK_TOTAL / 4; i += kBwdThreads) dst[i] = src[i];
        float4 *dstt = reinterpret_cast<float4 *>(ldt);
        for (int i = threadIdx.x; i < PKT_TOTAL / 4; i += kBwdThreads) dstt[i] = srct[i];
        if (threadIdx.x < 64) {
            int lv = threadIdx.x >> 2, c = threadIdx.x & 3;
            lres[threadIdx.x] = c < 3 ? a.f.resolutions[3 * lv + c] : 0;
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, sl = lane & 31, h = lane >> 5, wv = threadIdx.x >> 6;
    float *scr = lds + PK_TOTAL + PKT_TOTAL + 64 + wv * kScrFloats;
    const int wave0 = blockIdx.x * (kBwdThreads / 64) + wv, nwaves = gridDim.x * (kBwdThreads / 64);
    const int S = a.f.S, ntiles = (S + 31) >> 5;

    // ---- weight-gradient accumulators (live for the whole kernel) --------------------------------
    const v16f zero16 = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    v16f gW_L0[2] = { zero16, zero16 };
    v16f gW_L1[2][2] = { { zero16, zero16 }, { zero16, zero16 } };
    v16f gW_D0H[2] = { zero16, zero16 };
    v16f gW_D1[2][2] = { { zero16, zero16 }, { zero16, zero16 } };
    float gW_D0S[2][8], gW_head[7], gW_D2[2][3], gB[4][2], gB_head[7], gB_d2[3];
#pragma unroll
    for (int i = 0; i < 8; ++i) gW_D0S[0][i] = gW_D0S[1][i] = 0.0f;
#pragma unroll
    for (int i = 0; i < 7; ++i) gW_head[i] = gB_head[i] = 0.0f;
#pragma unroll
    for (int i = 0; i < 3; ++i) gW_D2[0][i] = gW_D2[1][i] = gB_d2[i] = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) gB[i][0] = gB[i][1] = 0.0f;

    for (int ray = wave0; ray < a.f.B; ray += nwaves) {
        const bool rvalid = !(a.f.ray_valid && !a.f.ray_valid[ray]);
        if (!rvalid) {
            // invalid rays contribute nothing; their feature gradients are zero
            for (int s = sl; s < S; s += 32)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int level = 4 * (j >> 1) + 2 * h + (j & 1);
                    reinterpret_cast<float2 *>(a.dfeat)[(size_t)level * a.f.B * S + (size_t)ray * S + s] = make_float2(0, 0);
                }
            continue;
        }
        float o[3], d[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            o[k] = a.f.rays_o[3 * ray + k];
            d[k] = a.f.rays_d[3 * ray + k];
        }
        const float dnorm = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        float sh[16];
        ray_sh(d, dnorm, sh);
        v16f dinit[2];
        {
            v16f shb;
#pragma unroll
            for (int r = 0; r < 8; ++r) shb[r] = h ? sh[2 * r + 1] : sh[2 * r];
            dinit[0] = load_bias(lds, 2, 0, h);
            dinit[1] = load_bias(lds, 2, 1, h);
            const float4 *A = reinterpret_cast<const float4 *>(lds + PK_D0S) + lane;
            float4 a00 = A[0], a01 = A[64], a10 = A[128], a11 = A[192];
            MFMA4(dinit[0], a00, shb[0], shb[1], shb[2], shb[3])
            MFMA4(dinit[0], a01, shb[4], shb[5], shb[6], shb[7])
            MFMA4(dinit[1], a10, shb[0], shb[1], shb[2], shb[3])
            MFMA4(dinit[1], a11, shb[4], shb[5], shb[6], shb[7])
        }
        // upstream gradients of this ray and the forward outputs they refer to
        const float *go = a.grad_out + (size_t)ray * SCANERF_RAY_OUT;
        const float *fo = a.f.out_ray + (size_t)ray * SCANERF_RAY_OUT;
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
        float Rsuf = 0.0f;  // sum_{j in later tiles} a_j w_j

        for (int tile = ntiles - 1; tile >= 0; --tile) {
            const int s = tile * 32 + sl;
            const bool live = s < S;
            const float z = live ? a.f.z_vals[(size_t)ray * S + s] : 0.0f;
            float delta = live ? a.f.dists[(size_t)ray * S + s] * dnorm : 0.0f;
            if (a.f.infinity && s == S - 1) delta = 1e10f;
            float p[3];
            contract_point(a.f, o, d, z, p);

            // ================= forward recompute =================
            v16f x;
            encode8<DT>(a.f, lres, h, p, x);
            v16f u0[2] = { load_bias(lds, 0, 0, h), load_bias(lds, 0, 1, h) };
            mma_block16(u0[0], lds + PK_L0, 0, lane, x);
            mma_block16(u0[1], lds + PK_L0, 4, lane, x);
            v16f H[2] = { load_bias(lds, 1, 0, h), load_bias(lds, 1, 1, h) };
            {
                v16f a0 = act16(u0[0]), a1 = act16(u0[1]);
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
            v16f v0[2] = { dinit[0], dinit[1] };
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

            // ================= compositing: recompute, then adjoint =================
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
            const float Ti = a.tile_T[(size_t)ray * ntiles + tile] * excl;
            const float w = alpha * Ti;

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
            const float suffix = Rsuf + rs - aw;
            Rsuf += __shfl(rs, 0, 32);
            float dalpha = Ti * ai - (suffix + ((s < S - 1) ? gTl * Tl : 0.0f)) / fi;
            if (!live) dalpha = 0.0f;
            const float dsigma = dalpha * delta * ex;
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

            // small per-sample rows (head / rgb pre-activation gradients) for the VALU weight grads:
            // scratch rows [c][sample]; read back as broadcasts
            auto put_rows = [&](const float *vals, int n) {
                if (h == 0)
                    for (int c = 0; c < n; ++c) scr[c * kScrStride + sl] = vals[c];
            };

            // ================= decoder adjoint =================
            // ---- Directional_MLP.mlp.4 (64 -> 3): dc1 = W^T g ; weight grads on the VALU
            v16f dv1[2];
            {
                const float4 *W = reinterpret_cast<const float4 *>(lds + PK_D2 + h * 128);
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    float4 w0 = W[g], w1 = W[16 + g];
                    const float c10 = gauss_act(v1[0][g]), c11 = gauss_act(v1[1][g]);
                    const float dc0 = w0.x * gs3[0] + w0.y * gs3[1] + w0.z * gs3[2];
                    const float dc1 = w1.x * gs3[0] + w1.y * gs3[1] + w1.z * gs3[2];
                    dv1[0][g] = dc0 * dgauss(v1[0][g], c10);
                    dv1[1][g] = dc1 * dgauss(v1[1][g], c11);
                }
            }
            {   // dW_D2[c][k] += sum_s g[c][s] c1[k][s]
                put_rows(gs3, 3);
                v16f t0 = act16(v1[0]), t1 = act16(v1[1]);
                wave_lds_fence();
                float4 gr[3][4];
#pragma unroll
                for (int c = 0; c < 3; ++c)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        gr[c][q] = reinterpret_cast<const float4 *>(scr + c * kScrStride + 16 * h)[q];
                wave_lds_fence();
                scr_put(scr, lane, 0, t0);
                scr_put(scr, lane, 32, t1);
                wave_lds_fence();
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    v16f xo = scr_get(scr, lane, 32 * b);
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        float acc = 0.0f;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            acc = fmaf(gr[c][q].x, xo[4 * q + 0], acc);
                            acc = fmaf(gr[c][q].y, xo[4 * q + 1], acc);
                            acc = fmaf(gr[c][q].z, xo[4 * q + 2], acc);
                            acc = fmaf(gr[c][q].w, xo[4 * q + 3], acc);
                        }
                        gW_D2[b][c] += acc;
                    }
                }
                wave_lds_fence();
            }
            // ---- Directional_MLP.mlp.2 (64 -> 64): weight grads dv1 x c0, then dc0 = W^T dv1
            {
                scr_put(scr, lane, 0, dv1[0]);
                scr_put(scr, lane, 32, dv1[1]);
                wave_lds_fence();
                v16f dy0 = scr_get(scr, lane, 0), dy1 = scr_get(scr, lane, 32);
                wave_lds_fence();
                gB[3][0] += sum16(dy0);
                gB[3][1] += sum16(dy1);
                v16f t0 = act16(v0[0]), t1 = act16(v0[1]);
                scr_put(scr, lane, 0, t0);
                scr_put(scr, lane, 32, t1);
                wave_lds_fence();
                v16f x0 = scr_get(scr, lane, 0), x1 = scr_get(scr, lane, 32);
                wave_lds_fence();
                mma_ws(gW_D1[0][0], dy0, x0);
                mma_ws(gW_D1[0][1], dy0, x1);
                mma_ws(gW_D1[1][0], dy1, x0);
                mma_ws(gW_D1[1][1], dy1, x1);
            }
            v16f dv0[2] = { zero16, zero16 };
            mma_block16(dv0[0], ldt + PKT_D1, 0, lane, dv1[0]);
            mma_block16(dv0[0], ldt + PKT_D1, 4, lane, dv1[1]);
            mma_block16(dv0[1], ldt + PKT_D1, 8, lane, dv1[0]);
            mma_block16(dv0[1], ldt + PKT_D1, 12, lane, dv1[1]);
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                dv0[0][g] *= dgauss(v0[0][g], gauss_act(v0[0][g]));
                dv0[1][g] *= dgauss(v0[1][g], gauss_act(v0[1][g]));
            }
            // ---- Directional_MLP.mlp.0 (48 -> 64): weight grads dv0 x [H1, SH], dH1 = W[:, :32]^T dv0
            {
                scr_put(scr, lane, 0, dv0[0]);
                scr_put(scr, lane, 32, dv0[1]);
                wave_lds_fence();
                v16f dy0 = scr_get(scr, lane, 0), dy1 = scr_get(scr, lane, 32);
                wave_lds_fence();
                float r0 = sum16(dy0), r1 = sum16(dy1);
                gB[2][0] += r0;
                gB[2][1] += r1;
                r0 += __shfl_xor(r0, 32, 64);  // full row sums (both halves of the tile)
                r1 += __shfl_xor(r1, 32, 64);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float shv = h ? sh[8 + j] : sh[j];
                    gW_D0S[0][j] = fmaf(r0, shv, gW_D0S[0][j]);
                    gW_D0S[1][j] = fmaf(r1, shv, gW_D0S[1][j]);
                }
                scr_put(scr, lane, 0, H[1]);
                wave_lds_fence();
                v16f x0 = scr_get(scr, lane, 0);
                wave_lds_fence();
                mma_ws(gW_D0H[0], dy0, x0);
                mma_ws(gW_D0H[1], dy1, x0);
            }
            v16f dH[2] = { zero16, zero16 };
            mma_block16(dH[1], ldt + PKT_D0H, 0, lane, dv0[0]);
            mma_block16(dH[1], ldt + PKT_D0H, 4, lane, dv0[1]);
            // ---- heads (32 -> 1+3+3): dH0 = W^T g ; weight grads on the VALU
            {
                const float4 *W = reinterpret_cast<const float4 *>(lds + PK_HEAD + h * 128);
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    float4 wa = W[2 * g], wb = W[2 * g + 1];
                    dH[0][g] = wa.x * gh[0] + wa.y * gh[1] + wa.z * gh[2] + wa.w * gh[3] + wb.x * gh[4] + wb.y * gh[5] +
                               wb.z * gh[6];
                }
                put_rows(gh, 7);
                wave_lds_fence();
                float4 gr[7][4];
#pragma unroll
                for (int c = 0; c < 7; ++c)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        gr[c][q] = reinterpret_cast<const float4 *>(scr + c * kScrStride + 16 * h)[q];
                wave_lds_fence();
                scr_put(scr, lane, 0, H[0]);
                wave_lds_fence();
                v16f xo = scr_get(scr, lane, 0);
                wave_lds_fence();
#pragma unroll
                for (int c = 0; c < 7; ++c) {
                    float acc = 0.0f;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        acc = fmaf(gr[c][q].x, xo[4 * q + 0], acc);
                        acc = fmaf(gr[c][q].y, xo[4 * q + 1], acc);
                        acc = fmaf(gr[c][q].z, xo[4 * q + 2], acc);
                        acc = fmaf(gr[c][q].w, xo[4 * q + 3], acc);
                    }
                    gW_head[c] += acc;
                }
            }
            // ---- Spatial_MLP.mlp.2 (64 -> 64): weight grads dH x a0, da0 = W^T dH
            {
                scr_put(scr, lane, 0, dH[0]);
                scr_put(scr, lane, 32, dH[1]);
                wave_lds_fence();
                v16f dy0 = scr_get(scr, lane, 0), dy1 = scr_get(scr, lane, 32);
                wave_lds_fence();
                gB[1][0] += sum16(dy0);
                gB[1][1] += sum16(dy1);
                v16f t0 = act16(u0[0]), t1 = act16(u0[1]);
                scr_put(scr, lane, 0, t0);
                scr_put(scr, lane, 32, t1);
                wave_lds_fence();
                v16f x0 = scr_get(scr, lane, 0), x1 = scr_get(scr, lane, 32);
                wave_lds_fence();
                mma_ws(gW_L1[0][0], dy0, x0);
                mma_ws(gW_L1[0][1], dy0, x1);
                mma_ws(gW_L1[1][0], dy1, x0);
                mma_ws(gW_L1[1][1], dy1, x1);
            }
            v16f du0[2] = { zero16, zero16 };
            mma_block16(du0[0], ldt + PKT_L1, 0, lane, dH[0]);
            mma_block16(du0[0], ldt + PKT_L1, 4, lane, dH[1]);
            mma_block16(du0[1], ldt + PKT_L1, 8, lane, dH[0]);
            mma_block16(du0[1], ldt + PKT_L1, 12, lane, dH[1]);
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                du0[0][g] *= dgauss(u0[0][g], gauss_act(u0[0][g]));
                du0[1][g] *= dgauss(u0[1][g], gauss_act(u0[1][g]));
            }
            // ---- Spatial_MLP.mlp.0 (32 -> 64): weight grads du0 x x, dx = W'^T du0
            {
                scr_put(scr, lane, 0, du0[0]);
                scr_put(scr, lane, 32, du0[1]);
                wave_lds_fence();
                v16f dy0 = scr_get(scr, lane, 0), dy1 = scr_get(scr, lane, 32);
                wave_lds_fence();
                gB[0][0] += sum16(dy0);
                gB[0][1] += sum16(dy1);
                scr_put(scr, lane, 0, x);
                wave_lds_fence();
                v16f x0 = scr_get(scr, lane, 0);
                wave_lds_fence();
                mma_ws(gW_L0[0], dy0, x0);
                mma_ws(gW_L0[1], dy1, x0);
            }
            v16f dx = zero16;
            mma_block16(dx, ldt + PKT_L0, 0, lane, du0[0]);
            mma_block16(dx, ldt + PKT_L0, 4, lane, du0[1]);
            // ---- feature gradients, level-major (register 2j+f of half h = level 4(j>>1)+2h+(j&1))
            if (live) {
                const size_t n = (size_t)ray * S + s, NS = (size_t)a.f.B * S;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int level = 4 * (j >> 1) + 2 * h + (j & 1);
                    reinterpret_cast<float2 *>(a.dfeat)[(size_t)level * NS + n] = make_float2(dx[2 * j], dx[2 * j + 1]);
                }
            }
        }
    }

    // ---- flush this wave's weight-gradient partial sums in blob order -------------------------------
    float *out = a.dw_partial + (size_t)wave0 * SCANERF_PARAMSIZE;
    const int k = lane & 31;
    auto put_w = [&](const v16f &acc, int base, int rb, int cb) {
#pragma unroll
        for (int g = 0; g < 16; ++g) out[base + 64 + (32 * cb + k) * 64 + 32 * rb + nmap(g, h)] = acc[g];
    };
    put_w(gW_L0[0], BLOB_S0, 0, 0);
    put_w(gW_L0[1], BLOB_S0, 1, 0);
    put_w(gW_D0H[0], BLOB_D0, 0, 0);
    put_w(gW_D0H[1], BLOB_D0, 1, 0);
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            put_w(gW_L1[rb][cb], BLOB_S1, rb, cb);
            put_w(gW_D1[rb][cb], BLOB_D1, rb, cb);
        }
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int j = 0; j < 8; ++j) out[BLOB_D0 + 64 + (32 + 8 * h + j) * 64 + 32 * rb + k] = gW_D0S[rb][j];
    const int bases[4] = { BLOB_S0, BLOB_S1, BLOB_D0, BLOB_D1 };
#pragma unroll
    for (int l = 0; l < 4; ++l)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            float v = gB[l][rb] + __shfl_xor(gB[l][rb], 32, 64);
            if (h == 0) out[bases[l] + 32 * rb + k] = v;
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

// grad_blob[e] += sum over waves of the partials; first-layer weights carry the folded weight_feature
__global__ void __launch_bounds__(256) k_reduce_dw(const float *__restrict__ partial, int nwaves,
                                                   const float *__restrict__ wf, float *__restrict__ grad_blob)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= SCANERF_PARAMSIZE) return;
    float s = 0.0f;
    for (int w = 0; w < nwaves; ++w) s += partial[(size_t)w * SCANERF_PARAMSIZE + e];
    if (e >= BLOB_S0 + 64 && e < BLOB_S1) s *= wf[(e - 64) / 64];
    grad_blob[e] += s;
}

}  // namespace

// ---------------------------------------------------------------------------- C ABI
SCANERF_API int scanerf_render_backward_grid(int B) { int blocks = ceil_div(B, 4); return blocks > kNumCU ? kNumCU : blocks; }

// dw_partial: [4 * scanerf_render_backward_grid(B)][13994] f32 scratch; grad_blob [13994] is accumulated into.
SCANERF_API int scanerf_render_backward(const float *rays_o, const float *rays_d, const float *z_vals, const float *dists,
                                        const void *features, int feat_dtype, const int32_t *resolutions,
                                        const float *workspace, const float *weight_feature,
                                        const scanerf_render_cfg *cfg, const uint8_t *ray_valid, const float *out_ray,
                                        const float *tile_T, const float *grad_out, float *dfeat, float *dw_partial,
                                        float *grad_blob, int B, int S, int T, scanerf_stream_t stream)
{
    SCANERF_REQUIRE(B >= 0 && S >= 1, "render_backward: B=%d S=%d", B, S);
    SCANERF_REQUIRE(T >= 2 && (T & (T - 1)) == 0, "render_backward: T=%d must be a power of two", T);
    SCANERF_REQUIRE(feat_dtype >= 0 && feat_dtype <= 2, "render_backward: feat_dtype=%d", feat_dtype);
    SCANERF_REQUIRE(cfg, "render_backward: cfg is null");
    if (B == 0) return 0;
    SCANERF_REQUIRE(rays_o && rays_d && z_vals && dists && features && resolutions && workspace && weight_feature &&
                        out_ray && tile_T && grad_out && dfeat && dw_partial && grad_blob,
                    "render_backward: null pointer");
    BwdArgs a;
    a.f.rays_o = rays_o; a.f.rays_d = rays_d; a.f.z_vals = z_vals; a.f.dists = dists;
    a.f.features = features; a.f.resolutions = resolutions; a.f.packed = workspace; a.f.ray_valid = ray_valid;
    a.f.out_ray = const_cast<float *>(out_ray); a.f.weights = nullptr; a.f.tile_T = nullptr;
    a.f.B = B; a.f.S = S; a.f.T = T;
    a.f.contract_mode = cfg->contract_mode; a.f.infinity = cfg->infinity;
    for (int k = 0; k < 3; ++k) {
        a.f.min_bbox[k] = cfg->min_bbox[k];
        a.f.bbox_size[k] = cfg->bbox_size[k];
        a.f.inv_size4[k] = 4.0f / cfg->bbox_size[k];
    }
    a.grad_out = grad_out; a.tile_T = tile_T; a.packed_t = workspace + PK_TOTAL; a.dfeat = dfeat;
    a.dw_partial = dw_partial;
    const int blocks = scanerf_render_backward_grid(B);
    const size_t lds_bytes = (size_t)kBwdLdsFloats * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
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
    if (int e = check_launch("render_backward")) return e;
    hipLaunchKernelGGL(k_reduce_dw, dim3(ceil_div(SCANERF_PARAMSIZE, 256)), dim3(256), 0, st, dw_partial, blocks * 4,
                       weight_feature, grad_blob);
    return check_launch("render_backward(reduce)");
}

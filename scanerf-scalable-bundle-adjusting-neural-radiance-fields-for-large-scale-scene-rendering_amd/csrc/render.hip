// render.hip -- fused per-ray volume rendering, forward (gfx950).
//
// One launch covers hashgrid/__init__.py:512-596 (HashGrid.render_batch_rays):
//   samples = o + z d            -> contract_fore / contract_bg        (:394-411)
//   16-level hash lookup                                                (hashgrid_bg_kernel.cu:107-150)
//   ShallowMLP                                                          (network.py:172-190)
//   cal_integrate_weight + accumulate                                   (:344-366, :564-574)
//
// Structure: persistent 512-thread workgroups, one wave per ray, the ray walked in tiles of
// 32 samples.  In the encode stage lane (s, h) looks up levels {2j+h}: 64 8-byte gathers per
// lane, all independent.  The decoder runs on fp32 MFMA (exact fp32, needed by the Gaussian
// activation's sensitivity) with weights in LDS and activations in registers only
// (render_common.h).  Compositing is a 32-lane prefix product per tile with the
// transmittance carried across tiles.  Per ray 48 B are read and 64 B + S*4 B written;
// everything else is table gathers.
#include <stdlib.h>

#include "render_h3.h"
#include "render_t16.h"
#include "scatter_common.h"

using namespace scanerf;

namespace {

constexpr int kRenderThreads = 512;
constexpr int kLdsFloats = PK_TOTAL + 64;  // + resolutions [16][4] i32

// ------------------------------------------------------------------ pack kernel
__device__ __forceinline__ float blob_w(const float *blob, int base, int n_out, int n, int k)
{
    return blob[base + n_out + k * n_out + n];  // W[n][k] of a layer stored [bias, W^T]
}

__global__ void __launch_bounds__(256) k_pack_decoder(const float *__restrict__ blob, const float *__restrict__ wf,
                                                      float *__restrict__ pk)
{
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < PK_TOTAL; e += gridDim.x * blockDim.x) {
        float v = 0.0f;
        if (e < PK_BIAS) {
            int base, layer, ngrp;
            if (e < PK_L1) { base = PK_L0; layer = 0; ngrp = 4; }
            else if (e < PK_D0H) { base = PK_L1; layer = 1; ngrp = 8; }
            else if (e < PK_D0S) { base = PK_D0H; layer = 2; ngrp = 4; }
            else if (e < PK_D1) { base = PK_D0S; layer = 3; ngrp = 2; }
            else { base = PK_D1; layer = 4; ngrp = 8; }
            const int t = e - base;
            const int gall = t / PK_GRP, rem = t % PK_GRP, slot = rem >> 2, q = rem & 3;
            const int grp = gall % ngrp, blk = gall / ngrp;
            const bool pad = slot == 32 || slot > 64;  // the upper half-wave sits one slot higher
            const int lane = slot < 32 ? slot : slot - 1;
            const int r = grp * 4 + q, h = lane >> 5, n = (lane & 31) + 32 * blk;
            if (pad) {
                v = 0.0f;
            } else if (layer == 0) {
                int k = nmap(r, h);
                v = blob_w(blob, BLOB_S0, 64, n, k) * wf[k];
            } else if (layer == 1) {
                int k = 32 * (r >> 4) + nmap(r & 15, h);
                v = blob_w(blob, BLOB_S1, 64, n, k);
            } else if (layer == 2) {
                v = blob_w(blob, BLOB_D0, 64, n, nmap(r, h));
            } else if (layer == 3) {
                v = blob_w(blob, BLOB_D0, 64, n, 32 + 2 * r + h);
            } else {
                int k = 32 * (r >> 4) + nmap(r & 15, h);
                v = blob_w(blob, BLOB_D1, 64, n, k);
            }
        } else if (e < PK_HEAD) {
            int t = e - PK_BIAS;
            int g = t & 15, h = (t >> 4) & 1, blk = (t >> 5) & 1, layer = t >> 6;
            const int bases[4] = { BLOB_S0, BLOB_S1, BLOB_D0, BLOB_D1 };
            v = blob[bases[layer] + 32 * blk + nmap(g, h)];
        } else if (e < PK_D2) {
            int t = e - PK_HEAD;
            int c = t & 7, g = (t >> 3) & 15, h = t >> 7;
            int n = nmap(g, h);
            if (c == 0) v = blob_w(blob, BLOB_SIG, 1, 0, n);
            else if (c < 4) v = blob_w(blob, BLOB_DIF, 3, c - 1, n);
            else if (c < 7) v = blob_w(blob, BLOB_TINT, 3, c - 4, n);
        } else if (e < PK_HB) {
            int t = e - PK_D2;
            int c = t & 3, g = (t >> 2) & 15, blk = (t >> 6) & 1, h = t >> 7;
            if (c < 3) v = blob_w(blob, BLOB_D2, 3, c, 32 * blk + nmap(g, h));
        } else {
            int t = e - PK_HB;
            if (t == 0) v = blob[BLOB_SIG];
            else if (t < 4) v = blob[BLOB_DIF + t - 1];
            else if (t < 7) v = blob[BLOB_TINT + t - 4];
            else if (t >= 8 && t < 11) v = blob[BLOB_D2 + t - 8];
        }
        pk[e] = v;
    }
}

template <int DT>
__global__ void __launch_bounds__(kRenderThreads, 2) k_render_fwd(RenderArgs a)
{
    __shared__ __attribute__((aligned(16))) float lds[kLdsFloats];
    {
        const float4 *src = reinterpret_cast<const float4 *>(a.packed);
        float4 *dst = reinterpret_cast<float4 *>(lds);
        for (int i = threadIdx.x; i < PK_TOTAL / 4; i += kRenderThreads) dst[i] = src[i];
        int *lres = reinterpret_cast<int *>(lds + PK_TOTAL);
        if (threadIdx.x < 64) {
            int lv = threadIdx.x >> 2, c = threadIdx.x & 3;
            lres[threadIdx.x] = c < 3 ? a.resolutions[3 * lv + c] : 0;
        }
    }
    __syncthreads();
    const int *lds_res = reinterpret_cast<const int *>(lds + PK_TOTAL);
    const int lane = threadIdx.x & 63, sl = lane & 31, h = lane >> 5;
    const int waves_per_block = kRenderThreads / 64;
    const int wave0 = blockIdx.x * waves_per_block + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * waves_per_block;
    const int S = a.S, ntiles = (S + 31) >> 5, nt16 = (S + 15) >> 4;

    for (int ray = wave0; ray < a.B; ray += nwaves) {
        float *outp = a.out_ray + (size_t)ray * SCANERF_RAY_OUT;
        if (a.ray_valid && !a.ray_valid[ray]) {
            // hashgrid/__init__.py:427-431: invalid rays render as zeros with T = 1
            if (lane < 16) outp[lane] = (lane == 4) ? 1.0f : 0.0f;
            if (a.weights)
                for (int s = lane; s < S; s += 64) a.weights[(size_t)ray * S + s] = 0.0f;
            continue;
        }
        float o[3], d[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            o[k] = a.rays_o[3 * ray + k];
            d[k] = a.rays_d[3 * ray + k];
        }
        const float dnorm = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        // Dir layer-0 start: bias + W[:, 32:48] SH(dir), once per ray (2 x 8 MFMA steps)
        v16f dinit[2];
        {
            float sh[16];
            ray_sh(d, dnorm, sh);
            v16f shb;  // step r: half h supplies SH[2r+h]
#pragma unroll
            for (int r = 0; r < 8; ++r) shb[r] = h ? sh[2 * r + 1] : sh[2 * r];
            dinit[0] = load_bias(lds, 2, 0, h);
            dinit[1] = load_bias(lds, 2, 1, h);
            const float *A = lds + PK_D0S + (lane + (lane >> 5)) * 4;
            const float4 a00 = *reinterpret_cast<const float4 *>(A), a01 = *reinterpret_cast<const float4 *>(A + PK_GRP),
                         a10 = *reinterpret_cast<const float4 *>(A + 2 * PK_GRP),
                         a11 = *reinterpret_cast<const float4 *>(A + 3 * PK_GRP);
            MFMA4(dinit[0], a00, shb[0], shb[1], shb[2], shb[3])
            MFMA4(dinit[0], a01, shb[4], shb[5], shb[6], shb[7])
            MFMA4(dinit[1], a10, shb[0], shb[1], shb[2], shb[3])
            MFMA4(dinit[1], a11, shb[4], shb[5], shb[6], shb[7])
        }

        float T_run = 1.0f, T_left = 1.0f;
        float acc[11];
#pragma unroll
        for (int k = 0; k < 11; ++k) acc[k] = 0.0f;

        for (int tile = 0; tile < ntiles; ++tile) {
            const int s = tile * 32 + sl;
            const bool live = s < S;
            const float z = live ? a.z_vals[(size_t)ray * S + s] : 0.0f;
            float delta = live ? a.dists[(size_t)ray * S + s] * dnorm : 0.0f;
            if (a.infinity && s == S - 1) delta = 1e10f;

            float p[3];
            contract_point(a, o, d, z, p);
            v16f x;
            encode8<DT>(a, lds_res, h, p, x);
            if (a.xstash && live) {
                float4 *xs = reinterpret_cast<float4 *>(a.xstash + ((size_t)ray * S + s) * 32 + 16 * h);
                xs[0] = make_float4(x[0], x[1], x[2], x[3]);
                xs[1] = make_float4(x[4], x[5], x[6], x[7]);
                xs[2] = make_float4(x[8], x[9], x[10], x[11]);
                xs[3] = make_float4(x[12], x[13], x[14], x[15]);
                SCANERF_STORE_GUARD();
            }
            SampleOut so = decode_tile(lds, lane, x, dinit);

            // compositing (hashgrid/__init__.py:344-360): alpha, T = cumprod(1 - alpha + 1e-6)
            const float alpha = live ? 1.0f - expf(-so.sigma * delta) : 0.0f;
            float incl = 1.0f - alpha + 1e-6f;
#pragma unroll
            for (int off = 1; off < 32; off <<= 1) {
                float t = __shfl_up(incl, off, 32);
                if (sl >= off) incl *= t;
            }
            float excl = __shfl_up(incl, 1, 32);
            if (sl == 0) excl = 1.0f;
            const float Ti = T_run * excl;
            // transmittance entering each 16-sample tile (samples 32 tile and 32 tile + 16): what the backward kernels start from
            if (a.tile_T && (lane == 0 || (lane == 16 && s < S))) a.tile_T[(size_t)ray * nt16 + 2 * tile + (lane >> 4)] = Ti;
            const float w = alpha * Ti;
            T_run *= __shfl(incl, 31, 32);
            if (tile == ntiles - 1) T_left = __shfl(Ti, (S - 1) & 31, 32);  // T before the last sample (:358-360)
            if (a.weights && live && h == 0) a.weights[(size_t)ray * S + s] = w;
            if (a.weights || a.tile_T) SCANERF_STORE_GUARD();

            acc[0] = fmaf(w, z, acc[0]);
            float s2 = 0.0f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                acc[1 + c] = fmaf(w, so.dif[c], acc[1 + c]);
                acc[4 + c] = fmaf(w, so.tint[c] * so.spec[c], acc[4 + c]);
                acc[7 + c] = fmaf(w, so.tint[c], acc[7 + c]);
                s2 = fmaf(so.spec[c], so.spec[c], s2);
            }
            acc[10] = fmaf(w, s2, acc[10]);
        }
#pragma unroll
        for (int k = 0; k < 11; ++k) acc[k] = half_sum(acc[k]);
        if (lane == 0) {
            float4 *o4 = reinterpret_cast<float4 *>(outp);
            auto clamp01 = [](float v) { return fminf(fmaxf(v, 0.0f), 1.0f); };
            o4[0] = make_float4(clamp01(acc[1] + acc[4]), clamp01(acc[2] + acc[5]), clamp01(acc[3] + acc[6]), acc[0]);
            o4[1] = make_float4(T_left, acc[1], acc[2], acc[3]);
            o4[2] = make_float4(acc[4], acc[5], acc[6], acc[7]);
            o4[3] = make_float4(acc[8], acc[9], acc[10], 0.0f);
        }
    }
}


// ------------------------------------------------------------------ h3 (f16 split) pack + forward
// One thread per (sub-image pair, lane, element): W in f32 -> hi / lo f16 at the two parts.
__global__ void __launch_bounds__(256) k_pack_decoder_h3(const float *__restrict__ blob, const float *__restrict__ wf,
                                                         char *__restrict__ out)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < 32 * 512) {
        const int pair = e >> 9, lane = (e >> 3) & 63, j = e & 7;
        const int r = lane & 31, h = lane >> 5;
        int base, ks, q = pair, layer;
        if (q < 4) { layer = 0; base = H3_L0; ks = 2; }
        else if ((q -= 4) < 8) { layer = 1; base = H3_L1; ks = 4; }
        else if ((q -= 8) < 2) { layer = 2; base = H3_HEAD; ks = 2; }
        else if ((q -= 2) < 6) { layer = 3; base = H3_D0; ks = 3; }
        else if ((q -= 6) < 8) { layer = 4; base = H3_D1; ks = 4; }
        else { q -= 8; layer = 5; base = H3_D2; ks = 4; }
        const int b = q / ks, s = q % ks;
        const int n = 32 * b + r, k = 16 * s + 8 * (j >> 2) + 4 * h + (j & 3);
        float w = 0.0f;
        if (layer == 0) w = blob_w(blob, BLOB_S0, 64, n, k) * wf[k];
        else if (layer == 1) w = blob_w(blob, BLOB_S1, 64, n, k);
        else if (layer == 2) {
            // rows 0-3: sigma, dif xyz; rows 8-10: tint xyz; replicas 4 rows higher (for the upper half-wave)
            const int rr = r & ~4;
            if (r < 16) {
                if (rr == 0) w = blob_w(blob, BLOB_SIG, 1, 0, k);
                else if (rr < 4) w = blob_w(blob, BLOB_DIF, 3, rr - 1, k);
                else if (rr >= 8 && rr < 11) w = blob_w(blob, BLOB_TINT, 3, rr - 8, k);
            }
        } else if (layer == 3) {
            w = s < 2 ? blob_w(blob, BLOB_D0, 64, n, k) : blob_w(blob, BLOB_D0, 64, n, 32 + 8 * h + j);
        } else if (layer == 4) w = blob_w(blob, BLOB_D1, 64, n, k);
        else {
            const int rr = r & ~4;
            if (r < 8 && rr < 3) w = blob_w(blob, BLOB_D2, 3, rr, k);
        }
        const _Float16 hi = (_Float16)w, lo = (_Float16)(w - (float)hi);
        char *p = out + base + ((b * ks + s) * 2) * H3_SUB + r * 16 + h * 576 + j * 2;
        *reinterpret_cast<_Float16 *>(p) = hi;
        *reinterpret_cast<_Float16 *>(p + H3_SUB) = lo;
    } else if (e < 32 * 512 + 64) {  // the 64-byte gaps between the half-waves and after the upper one
        const int sub = e - 32 * 512;  // 64 sub-images: zero the 64-B gap between their half-waves
        float4 *g0 = reinterpret_cast<float4 *>(out + sub * H3_SUB + 512);
        for (int i = 0; i < 4; ++i) g0[i] = make_float4(0, 0, 0, 0);
    } else if (e < 32 * 512 + 64 + 288) {
        const int t = e - 32 * 512 - 64;
        float v = 0.0f;
        if (t < 256) {
            const int g = t & 15, h = (t >> 4) & 1, blk = (t >> 5) & 1, layer = t >> 6;
            const int bases[4] = { BLOB_S0, BLOB_S1, BLOB_D0, BLOB_D1 };
            v = blob[bases[layer] + 32 * blk + nmap(g, h)];
        } else if (t < 272) {
            const int g = t - 256;
            if (g == 0) v = blob[BLOB_SIG];
            else if (g < 4) v = blob[BLOB_DIF + g - 1];
            else if (g < 7) v = blob[BLOB_TINT + g - 4];
        } else {
            const int g = t - 272;
            if (g < 3) v = blob[BLOB_D2 + g];
        }
        reinterpret_cast<float *>(out + H3_BIAS)[t] = v;
    }
}

#ifndef FWD_GATHER_BATCH
#define FWD_GATHER_BATCH 2
#endif
constexpr int kH3LdsBytes = H3_BYTES + 64 * 4;  // + resolutions [16][4] i32

constexpr int kPlanHistWords = 16 * 256;  // scatter bins of the fused count: NB <= 256 buckets per level

// COUNT: also count the backward's scatter records (RenderArgs::plan_counts); JST: also write RenderArgs::jstash
template <int DT, bool COUNT, bool JST = false>
__global__ void __launch_bounds__(kRenderThreads, 2) k_render_fwd_h3(RenderArgs a)
{
    __shared__ __attribute__((aligned(16))) char lds[kH3LdsBytes];
    __shared__ uint32_t hist_store[COUNT ? kPlanHistWords : 1];
    uint32_t *hist = COUNT ? hist_store : nullptr;
    if (COUNT) {
        for (int i = threadIdx.x; i < 16 * a.plan_NB; i += kRenderThreads) hist[i] = 0;
        if (blockIdx.x == 0 && threadIdx.x == 0) {  // as k_bin_count_rays: launch maximum, format word, overflow flag
            *a.plan_maxbits = 0;
            *a.plan_overflow = 0;
            a.plan_overflow[-1] = (uint32_t)a.plan_rec8;  // scatter_common.h format_word()
            a.plan_overflow[-2] = a.skip_levels;          // skip_word(): the levels the counts below leave out
        }
    }
    {
        const float4 *src = reinterpret_cast<const float4 *>(a.packed + PK_TOTAL);
        float4 *dst = reinterpret_cast<float4 *>(lds);
        for (int i = threadIdx.x; i < H3_BYTES / 16; i += kRenderThreads) dst[i] = src[i];
        int *lres = reinterpret_cast<int *>(lds + H3_BYTES);
        if (threadIdx.x < 64) {
            int lv = threadIdx.x >> 2, c = threadIdx.x & 3;
            lres[threadIdx.x] = c < 3 ? a.resolutions[3 * lv + c] : 0;
        }
    }
    __syncthreads();
    const int *lds_res = reinterpret_cast<const int *>(lds + H3_BYTES);
    const int lane = threadIdx.x & 63, sl = lane & 31, h = lane >> 5;
    const int waves_per_block = kRenderThreads / 64;
    const int wave0 = blockIdx.x * waves_per_block + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * waves_per_block;
    const int S = a.S, ntiles = (S + 31) >> 5, nt16 = (S + 15) >> 4;

    for (int ray = wave0; ray < a.B; ray += nwaves) {
        float *outp = a.out_ray + (size_t)ray * SCANERF_RAY_OUT;
        if (a.ray_valid && !a.ray_valid[ray]) {
            if (lane < 16) outp[lane] = (lane == 4) ? 1.0f : 0.0f;
            if (a.weights)
                for (int s = lane; s < S; s += 64) a.weights[(size_t)ray * S + s] = 0.0f;
            continue;
        }
        float o[3], d[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            o[k] = a.rays_o[3 * ray + k];
            d[k] = a.rays_d[3 * ray + k];
        }
        const float dnorm = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        v16f dinit[2];
        {
            float sh[16];
            ray_sh(d, dnorm, sh);
            h3_dinit(lds, lane, sh, dinit);
        }
        float T_run = 1.0f, T_left = 1.0f;
        float acc[11];
#pragma unroll
        for (int k = 0; k < 11; ++k) acc[k] = 0.0f;

        for (int tile = 0; tile < ntiles; ++tile) {
            const int s = tile * 32 + sl;
            const bool live = s < S;
            const float z = live ? a.z_vals[(size_t)ray * S + s] : 0.0f;
            float delta = live ? a.dists[(size_t)ray * S + s] * dnorm : 0.0f;
            if (a.infinity && s == S - 1) delta = 1e10f;
            float p[3];
            contract_point(a, o, d, z, p);
            v16f x;
            if (a.dbg == 2) {
#pragma unroll
                for (int g = 0; g < 16; ++g) x[g] = p[g % 3] * (0.01f * g);
            } else {
                // (dead lanes of the last tile write their own, unused, slots: no branch around the stores)
                uint32_t *jrow = JST ? a.jstash + ((size_t)ray * ntiles + tile) * (8 * 4 * 64) + lane : nullptr;
                encode8<DT, JST ? 1 : FWD_GATHER_BATCH, true, COUNT, JST>(a, lds_res, h, p, x, hist, live, jrow);
            }
            if (a.xstash && live) {  // (plain stores: streaming / nontemporal ones measured 3.16 -> 3.41 ms)
                float4 *xs = reinterpret_cast<float4 *>(a.xstash + ((size_t)ray * S + s) * 32 + 16 * h);
                xs[0] = make_float4(x[0], x[1], x[2], x[3]);
                xs[1] = make_float4(x[4], x[5], x[6], x[7]);
                xs[2] = make_float4(x[8], x[9], x[10], x[11]);
                xs[3] = make_float4(x[12], x[13], x[14], x[15]);
                SCANERF_STORE_GUARD();
            }
            SampleOut so;
            if (a.dbg == 1) {
                so.sigma = x[0] + x[5] + x[10] + x[15];
#pragma unroll
                for (int c = 0; c < 3; ++c) { so.dif[c] = x[1 + c] + x[12 + c]; so.tint[c] = x[4 + c] + x[9 + c]; so.spec[c] = x[7 + c] + x[6 + c]; }
            } else {
                so = decode_tile_h3(lds, lane, x, dinit);
            }

            const float alpha = live ? 1.0f - __builtin_amdgcn_exp2f(-1.4426950408889634f * (so.sigma * delta)) : 0.0f;   // (as the 16-sample-tile backward forms it)
            float incl = 1.0f - alpha + 1e-6f;
#pragma unroll
            for (int off = 1; off < 32; off <<= 1) {
                float t = __shfl_up(incl, off, 32);
                if (sl >= off) incl *= t;
            }
            float excl = __shfl_up(incl, 1, 32);
            if (sl == 0) excl = 1.0f;
            const float Ti = T_run * excl;
            // transmittance entering each 16-sample tile (samples 32 tile and 32 tile + 16): what the backward kernels start from
            if (a.tile_T && (lane == 0 || (lane == 16 && s < S))) a.tile_T[(size_t)ray * nt16 + 2 * tile + (lane >> 4)] = Ti;
            const float w = alpha * Ti;
            T_run *= __shfl(incl, 31, 32);
            if (tile == ntiles - 1) T_left = __shfl(Ti, (S - 1) & 31, 32);
            if (a.weights && live && h == 0) a.weights[(size_t)ray * S + s] = w;
            if (a.weights || a.tile_T) SCANERF_STORE_GUARD();

            acc[0] = fmaf(w, z, acc[0]);
            float s2 = 0.0f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                acc[1 + c] = fmaf(w, so.dif[c], acc[1 + c]);
                acc[4 + c] = fmaf(w, so.tint[c] * so.spec[c], acc[4 + c]);
                acc[7 + c] = fmaf(w, so.tint[c], acc[7 + c]);
                s2 = fmaf(so.spec[c], so.spec[c], s2);
            }
            acc[10] = fmaf(w, s2, acc[10]);
        }
#pragma unroll
        for (int k = 0; k < 11; ++k) acc[k] = half_sum(acc[k]);
        if (lane == 0) {
            float4 *o4 = reinterpret_cast<float4 *>(outp);
            auto clamp01 = [](float v) { return fminf(fmaxf(v, 0.0f), 1.0f); };
            o4[0] = make_float4(clamp01(acc[1] + acc[4]), clamp01(acc[2] + acc[5]), clamp01(acc[3] + acc[6]), acc[0]);
            o4[1] = make_float4(T_left, acc[1], acc[2], acc[3]);
            o4[2] = make_float4(acc[4], acc[5], acc[6], acc[7]);
            o4[3] = make_float4(acc[8], acc[9], acc[10], 0.0f);
        }
    }
    if (COUNT) {
        __syncthreads();
        for (int i = threadIdx.x; i < 16 * a.plan_NB; i += kRenderThreads) a.plan_counts[(size_t)i * a.plan_W + blockIdx.x] = hist[i];
    }
}

}  // namespace

// ---------------------------------------------------------------------------- C ABI
SCANERF_API int scanerf_render_workspace_floats(void) { return WS_FLOATS; }

SCANERF_API int scanerf_pack_decoder(const float *mlp_blob, const float *weight_feature, float *workspace,
                                     scanerf_stream_t stream)
{
    SCANERF_REQUIRE(mlp_blob && weight_feature && workspace, "pack_decoder: null pointer");
    SCANERF_REQUIRE(((uintptr_t)workspace & 15) == 0, "pack_decoder: workspace must be 16-byte aligned");
    hipLaunchKernelGGL(k_pack_decoder, dim3(32), dim3(256), 0, (hipStream_t)stream, mlp_blob, weight_feature,
                       workspace);
    // the same decoder as f16 hi/lo operand pairs for the split-precision kernels (render_h3.h)
    hipLaunchKernelGGL(k_pack_decoder_h3, dim3((32 * 512 + 64 + 288 + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       mlp_blob, weight_feature, reinterpret_cast<char *>(workspace + PK_TOTAL));
    // and as the 16-sample-tile images of the two-waves-per-SIMD backward kernel (render_t16.h)
    launch_pack_decoder_t16(mlp_blob, weight_feature, reinterpret_cast<char *>(workspace + WS_T16), (hipStream_t)stream);
    return check_launch("pack_decoder");
}

static int render_forward(const float *rays_o, const float *rays_d, const float *z_vals, const float *dists,
                          const void *features, int feat_dtype, const int32_t *resolutions, const float *packed,
                          const scanerf_render_cfg *cfg, const uint8_t *ray_valid, float *out_ray, float *weights,
                          float *tile_T, float *xstash, void *jstash, int B, int S, int T, void *scatter_ws,
                          size_t scatter_ws_bytes, scanerf_stream_t stream)
{
    SCANERF_REQUIRE(B >= 0 && S >= 1, "render_forward: B=%d S=%d", B, S);
    SCANERF_REQUIRE(T >= 2 && (T & (T - 1)) == 0, "render_forward: T=%d must be a power of two", T);
    SCANERF_REQUIRE(feat_dtype >= 0 && feat_dtype <= 2, "render_forward: feat_dtype=%d", feat_dtype);
    SCANERF_REQUIRE(cfg, "render_forward: cfg is null");
    if (B == 0) return 0;
    SCANERF_REQUIRE(rays_o && rays_d && z_vals && dists && features && resolutions && packed && out_ray,
                    "render_forward: null pointer");
    SCANERF_REQUIRE(((uintptr_t)packed & 15) == 0 && ((uintptr_t)out_ray & 15) == 0,
                    "render_forward: packed/out_ray must be 16-byte aligned");
    RenderArgs a;
    a.rays_o = rays_o; a.rays_d = rays_d; a.z_vals = z_vals; a.dists = dists;
    a.features = features; a.resolutions = resolutions; a.packed = packed; a.ray_valid = ray_valid;
    a.out_ray = out_ray; a.weights = weights; a.tile_T = tile_T; a.xstash = xstash; a.jstash = static_cast<uint32_t *>(jstash);
    a.B = B; a.S = S; a.T = T;
    a.contract_mode = cfg->contract_mode; a.infinity = cfg->infinity;
    a.skip_levels = tune_set("SCANERF_NO_LEVEL_SKIP") ? 0u : pair_masked_levels(cfg->skip_levels);
    for (int k = 0; k < 3; ++k) {
        a.min_bbox[k] = cfg->min_bbox[k];
        a.bbox_size[k] = cfg->bbox_size[k];
        a.inv_size4[k] = 4.0f / cfg->bbox_size[k];
    }
    a.dbg = tune_int("SCANERF_DEBUG_FWD", 0);
    const int waves_per_block = kRenderThreads / 64;
    int blocks = ceil_div(B, waves_per_block);
    if (blocks > kNumCU) blocks = kNumCU;  // one resident 512-thread workgroup per CU (VGPR-bound), persistent
    { const int v = tune_int("SCANERF_FWD_GRID", 0); if (v >= 1 && v < blocks) blocks = v; }   // (render_bwd.hip scanerf_render_backward_grid)
    dim3 grid(blocks), block(kRenderThreads);
    hipStream_t st = (hipStream_t)stream;
    SCANERF_REQUIRE(cfg->arith >= SCANERF_ARITH_F32 && cfg->arith <= SCANERF_ARITH_T16S, "render_forward: arith=%d", cfg->arith);
    a.plan_counts = nullptr;
    if (scatter_ws) {
        // The t16 backward visits ray (wg + i W) 8 + r; this kernel's wave w of workgroup b visits (b + i grid) 8 + w: the same
        // rays per workgroup when the grids are equal, so this launch can fill the plan's count matrix itself.
        SCANERF_REQUIRE(cfg->arith == SCANERF_ARITH_T16 || cfg->arith == SCANERF_ARITH_T16S, "render_forward_plan: the fused count is the t16 / t16s backward's plan (arith=%d)", cfg->arith);
        SCANERF_REQUIRE(scanerf_render_forward_plan_supported(B, S, T), "render_forward_plan: B=%d S=%d T=%d not supported", B, S, T);
        if (int e = scatter_plan_attach(scatter_ws, scatter_ws_bytes, B, S, T, cfg->arith, blocks, a)) return e;
    }
    if (cfg->arith != SCANERF_ARITH_F32) {  // (H3 and T16 differ in the backward kernel only)
        if (a.jstash) {
            SCANERF_REQUIRE(feat_dtype == SCANERF_F32, "render_forward: the position-Jacobian stash is written from fp32 tables only");
            if (a.plan_counts) hipLaunchKernelGGL((k_render_fwd_h3<SCANERF_F32, true, true>), grid, block, 0, st, a);
            else hipLaunchKernelGGL((k_render_fwd_h3<SCANERF_F32, false, true>), grid, block, 0, st, a);
        } else if (a.plan_counts) {
            if (feat_dtype == SCANERF_F32) hipLaunchKernelGGL((k_render_fwd_h3<SCANERF_F32, true>), grid, block, 0, st, a);
            else if (feat_dtype == SCANERF_F16) hipLaunchKernelGGL((k_render_fwd_h3<SCANERF_F16, true>), grid, block, 0, st, a);
            else hipLaunchKernelGGL((k_render_fwd_h3<SCANERF_BF16, true>), grid, block, 0, st, a);
        } else if (feat_dtype == SCANERF_F32) hipLaunchKernelGGL((k_render_fwd_h3<SCANERF_F32, false>), grid, block, 0, st, a);
        else if (feat_dtype == SCANERF_F16) hipLaunchKernelGGL((k_render_fwd_h3<SCANERF_F16, false>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((k_render_fwd_h3<SCANERF_BF16, false>), grid, block, 0, st, a);
    } else if (feat_dtype == SCANERF_F32) hipLaunchKernelGGL((k_render_fwd<SCANERF_F32>), grid, block, 0, st, a);
    else if (feat_dtype == SCANERF_F16) hipLaunchKernelGGL((k_render_fwd<SCANERF_F16>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((k_render_fwd<SCANERF_BF16>), grid, block, 0, st, a);
    if (int e = check_launch("render_forward")) return e;
    return a.plan_counts ? scatter_plan_finish(scatter_ws, scatter_ws_bytes, B, S, T, cfg->arith, stream) : 0;
}

SCANERF_API int scanerf_render_forward_packed(const float *rays_o, const float *rays_d, const float *z_vals,
                                              const float *dists, const void *features, int feat_dtype,
                                              const int32_t *resolutions, const float *packed,
                                              const scanerf_render_cfg *cfg, const uint8_t *ray_valid,
                                              float *out_ray, float *weights, float *tile_T, float *xstash, int B, int S,
                                              int T, scanerf_stream_t stream)
{
    return render_forward(rays_o, rays_d, z_vals, dists, features, feat_dtype, resolutions, packed, cfg, ray_valid, out_ray,
                          weights, tile_T, xstash, nullptr, B, S, T, nullptr, 0, stream);
}

// The same launch, also doing scanerf_render_scatter_plan's work for the t16 backward of these rays (counts in the forward
// kernel, where the hash indices already are; then the scan): call INSTEAD of scanerf_render_scatter_plan, with that
// function's workspace.  Only where scanerf_render_forward_plan_supported(B, S, T) (equal forward and backward grids).
// jstash (may be NULL; fp32 tables): [B][ceil(S/32)][8][4][64] u32 (render_device.h jst_pack: six 20-bit significands under one exponent per (sample, level)), the encoder's position Jacobians for scanerf_render_backward's g_raypos.
SCANERF_API int scanerf_render_forward_packed_plan(const float *rays_o, const float *rays_d, const float *z_vals,
                                                   const float *dists, const void *features, int feat_dtype,
                                                   const int32_t *resolutions, const float *packed,
                                                   const scanerf_render_cfg *cfg, const uint8_t *ray_valid,
                                                   float *out_ray, float *weights, float *tile_T, float *xstash,
                                                   void *jstash, int B, int S, int T, void *scatter_ws,
                                                   size_t scatter_ws_bytes, scanerf_stream_t stream)
{
    // (scatter_ws may be NULL: the plain forward with the jstash output)
    return render_forward(rays_o, rays_d, z_vals, dists, features, feat_dtype, resolutions, packed, cfg, ray_valid, out_ray,
                          weights, tile_T, xstash, jstash, B, S, T, scatter_ws, scatter_ws_bytes, stream);
}

SCANERF_API int scanerf_render_forward_plan_supported(int B, int S, int T)
{
    if (scanerf_render_scatter_workspace_bytes(B, S, T) == 0) return 0;
    int fwd_grid = ceil_div(B, kRenderThreads / 64) > kNumCU ? kNumCU : ceil_div(B, kRenderThreads / 64);
    { const int v = tune_int("SCANERF_FWD_GRID", 0); if (v >= 1 && v < fwd_grid) fwd_grid = v; }
    if (fwd_grid != scanerf_render_backward_grid(B)) return 0;
    return 1;  // (NB <= 256 buckets per level whatever T: scatter_common.h fused_bucket_log)
}

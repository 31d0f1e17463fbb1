// hashgrid.hip -- stand-alone multi-resolution hash encoder, forward and backward (gfx950).
//
// Reference behaviour: hashgrid/src/hashgrid_bg_kernel.cu:107-275 (contracted space) and
// hashgrid/src/hashgrid_kernel.cu:106-300 (world-space box).
//
// Layout in HBM: features [L][T][2] (8 B entries, fp32; 4 B for f16/bf16), one 4 MiB slice
// per level at T=2^19 -- exactly one XCD L2.  Two forward mappings are built:
//
//   XCD-partitioned (default for N*L large): block b runs on XCD b%8 (round-robin dispatch;
//     a placement guess that only affects speed) and works on levels l = b%8 (mod 8) one
//     after the other, so the gathers of the 32 CUs of an XCD share one table slice in
//     their L2 instead of thrashing all 16 slices.  Output layout is either the binding
//     surface's [N][L][2] or level-major [L][N][2] (coalesced 512-B stores; used by the
//     two-kernel render path).
//   level-fastest: thread t -> (point t/L, level t%L).  Stores to [N][L][2] are fully
//     coalesced; used for small launches.
//
// Backward: level-fastest mapping.  grad_in reads are coalesced, the point gradient is
// reduced over the L lanes of a point with shuffles (no atomics; the reference issues 3
// atomics per (point, level)), table gradients use hardware fp32 atomics.
#include "hashgrid_common.h"

using namespace scanerf;

namespace {

enum { OUT_NLF = 0, OUT_LNF = 1 };

struct BoxArgs {
    const float *corner;  // device [3] or nullptr for the contracted-space variant
    const float *size;
};

template <bool BOX>
__device__ __forceinline__ void locate3(const float p[3], const int res[3], const float bc[3], const float bs[3],
                                        int b[3], float t[3], float sc[3])
{
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (BOX) locate_box(p[k], res[k], bc[k], bs[k], b[k], t[k], sc[k]);
        else locate_bg(p[k], res[k], b[k], t[k], sc[k]);
    }
}

template <int DT>
__device__ __forceinline__ float2 interp(const void *slice, const int b[3], const float t[3], uint32_t mask)
{
    uint32_t idx[8];
    float w[8];
    corner_indices(idx, b[0], b[1], b[2], mask);
    trilinear_weights(w, t[0], t[1], t[2]);
    float2 f[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) f[c] = TableElem<DT>::load(slice, idx[c]);
    float ax = 0.0f, ay = 0.0f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        ax = fmaf(w[c], f[c].x, ax);
        ay = fmaf(w[c], f[c].y, ay);
    }
    return make_float2(ax, ay);
}

// ---- forward, XCD-partitioned by level -------------------------------------------
template <int DT, bool BOX, int LAYOUT>
__global__ void __launch_bounds__(256) k_embed_fwd_xcd(const float *__restrict__ points, float2 *__restrict__ out,
                                                       const void *__restrict__ features,
                                                       const int32_t *__restrict__ resolutions, BoxArgs box, int N,
                                                       int L, int T, int chunks)
{
    const int xcd = blockIdx.x & (kNumXCD - 1);
    const int j0 = blockIdx.x >> 3, nb = gridDim.x >> 3;
    const int my_levels = (L - xcd + kNumXCD - 1) / kNumXCD;  // levels xcd, xcd+8, ...
    const uint32_t mask = (uint32_t)T - 1u;
    float bc[3] = { 0, 0, 0 }, bs[3] = { 1, 1, 1 };
    if (BOX) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { bc[k] = box.corner[k]; bs[k] = box.size[k]; }
    }
    for (int w = j0; w < my_levels * chunks; w += nb) {
        const int level = xcd + kNumXCD * (w / chunks);
        const int i = (w % chunks) * 256 + threadIdx.x;
        if (i >= N) continue;
        const int res[3] = { resolutions[3 * level], resolutions[3 * level + 1], resolutions[3 * level + 2] };
        const float p[3] = { points[3 * i], points[3 * i + 1], points[3 * i + 2] };
        int b[3];
        float t[3], sc[3];
        locate3<BOX>(p, res, bc, bs, b, t, sc);
        const char *slice = (const char *)features + (size_t)level * T * TableElem<DT>::bytes;
        float2 r = interp<DT>(slice, b, t, mask);
        if (LAYOUT == OUT_NLF) out[(size_t)i * L + level] = r;
        else out[(size_t)level * N + i] = r;
    }
}

// ---- forward, level-fastest ---------------------------------------------------------
template <int DT, bool BOX>
__global__ void __launch_bounds__(256) k_embed_fwd_lf(const float *__restrict__ points, float2 *__restrict__ out,
                                                      const void *__restrict__ features,
                                                      const int32_t *__restrict__ resolutions, BoxArgs box, int N,
                                                      int L, int T)
{
    const uint32_t mask = (uint32_t)T - 1u;
    float bc[3] = { 0, 0, 0 }, bs[3] = { 1, 1, 1 };
    if (BOX) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { bc[k] = box.corner[k]; bs[k] = box.size[k]; }
    }
    const int64_t total = (int64_t)N * L;
    for (int64_t tt = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; tt < total;
         tt += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(tt / L), level = (int)(tt % L);
        const int res[3] = { resolutions[3 * level], resolutions[3 * level + 1], resolutions[3 * level + 2] };
        const float p[3] = { points[3 * i], points[3 * i + 1], points[3 * i + 2] };
        int b[3];
        float t[3], sc[3];
        locate3<BOX>(p, res, bc, bs, b, t, sc);
        const char *slice = (const char *)features + (size_t)level * T * TableElem<DT>::bytes;
        out[tt] = interp<DT>(slice, b, t, mask);
    }
}

// ---- forward, row mapping (round 5): what a render batch's sample points want ------------------------------------------
// One wave per 32 CONSECUTIVE points; lane (s, h) = point s, levels 8h .. 8h+7: the 64 gathers of a lane are independent
// (issued four levels at a time), consecutive points -- samples along one ray -- share cells at the coarse levels and
// lines at the fine ones (the fused forward's locality, render.hip), and a lane's 8 results are 64 CONTIGUOUS bytes of the
// binding surface's [N][16][2] output: the wave writes 4 KB in full lines.  The XCD-partitioned kernel above writes that
// layout as one 8-byte piece per (point, level) from eight different L2s -- 1.3e8 partial-line writes for 8.4e6 points, and
// its threads hold 8 gathers each: 6.0 ms for a 65 536 x 128 batch against 3.2 ms here.  Same arithmetic per (point, level)
// (locate_bg, corner_indices, trilinear_weights, the fmaf chain of interp): the same bits.
constexpr int kRowsThreads = 512;
template <int DT>
__global__ void __launch_bounds__(kRowsThreads, 2) k_embed_fwd_rows(const float *__restrict__ points, float *__restrict__ out,
                                                                   const void *__restrict__ features,
                                                                   const int32_t *__restrict__ resolutions, int N, int T)
{
    __shared__ int lres[16 * 4];
    if (threadIdx.x < 64) {
        const int lv = threadIdx.x >> 2, c = threadIdx.x & 3;
        lres[threadIdx.x] = c < 3 ? resolutions[3 * lv + c] : 0;
    }
    __syncthreads();
    const uint32_t mask = (uint32_t)T - 1u;
    const int lane = threadIdx.x & 63, s = lane & 31, h = lane >> 5;
    const int ntiles = (N + 31) >> 5;
    const int stride = gridDim.x * (kRowsThreads / 64);
    for (int tile = blockIdx.x * (kRowsThreads / 64) + (threadIdx.x >> 6); tile < ntiles; tile += stride) {
        const int i = tile * 32 + s;
        const bool live = i < N;
        const int ic = live ? i : N - 1;
        const float p[3] = { points[3 * (size_t)ic], points[3 * (size_t)ic + 1], points[3 * (size_t)ic + 2] };
        float2 r[8];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            uint32_t idx[4][8];
            float w[4][8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int level = 8 * h + 4 * half + j;
                const int4 rr = *reinterpret_cast<const int4 *>(lres + 4 * level);
                const int res[3] = { rr.x, rr.y, rr.z };
                int b[3];
                float t[3], sc[3];
                locate3<false>(p, res, nullptr, nullptr, b, t, sc);
                corner_indices(idx[j], b[0], b[1], b[2], mask);
                trilinear_weights(w[j], t[0], t[1], t[2]);
            }
            float2 f[4][8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const char *slice = (const char *)features + (size_t)(8 * h + 4 * half + j) * T * TableElem<DT>::bytes;
#pragma unroll
                for (int c = 0; c < 8; ++c) f[j][c] = TableElem<DT>::load(slice, idx[j][c]);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float ax = 0.0f, ay = 0.0f;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    ax = fmaf(w[j][c], f[j][c].x, ax);
                    ay = fmaf(w[j][c], f[j][c].y, ay);
                }
                r[4 * half + j] = make_float2(ax, ay);
            }
        }
        if (live) {
            float4 *o = reinterpret_cast<float4 *>(out + (size_t)i * 32 + 16 * h);
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = make_float4(r[2 * q].x, r[2 * q].y, r[2 * q + 1].x, r[2 * q + 1].y);
        }
    }
}

// ---- backward, level-fastest ----------------------------------------------------------
// LP = L rounded up to a power of two (<= 64): the lanes of one point form an aligned group.
template <bool BOX, int LP, bool GRAD_LM = false>
__global__ void __launch_bounds__(256) k_embed_bwd(const float *__restrict__ points, const float2 *__restrict__ grad_in,
                                                   float *__restrict__ grad_points, float *__restrict__ grad_features,
                                                   const float *__restrict__ features,
                                                   const int32_t *__restrict__ resolutions, BoxArgs box, int N, int L,
                                                   int T)
{
    const uint32_t mask = (uint32_t)T - 1u;
    float bc[3] = { 0, 0, 0 }, bs[3] = { 1, 1, 1 };
    if (BOX) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { bc[k] = box.corner[k]; bs[k] = box.size[k]; }
    }
    const int64_t total = (int64_t)N * LP;
    // trip count is uniform per group of LP lanes (tt/LP is shared), shuffles stay inside the group
    for (int64_t tt = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; tt < ((total + 255) / 256) * 256;
         tt += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(tt / LP), level = (int)(tt % LP);
        const bool live = (i < N) && (level < L);
        float gpx = 0.0f, gpy = 0.0f, gpz = 0.0f;
        if (live) {
            const int res[3] = { resolutions[3 * level], resolutions[3 * level + 1], resolutions[3 * level + 2] };
            const float p[3] = { points[3 * i], points[3 * i + 1], points[3 * i + 2] };
            int b[3];
            float t[3], sc[3];
            locate3<BOX>(p, res, bc, bs, b, t, sc);
            uint32_t idx[8];
            float w[8];
            corner_indices(idx, b[0], b[1], b[2], mask);
            trilinear_weights(w, t[0], t[1], t[2]);
            const float2 g = GRAD_LM ? grad_in[(size_t)level * N + i] : grad_in[(size_t)i * L + level];
            if (grad_features) {
                float *gs = grad_features + (size_t)level * T * 2;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    unsafeAtomicAdd(gs + 2 * idx[c], w[c] * g.x);
                    unsafeAtomicAdd(gs + 2 * idx[c] + 1, w[c] * g.y);
                }
            }
            if (grad_points) {
                const float2 *slice = reinterpret_cast<const float2 *>(features) + (size_t)level * T;
                const float tx = t[0], ty = t[1], tz = t[2], ax = 1 - tx, ay = 1 - ty, az = 1 - tz;
                // d(out)/d(offset) = sum_c f_c * dw_c/d(offset); g-contracted first: s_c = <g, f_c>
                float s[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    float2 f = slice[idx[c]];
                    s[c] = g.x * f.x + g.y * f.y;
                }
                float dx = -(ay * az) * s[0] - (ay * tz) * s[1] - (ty * az) * s[2] - (ty * tz) * s[3] +
                           (ay * az) * s[4] + (ay * tz) * s[5] + (ty * az) * s[6] + (ty * tz) * s[7];
                float dy = -(ax * az) * s[0] - (ax * tz) * s[1] + (ax * az) * s[2] + (ax * tz) * s[3] -
                           (tx * az) * s[4] - (tx * tz) * s[5] + (tx * az) * s[6] + (tx * tz) * s[7];
                float dz = -(ax * ay) * s[0] + (ax * ay) * s[1] - (ax * ty) * s[2] + (ax * ty) * s[3] -
                           (tx * ay) * s[4] + (tx * ay) * s[5] - (tx * ty) * s[6] + (tx * ty) * s[7];
                gpx = sc[0] * dx;
                gpy = sc[1] * dy;
                gpz = sc[2] * dz;
            }
        }
        if (grad_points) {
#pragma unroll
            for (int off = LP / 2; off > 0; off >>= 1) {
                gpx += __shfl_xor(gpx, off, 64);
                gpy += __shfl_xor(gpy, off, 64);
                gpz += __shfl_xor(gpz, off, 64);
            }
            if (level == 0 && i < N) {
                grad_points[3 * i + 0] += gpx;
                grad_points[3 * i + 1] += gpy;
                grad_points[3 * i + 2] += gpz;
            }
        }
    }
}

template <bool BOX>
int launch_fwd(const float *points, float *outputs, const void *features, const int32_t *resolutions, BoxArgs box,
               int N, int L, int T, int dt, int variant, int layout, hipStream_t st)
{
    float2 *out = reinterpret_cast<float2 *>(outputs);
    // variant: 0 auto, 1 XCD-partitioned, 2 level-fastest, 3 row mapping (contracted variant, 16 levels, [N][L][2] output)
    const bool rows_ok = !BOX && L == 16 && layout == OUT_NLF && ((uintptr_t)outputs & 15) == 0;
    if (variant == 0) variant = ((int64_t)N * L >= (1 << 20)) ? (rows_ok ? 3 : 1) : 2;
    if (variant == 3 && !rows_ok) variant = 1;
    if (layout == OUT_LNF) variant = 1;
    if constexpr (!BOX) {
        if (variant == 3) {
            int blocks = ceil_div(ceil_div(N, 32), kRowsThreads / 64);
            if (blocks > 2 * kNumCU) blocks = 2 * kNumCU;   // persistent
            dim3 grid(blocks), block(kRowsThreads);
            if (dt == SCANERF_F32) hipLaunchKernelGGL((k_embed_fwd_rows<SCANERF_F32>), grid, block, 0, st, points, outputs, features, resolutions, N, T);
            else if (dt == SCANERF_F16) hipLaunchKernelGGL((k_embed_fwd_rows<SCANERF_F16>), grid, block, 0, st, points, outputs, features, resolutions, N, T);
            else hipLaunchKernelGGL((k_embed_fwd_rows<SCANERF_BF16>), grid, block, 0, st, points, outputs, features, resolutions, N, T);
            return check_launch("embedding_forward(rows)");
        }
    }
    if (variant == 1) {
        const int chunks = ceil_div(N, 256);
        int64_t want = (int64_t)chunks * ((L + kNumXCD - 1) / kNumXCD);  // work items per XCD
        int nb = (int)(want < 32 * 16 ? want : 32 * 16);                 // blocks per XCD (32 CUs x 16)
        if (nb < 1) nb = 1;
        dim3 grid(nb * kNumXCD), block(256);
#define SCANERF_FWD_XCD(DT)                                                                                        \
    if (layout == OUT_NLF)                                                                                         \
        hipLaunchKernelGGL((k_embed_fwd_xcd<DT, BOX, OUT_NLF>), grid, block, 0, st, points, out, features,          \
                           resolutions, box, N, L, T, chunks);                                                     \
    else                                                                                                           \
        hipLaunchKernelGGL((k_embed_fwd_xcd<DT, BOX, OUT_LNF>), grid, block, 0, st, points, out, features,          \
                           resolutions, box, N, L, T, chunks);
        if (dt == SCANERF_F32) { SCANERF_FWD_XCD(SCANERF_F32) }
        else if (dt == SCANERF_F16) { SCANERF_FWD_XCD(SCANERF_F16) }
        else { SCANERF_FWD_XCD(SCANERF_BF16) }
#undef SCANERF_FWD_XCD
    } else {
        dim3 grid(stream_grid((int64_t)N * L, 256)), block(256);
        if (dt == SCANERF_F32)
            hipLaunchKernelGGL((k_embed_fwd_lf<SCANERF_F32, BOX>), grid, block, 0, st, points, out, features,
                               resolutions, box, N, L, T);
        else if (dt == SCANERF_F16)
            hipLaunchKernelGGL((k_embed_fwd_lf<SCANERF_F16, BOX>), grid, block, 0, st, points, out, features,
                               resolutions, box, N, L, T);
        else
            hipLaunchKernelGGL((k_embed_fwd_lf<SCANERF_BF16, BOX>), grid, block, 0, st, points, out, features,
                               resolutions, box, N, L, T);
    }
    return check_launch("embedding_forward");
}

template <bool BOX>
int launch_bwd(const float *points, const float *grad_in, float *grad_points, float *grad_features,
               const float *features, const int32_t *resolutions, BoxArgs box, int N, int L, int T, hipStream_t st)
{
    int LP = 1;
    while (LP < L) LP <<= 1;
    dim3 grid(stream_grid((int64_t)N * LP, 256)), block(256);
    const float2 *g = reinterpret_cast<const float2 *>(grad_in);
#define SCANERF_BWD(LPV)                                                                                          \
    hipLaunchKernelGGL((k_embed_bwd<BOX, LPV>), grid, block, 0, st, points, g, grad_points, grad_features, features, \
                       resolutions, box, N, L, T)
    switch (LP) {
    case 1: SCANERF_BWD(1); break;
    case 2: SCANERF_BWD(2); break;
    case 4: SCANERF_BWD(4); break;
    case 8: SCANERF_BWD(8); break;
    case 16: SCANERF_BWD(16); break;
    case 32: SCANERF_BWD(32); break;
    default: SCANERF_BWD(64); break;
    }
#undef SCANERF_BWD
    return check_launch("embedding_backward");
}

int check_common(const char *what, int N, int L, int T)
{
    SCANERF_REQUIRE(N >= 0 && L >= 1 && L <= 64, "%s: N=%d L=%d (need 1<=L<=64)", what, N, L);
    SCANERF_REQUIRE(T >= 2 && (T & (T - 1)) == 0, "%s: T=%d must be a power of two", what, T);
    return 0;
}

}  // namespace

// ---------------------------------------------------------------------------- C ABI
SCANERF_API int scanerf_embedding_bg_forward_ex(const float *points, float *outputs, const void *features,
                                                const int32_t *resolutions, int N, int L, int T, int feat_dtype,
                                                int variant, int level_major_out, scanerf_stream_t stream)
{
    if (int e = check_common("embedding_bg_forward", N, L, T)) return e;
    SCANERF_REQUIRE(feat_dtype >= 0 && feat_dtype <= 2, "embedding_bg_forward: feat_dtype=%d", feat_dtype);
    if (N == 0) return 0;
    SCANERF_REQUIRE(points && outputs && features && resolutions, "embedding_bg_forward: null pointer");
    return launch_fwd<false>(points, outputs, features, resolutions, BoxArgs{ nullptr, nullptr }, N, L, T, feat_dtype,
                             variant, level_major_out ? OUT_LNF : OUT_NLF, (hipStream_t)stream);
}

SCANERF_API int scanerf_embedding_bg_forward(const float *points, float *outputs, const void *features,
                                             const int32_t *resolutions, int N, int L, int T, int feat_dtype,
                                             scanerf_stream_t stream)
{
    return scanerf_embedding_bg_forward_ex(points, outputs, features, resolutions, N, L, T, feat_dtype, 0, 0, stream);
}

SCANERF_API int scanerf_embedding_bg_backward(const float *points, const float *grad_in, float *grad_points,
                                              float *grad_features, const float *features,
                                              const int32_t *resolutions, int N, int L, int T,
                                              scanerf_stream_t stream)
{
    if (int e = check_common("embedding_bg_backward", N, L, T)) return e;
    if (N == 0) return 0;
    SCANERF_REQUIRE(points && grad_in && features && resolutions, "embedding_bg_backward: null pointer");
    return launch_bwd<false>(points, grad_in, grad_points, grad_features, features, resolutions,
                             BoxArgs{ nullptr, nullptr }, N, L, T, (hipStream_t)stream);
}

// dL/d(points) only, from a LEVEL-MAJOR feature gradient [L][N][2] (the fused backward's dfeat).
SCANERF_API int scanerf_embedding_bg_point_grad(const float *points, const float *dfeat_level_major, float *grad_points,
                                                const float *features, const int32_t *resolutions, int N, int L, int T,
                                                scanerf_stream_t stream)
{
    if (int e = check_common("embedding_bg_point_grad", N, L, T)) return e;
    if (N == 0) return 0;
    SCANERF_REQUIRE(points && dfeat_level_major && grad_points && features && resolutions, "embedding_bg_point_grad: null pointer");
    SCANERF_REQUIRE(L == 16, "embedding_bg_point_grad: L=%d (the fused path has 16 levels)", L);
    dim3 grid(stream_grid((int64_t)N * 16, 256)), block(256);
    hipLaunchKernelGGL((k_embed_bwd<false, 16, true>), grid, block, 0, (hipStream_t)stream, points,
                       reinterpret_cast<const float2 *>(dfeat_level_major), grad_points, (float *)nullptr, features,
                       resolutions, BoxArgs{ nullptr, nullptr }, N, L, T);
    return check_launch("embedding_bg_point_grad");
}

SCANERF_API int scanerf_embedding_forward(const float *points, float *outputs, const float *features,
                                          const float *block_corner, const float *block_size,
                                          const int32_t *resolutions, int N, int L, int T, scanerf_stream_t stream)
{
    if (int e = check_common("embedding_forward", N, L, T)) return e;
    if (N == 0) return 0;
    SCANERF_REQUIRE(points && outputs && features && resolutions && block_corner && block_size,
                    "embedding_forward: null pointer");
    return launch_fwd<true>(points, outputs, features, resolutions, BoxArgs{ block_corner, block_size }, N, L, T,
                            SCANERF_F32, 0, OUT_NLF, (hipStream_t)stream);
}

SCANERF_API int scanerf_embedding_backward(const float *points, const float *grad_in, float *grad_points,
                                           float *grad_features, const float *features, const float *block_corner,
                                           const float *block_size, const int32_t *resolutions, int N, int L, int T,
                                           scanerf_stream_t stream)
{
    if (int e = check_common("embedding_backward", N, L, T)) return e;
    if (N == 0) return 0;
    SCANERF_REQUIRE(points && grad_in && features && resolutions && block_corner && block_size,
                    "embedding_backward: null pointer");
    return launch_bwd<true>(points, grad_in, grad_points, grad_features, features, resolutions,
                            BoxArgs{ block_corner, block_size }, N, L, T, (hipStream_t)stream);
}

// rays.hip -- ray generation, box clipping and the occupancy-grid samplers (gfx950).
//
// Built with -ffp-contract=off: every mul/add pair is two IEEE roundings, the same
// arithmetic as the C oracle, so z_vals / dists / bounds compare bit for bit.
//
// Reference behaviour (file:line under the reference repo):
//   compute_ray_*          cuda/compute_ray_kernel.cu:18-136, cuda/include/cuda_utils.h:143-155
//   ray_aabb_intersection  cuda/helper_kernel.cu:108-197,  cuda_utils.h:564-613
//   sample_points_grid     cuda/helper_kernel.cu:540-671,  cuda/include/dda.h:206-268
//   samplers               cuda/sample_kernel.cu:18-126,   cuda_utils.h:61-113
#include "dda_device.h"

using namespace scanerf;

namespace {

// ------------------------------------------------------------------ compute_ray
__global__ void __launch_bounds__(256) k_compute_ray_fwd(float *__restrict__ rays_o, float *__restrict__ rays_d,
                                                         const float *__restrict__ Ks, const float *__restrict__ C2Ws,
                                                         const int32_t *__restrict__ locs, int B)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B; i += gridDim.x * blockDim.x) {
        int v = locs[3 * i], px = locs[3 * i + 1], py = locs[3 * i + 2];
        const float *K = Ks + 9 * v, *M = C2Ws + 12 * v;
        float x = (1.0f * px + 0.5f - K[2]) / K[0];
        float y = (1.0f * py + 0.5f - K[5]) / K[4];
        rays_d[3 * i + 0] = M[0] * x + M[1] * y + M[2];
        rays_d[3 * i + 1] = M[4] * x + M[5] * y + M[6];
        rays_d[3 * i + 2] = M[8] * x + M[9] * y + M[10];
        rays_o[3 * i + 0] = M[3];
        rays_o[3 * i + 1] = M[7];
        rays_o[3 * i + 2] = M[11];
    }
}

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Adjoint of k_compute_ray_fwd w.r.t. C2W, summed in a FIXED order (round 6; the reference issues 12 float atomics per ray into
// <= 400 rows, compute_ray_kernel.cu:46-92, and rounds 1-5 here issued 12 per (wave, view): the one launch-to-launch difference
// left in a training step).  One workgroup per camera: thread t adds up the camera's rays t, t+1024, ... in index order, the
// 16 waves reduce by the same butterfly every launch, wave sums are added 0..15 by one thread, which alone writes the camera's
// row (+=, the binding's accumulate-into-the-caller's-buffer contract).  Every workgroup reads the view column of locs once
// (B * 12 bytes from the L2s per camera: 0.3 GB at 65 536 rays x 400 cameras); only the owning camera's rays load gradients.
constexpr int kRayBwdThreads = 1024;
__global__ void __launch_bounds__(kRayBwdThreads) k_compute_ray_bwd(const float *__restrict__ g_o, const float *__restrict__ g_d,
                                                                    const float *__restrict__ Ks, float *__restrict__ grad_C2Ws,
                                                                    const int32_t *__restrict__ locs, int B, int num_cam)
{
    __shared__ float part[kRayBwdThreads / 64][12];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int v = blockIdx.x; v < num_cam; v += gridDim.x) {
        const float *K = Ks + 9 * v;
        const float k0 = K[0], k2 = K[2], k4 = K[4], k5 = K[5];
        float c[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) c[k] = 0.0f;
        for (int i = threadIdx.x; i < B; i += kRayBwdThreads) {
            if (locs[3 * i] != v) continue;
            int px = locs[3 * i + 1], py = locs[3 * i + 2];
            float x = (1.0f * px + 0.5f - k2) / k0;
            float y = (1.0f * py + 0.5f - k5) / k4;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                float gd = g_d[3 * i + r];
                c[4 * r + 0] += gd * x;
                c[4 * r + 1] += gd * y;
                c[4 * r + 2] += gd;
                c[4 * r + 3] += g_o[3 * i + r];
            }
        }
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            float s = wave_sum(c[k]);
            if (lane == 0) part[wave][k] = s;
        }
        __syncthreads();
        if (threadIdx.x < 12) {
            float s = 0.0f;
#pragma unroll
            for (int w = 0; w < kRayBwdThreads / 64; ++w) s += part[w][threadIdx.x];
            grad_C2Ws[12 * v + threadIdx.x] += s;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------ ray_aabb
__global__ void __launch_bounds__(256) k_ray_aabb(const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                                                  const float *__restrict__ center, const float *__restrict__ size,
                                                  float *__restrict__ bounds, int B, int K)
{
    int64_t total = (int64_t)B * K;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        int i = (int)(t / K), k = (int)(t % K);
        float o[3] = { rays_o[3 * i], rays_o[3 * i + 1], rays_o[3 * i + 2] };
        float d[3] = { rays_d[3 * i], rays_d[3 * i + 1], rays_d[3 * i + 2] };
        float c[3] = { center[3 * k], center[3 * k + 1], center[3 * k + 2] };
        float h[3] = { size[3 * k] / 2.0f, size[3 * k + 1] / 2.0f, size[3 * k + 2] / 2.0f };
        F2 r = clip_box(o, d, c, h);
        reinterpret_cast<float2 *>(bounds)[t] = make_float2(r.x, r.y);
    }
}

// One lane per ray.  Two DDA walks over the byte occupancy grid: walk 1 measures the occupied
// length, walk 2 apportions exactly S samples over the occupied segments.  Rows are written
// by the owning lane; a row's lines stay in L2 until complete, so HBM sees them once.
__global__ void __launch_bounds__(256) k_sample_points_grid(const float *__restrict__ rays_o,
                                                            const float *__restrict__ rays_d,
                                                            float *__restrict__ z_vals, float *__restrict__ dists,
                                                            const float *__restrict__ corner_p,
                                                            const float *__restrict__ size_p,
                                                            const uint8_t *__restrict__ occ,
                                                            const int32_t *__restrict__ log2dim, int B, int S)
{
    const int ly = log2dim[1], lz = log2dim[2];
    const int side[3] = { 1 << log2dim[0], 1 << ly, 1 << lz };
    const float corner[3] = { corner_p[0], corner_p[1], corner_p[2] };
    const float size[3] = { size_p[0], size_p[1], size_p[2] };
    float half[3], ctr[3], cs[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        half[k] = size[k] / 2.0f;
        ctr[k] = corner[k] + half[k];
        cs[k] = size[k] / (float)side[k];
    }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B; i += gridDim.x * blockDim.x) {
        float o[3] = { rays_o[3 * i], rays_o[3 * i + 1], rays_o[3 * i + 2] };
        float d[3] = { rays_d[3 * i], rays_d[3 * i + 1], rays_d[3 * i + 2] };
        F2 span = clip_box(o, d, ctr, half);
        if (span.x == -1.0f) continue;
        float og[3] = { o[0] - corner[0], o[1] - corner[1], o[2] - corner[2] };

        Walker w;
        w.start(og, d, span, side, cs);
        float total = 0.0f;
        int count = 0;
        while (!w.done()) {
            w.pick();
            uint32_t n = ((uint32_t)w.cell[0] << (ly + lz)) | ((uint32_t)w.cell[1] << lz) | (uint32_t)w.cell[2];
            if (occ[n]) {
                float len = w.t1 - w.t0;
                if (len > 0) { total += len; ++count; }
            }
            w.advance();
        }
        if (count == 0) continue;

        w.start(og, d, span, side, cs);
        int left = S, seg = 0;
        float *zrow = z_vals + (size_t)i * S, *drow = dists + (size_t)i * S;
        while (!w.done()) {
            w.pick();
            uint32_t n = ((uint32_t)w.cell[0] << (ly + lz)) | ((uint32_t)w.cell[1] << lz) | (uint32_t)w.cell[2];
            if (occ[n]) {
                float len = w.t1 - w.t0;
                if (len > 0) {
                    int num = (int)((float)S * len / total);
                    num = num < 1 ? 1 : num;
                    num = num > left ? left : num;
                    if (seg == count - 1) num = left;
                    float interval = (w.t1 - w.t0) / (float)num;
                    int at = S - left;
                    for (int k = 0; k < num; ++k) {
                        zrow[at + k] = w.t0 + (float)k * interval;
                        drow[at + k] = interval;
                    }
                    left -= num;
                    ++seg;
                }
            }
            w.advance();
        }
    }
}

// ------------------------------------------------------------------ other samplers
__global__ void __launch_bounds__(256) k_sample_insideout(const float *__restrict__ rays_o,
                                                          const float *__restrict__ rays_d, int S, int S_bg,
                                                          const float *__restrict__ center,
                                                          const float *__restrict__ size, float far_,
                                                          float *__restrict__ z_vals, float *__restrict__ z_bg,
                                                          int32_t *missed, int B)
{
    float c[3] = { center[0], center[1], center[2] };
    float h[3] = { size[0] / 2.0f, size[1] / 2.0f, size[2] / 2.0f };
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B; i += gridDim.x * blockDim.x) {
        float o[3] = { rays_o[3 * i], rays_o[3 * i + 1], rays_o[3 * i + 2] };
        float d[3] = { rays_d[3 * i], rays_d[3 * i + 1], rays_d[3 * i + 2] };
        F2 b = clip_box(o, d, c, h);
        if (b.x == -1.0f || b.y == -1.0f) {
            if (missed) atomicAdd(missed, 1);
            continue;
        }
        float interval = (b.y - b.x) / (float)(S - 1);
        for (int k = 0; k < S; ++k) z_vals[(size_t)i * S + k] = b.x + (float)k * interval;
        float inv_near = 1.0f / b.y, inv_far = 1.0f / far_;
        float inv_bound = inv_far - inv_near;
        float stp = 1.0f / (float)(S_bg - 1);
        for (int k = 0; k < S_bg; ++k) z_bg[(size_t)i * S_bg + k] = 1.0f / (stp * (float)k * inv_bound + inv_near);
    }
}

// one thread per sample: rows are written coalesced
__global__ void __launch_bounds__(256) k_background_sampling(const float *__restrict__ starts,
                                                             const float *__restrict__ bg_depth,
                                                             float *__restrict__ z_vals, int S, float range, int B)
{
    int64_t total = (int64_t)B * S;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        int i = (int)(t / S), k = (int)(t % S);
        float near_ = fmaxf(starts[i] + 0.00001f, bg_depth[i] - range * 0.5f);
        float far_ = near_ + range;
        float interval = (far_ - near_) / (float)(S - 1);
        z_vals[t] = near_ + (float)k * interval;
    }
}

}  // namespace

// ---------------------------------------------------------------------------- C ABI
SCANERF_API int scanerf_compute_ray_forward(float *rays_o, float *rays_d, const float *Ks, const float *C2Ws,
                                            const int32_t *locs, int B, scanerf_stream_t stream)
{
    SCANERF_REQUIRE(B >= 0, "compute_ray_forward: B=%d", B);
    if (B == 0) return 0;
    SCANERF_REQUIRE(rays_o && rays_d && Ks && C2Ws && locs, "compute_ray_forward: null pointer");
    hipLaunchKernelGGL(k_compute_ray_fwd, dim3(stream_grid(B, 256)), dim3(256), 0, (hipStream_t)stream, rays_o,
                       rays_d, Ks, C2Ws, locs, B);
    return check_launch("compute_ray_forward");
}

SCANERF_API int scanerf_compute_ray_backward(const float *g_o, const float *g_d, const float *Ks, float *grad_C2Ws,
                                             const int32_t *locs, int B, int num_cam, scanerf_stream_t stream)
{
    SCANERF_REQUIRE(B >= 0 && num_cam >= 0, "compute_ray_backward: B=%d num_cam=%d", B, num_cam);
    if (B == 0) return 0;
    SCANERF_REQUIRE(g_o && g_d && Ks && grad_C2Ws && locs, "compute_ray_backward: null pointer");
    if (num_cam == 0) return 0;
    hipLaunchKernelGGL(k_compute_ray_bwd, dim3(num_cam < kNumCU * 2 ? num_cam : kNumCU * 2), dim3(kRayBwdThreads), 0,
                       (hipStream_t)stream, g_o, g_d, Ks, grad_C2Ws, locs, B, num_cam);
    return check_launch("compute_ray_backward");
}

SCANERF_API int scanerf_ray_aabb_intersection(const float *rays_o, const float *rays_d, const float *center,
                                              const float *size, float *bounds, int B, int K,
                                              scanerf_stream_t stream)
{
    SCANERF_REQUIRE(B >= 0 && K >= 1, "ray_aabb_intersection: B=%d K=%d", B, K);
    if (B == 0) return 0;
    SCANERF_REQUIRE(rays_o && rays_d && center && size && bounds, "ray_aabb_intersection: null pointer");
    hipLaunchKernelGGL(k_ray_aabb, dim3(stream_grid((int64_t)B * K, 256)), dim3(256), 0, (hipStream_t)stream,
                       rays_o, rays_d, center, size, bounds, B, K);
    return check_launch("ray_aabb_intersection");
}

SCANERF_API int scanerf_sample_points_grid(const float *rays_o, const float *rays_d, float *z_vals, float *dists,
                                           const float *block_corner, const float *block_size,
                                           const uint8_t *occ, const int32_t *log2dim, int B, int S,
                                           scanerf_stream_t stream)
{
    SCANERF_REQUIRE(B >= 0 && S >= 1, "sample_points_grid: B=%d S=%d", B, S);
    if (B == 0) return 0;
    SCANERF_REQUIRE(rays_o && rays_d && z_vals && dists && block_corner && block_size && occ && log2dim,
                    "sample_points_grid: null pointer");
    // 64-thread blocks: DDA trip counts diverge per ray, small blocks retire independently
    hipLaunchKernelGGL(k_sample_points_grid, dim3(stream_grid(B, 64, kNumCU * 64)), dim3(64), 0,
                       (hipStream_t)stream, rays_o, rays_d, z_vals, dists, block_corner, block_size, occ, log2dim,
                       B, S);
    return check_launch("sample_points_grid");
}

SCANERF_API int scanerf_sample_insideout_block(const float *rays_o, const float *rays_d, int S, int S_bg,
                                               const float *block_center, const float *block_size, float far_,
                                               float *z_vals, float *z_vals_bg, int32_t *missed, int B,
                                               scanerf_stream_t stream)
{
    SCANERF_REQUIRE(B >= 0 && S >= 2 && S_bg >= 2, "sample_insideout_block: B=%d S=%d S_bg=%d", B, S, S_bg);
    if (B == 0) return 0;
    SCANERF_REQUIRE(rays_o && rays_d && block_center && block_size && z_vals && z_vals_bg,
                    "sample_insideout_block: null pointer");
    hipLaunchKernelGGL(k_sample_insideout, dim3(stream_grid(B, 64, kNumCU * 64)), dim3(64), 0, (hipStream_t)stream,
                       rays_o, rays_d, S, S_bg, block_center, block_size, far_, z_vals, z_vals_bg, missed, B);
    return check_launch("sample_insideout_block");
}

SCANERF_API int scanerf_background_sampling(const float *starts, const float *bg_depth, float *z_vals, int S,
                                            float sample_range, int B, scanerf_stream_t stream)
{
    SCANERF_REQUIRE(B >= 0 && S >= 2, "background_sampling: B=%d S=%d", B, S);
    if (B == 0) return 0;
    SCANERF_REQUIRE(starts && bg_depth && z_vals, "background_sampling: null pointer");
    hipLaunchKernelGGL(k_background_sampling, dim3(stream_grid((int64_t)B * S, 256)), dim3(256), 0,
                       (hipStream_t)stream, starts, bg_depth, z_vals, S, sample_range, B);
    return check_launch("background_sampling");
}

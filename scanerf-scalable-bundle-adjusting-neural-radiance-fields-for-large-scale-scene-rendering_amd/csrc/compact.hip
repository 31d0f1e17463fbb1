// compact.hip -- valid-ray compaction of a training batch in one pass over the data (gfx950).
//
// Reference behaviour: hashgrid/__init__.py:419-434 -- rays that meet no occupied cell keep the sampler's -1 sentinels,
// `valid = torch.all(z_vals != -1, dim=-1)`, and only rays_o[valid] / rays_d[valid] / z_vals[valid] / dists[valid] are
// rendered (boolean-mask indexing: a nonzero, five gathers and their temporaries).  The fused backward runs its waves in
// lock step, so a masked ray costs as much as a valid one: with a sparse occupancy grid the batch is compacted first.
//
// k_ray_valid:   one wave per ray row, valid = every sample != -1            (reads z once, coalesced)
// k_compact:     256 rays per workgroup; destination index = (valid rays before the workgroup: a sum over the flag bytes
//                already in L2) + (valid rays of earlier waves: LDS) + (valid lanes below: __ballot + popcount);
//                per-ray vectors are moved by their lane, the [S] rows of z and dists by the whole wave, row by row over
//                the set bits of the ballot.  Order-preserving, no atomics, deterministic.
#include "common.h"

using namespace scanerf;

namespace {

constexpr int kRaysPerBlock = 256;

__global__ void __launch_bounds__(256) k_ray_valid(const float *__restrict__ z_vals, uint8_t *__restrict__ valid, int B, int S)
{
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int ray = wave; ray < B; ray += nwaves) {
        bool ok = true;
        for (int s = lane; s < S; s += 64) ok &= z_vals[(size_t)ray * S + s] != -1.0f;
        const bool all_ok = __ballot(!ok) == 0ull;
        if (lane == 0) valid[ray] = all_ok ? 1 : 0;
    }
}

__global__ void __launch_bounds__(kRaysPerBlock) k_compact(const uint8_t *__restrict__ valid, int B, int S,
                                                           const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                                                           const float *__restrict__ target, const float *__restrict__ z_vals,
                                                           const float *__restrict__ dists, float *__restrict__ out_o,
                                                           float *__restrict__ out_d, float *__restrict__ out_t,
                                                           float *__restrict__ out_z, float *__restrict__ out_dist,
                                                           int32_t *__restrict__ out_index, int32_t *__restrict__ count)
{
    __shared__ int part[kRaysPerBlock / 64 + 1];
    __shared__ int red[kRaysPerBlock / 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int first = blockIdx.x * kRaysPerBlock;
    // valid rays before this workgroup (first is a multiple of 256: whole 16-byte chunks)
    int before = 0;
    for (int i = threadIdx.x; i < first / 16; i += kRaysPerBlock) {
        const uint4 v = reinterpret_cast<const uint4 *>(valid)[i];
        before += __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w);  // flags are 0 / 1 bytes
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) before += __shfl_xor(before, off, 64);
    if (lane == 0) red[wv] = before;
    const int ray = first + threadIdx.x;
    const bool keep = ray < B && valid[ray] != 0;
    const unsigned long long ballot = __ballot(keep);
    if (lane == 0) part[wv + 1] = __popcll(ballot);
    __syncthreads();
    int base = 0;
#pragma unroll
    for (int w = 0; w < kRaysPerBlock / 64; ++w) base += red[w];
    int wave_off = 0;
    for (int w = 0; w < wv; ++w) wave_off += part[w + 1];
    const int wave_base = base + wave_off;
    const int dst = wave_base + __popcll(ballot & ((1ull << lane) - 1ull));
    if (keep) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            out_o[3 * (size_t)dst + k] = rays_o[3 * (size_t)ray + k];
            out_d[3 * (size_t)dst + k] = rays_d[3 * (size_t)ray + k];
            if (target) out_t[3 * (size_t)dst + k] = target[3 * (size_t)ray + k];
        }
        if (out_index) out_index[dst] = ray;
    }
    // rows of z and dists: the whole wave moves one valid ray's row at a time
    unsigned long long todo = ballot;
    int n = 0;
    while (todo) {
        const int src_lane = __builtin_ctzll(todo);
        todo &= todo - 1ull;
        const size_t src = (size_t)(first + wv * 64 + src_lane) * S, dstrow = (size_t)(wave_base + n) * S;
        for (int s = lane; s < S; s += 64) {
            out_z[dstrow + s] = z_vals[src + s];
            out_dist[dstrow + s] = dists[src + s];
        }
        ++n;
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        int total = base;
        for (int w = 0; w < kRaysPerBlock / 64; ++w) total += part[w + 1];
        *count = total;
    }
}

}  // namespace

// valid [B] u8 = every sample of the ray's z row != -1 (hashgrid/__init__.py:419)
SCANERF_API int scanerf_ray_valid(const float *z_vals, uint8_t *valid, int B, int S, scanerf_stream_t stream)
{
    SCANERF_REQUIRE(B >= 0 && S >= 1, "ray_valid: B=%d S=%d", B, S);
    if (B == 0) return 0;
    SCANERF_REQUIRE(z_vals && valid, "ray_valid: null pointer");
    hipLaunchKernelGGL(k_ray_valid, dim3(stream_grid((int64_t)B * 64, 256)), dim3(256), 0, (hipStream_t)stream, z_vals, valid, B, S);
    return check_launch("ray_valid");
}

// Order-preserving compaction of the valid rays (hashgrid/__init__.py:419-434): out_* hold the valid rays' entries in their
// original order in rows [0, *count); target / out_t and out_index (the source ray of every output row) may be NULL.
// valid must be 16-byte aligned with B rounded up to a multiple of 16 readable bytes.
SCANERF_API int scanerf_compact_rays(const uint8_t *valid, int B, int S, const float *rays_o, const float *rays_d,
                                     const float *target, const float *z_vals, const float *dists, float *out_o, float *out_d,
                                     float *out_t, float *out_z, float *out_dist, int32_t *out_index, int32_t *count,
                                     scanerf_stream_t stream)
{
    SCANERF_REQUIRE(B >= 0 && S >= 1, "compact_rays: B=%d S=%d", B, S);
    SCANERF_REQUIRE(count, "compact_rays: count is null");
    if (B == 0) return hipMemsetAsync(count, 0, sizeof(int32_t), (hipStream_t)stream) == hipSuccess ? 0 : 1;
    SCANERF_REQUIRE(valid && rays_o && rays_d && z_vals && dists && out_o && out_d && out_z && out_dist && (!target || out_t),
                    "compact_rays: null pointer");
    SCANERF_REQUIRE(((uintptr_t)valid & 15) == 0, "compact_rays: valid must be 16-byte aligned");
    hipLaunchKernelGGL(k_compact, dim3(ceil_div(B, kRaysPerBlock)), dim3(kRaysPerBlock), 0, (hipStream_t)stream, valid, B, S, rays_o,
                       rays_d, target, z_vals, dists, out_o, out_d, out_t, out_z, out_dist, out_index, count);
    return check_launch("compact_rays");
}

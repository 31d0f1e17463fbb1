// voxelize.hip -- occupancy initialisation of a tile's sampling grid from a triangle mesh.
//
// Reference: cuda/include/voxelize.h:12-119 (a host loop over faces inside the CUDA_EXT module).  Per face: the
// axis-aligned box of the triangle, inflated 1.5x about its centre; faces whose box misses the grid box are skipped; the
// cells the box overlaps (indices clamped to the grid) become occupied; with init_out, every cell whose centre lies
// outside the union box of the accepted faces becomes occupied AND is flagged outside.
//
// Here: one wavefront per face marks its cell range (byte stores; racing writers all store 1), the union box is
// reduced with ordered-integer atomics, a second launch does the init_out sweep.  Built without FMA contraction:
// the float sequence is the reference's, so the grids are bit-identical to the oracle's.
#include "common.h"

using namespace scanerf;

namespace {

struct VoxArgs {
    const float *vertices;   // [V,3]
    const int32_t *faces;    // [F,3]
    int V, F;
    int l2d[3];
    float bmin[3], bsize[3];
    uint8_t *vis, *outside;
    uint32_t *geo;           // [6] ordered keys: min xyz, max xyz
};

// order-preserving float <-> uint32 map (atomicMin/Max on the keys = min/max on the floats)
__device__ __forceinline__ uint32_t fkey(float f)
{
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float fval(uint32_t k)
{
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

__global__ void k_vox_init(uint32_t *geo)
{
    if (threadIdx.x < 3) geo[threadIdx.x] = fkey(100000000.0f);
    else if (threadIdx.x < 6) geo[threadIdx.x] = fkey(-1.0f * 100000000.0f);
}

__global__ void __launch_bounds__(256) k_vox_faces(VoxArgs a)
{
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int f = wave; f < a.F; f += nwaves) {
        const int i0 = a.faces[3 * f], i1 = a.faces[3 * f + 1], i2 = a.faces[3 * f + 2];
        if ((unsigned)i0 >= (unsigned)a.V || (unsigned)i1 >= (unsigned)a.V || (unsigned)i2 >= (unsigned)a.V) continue;
        float mn[3], mx[3];
        int lo[3], hi[3];
        bool reject = false;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float A = a.vertices[3 * i0 + c], B = a.vertices[3 * i1 + c], C = a.vertices[3 * i2 + c];
            const float mnc = fminf(fminf(A, B), C), mxc = fmaxf(fmaxf(A, B), C);
            const float center = (mnc + mxc) / 2.0f;
            const float half = ((mxc - mnc) * 1.5f) / 2.0f;
            mn[c] = center - half;
            mx[c] = center + half;
            const float bmax = a.bmin[c] + a.bsize[c];
            reject |= mx[c] <= a.bmin[c] || mn[c] >= bmax;
        }
        if (reject) continue;  // wave-uniform
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int res = 1 << a.l2d[c];
            const float gs = a.bsize[c] / (float)res;
            const int l = (int)((mn[c] - a.bmin[c]) / gs), h = (int)((mx[c] - a.bmin[c]) / gs);
            lo[c] = l < 0 ? 0 : (l > res - 1 ? res - 1 : l);
            hi[c] = h < 0 ? 0 : (h > res - 1 ? res - 1 : h);
        }
        if (lane == 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                atomicMin(&a.geo[c], fkey(mn[c]));
                atomicMax(&a.geo[3 + c], fkey(mx[c]));
            }
        }
        const int ny = hi[1] - lo[1] + 1, nz = hi[2] - lo[2] + 1;
        const int64_t n = (int64_t)(hi[0] - lo[0] + 1) * ny * nz;
        for (int64_t i = lane; i < n; i += 64) {
            const int z = lo[2] + (int)(i % nz), y = lo[1] + (int)((i / nz) % ny), x = lo[0] + (int)(i / ((int64_t)nz * ny));
            a.vis[((uint32_t)x << (a.l2d[1] + a.l2d[2])) | ((uint32_t)y << a.l2d[2]) | (uint32_t)z] = 1;
        }
    }
}

__global__ void __launch_bounds__(256) k_vox_outside(VoxArgs a)
{
    const int64_t total = (int64_t)1 << (a.l2d[0] + a.l2d[1] + a.l2d[2]);
    float gmn[3], gmx[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        gmn[c] = fval(a.geo[c]);
        gmx[c] = fval(a.geo[3 + c]);
    }
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < total; n += (int64_t)gridDim.x * blockDim.x) {
        const int idx[3] = { (int)(n >> (a.l2d[1] + a.l2d[2])), (int)((n >> a.l2d[2]) & ((1 << a.l2d[1]) - 1)),
                             (int)(n & ((1 << a.l2d[2]) - 1)) };
        bool out = false;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float gs = a.bsize[c] / (float)(1 << a.l2d[c]);
            const float loc = a.bmin[c] + (float)idx[c] * gs + gs / 2.0f;
            out |= loc < gmn[c] || loc > gmx[c];
        }
        if (out) {
            a.vis[n] = 1;
            a.outside[n] = 1;
        }
    }
}

}  // namespace

SCANERF_API int scanerf_voxelize_mesh(const float *vertices, const int32_t *faces, int V, int F, const int32_t *log2dim,
                                      const float *block_corner, const float *block_size, uint8_t *vis, int init_out,
                                      uint8_t *outside, uint32_t *scratch6, scanerf_stream_t stream)
{
    SCANERF_REQUIRE(V >= 0 && F >= 0, "voxelize_mesh: V=%d F=%d", V, F);
    SCANERF_REQUIRE(log2dim && block_corner && block_size && vis && scratch6, "voxelize_mesh: null pointer");
    SCANERF_REQUIRE(!init_out || outside, "voxelize_mesh: init_out needs the outside grid");
    SCANERF_REQUIRE(F == 0 || (vertices && faces), "voxelize_mesh: null mesh");
    VoxArgs a;
    a.vertices = vertices; a.faces = faces; a.V = V; a.F = F; a.vis = vis; a.outside = outside; a.geo = scratch6;
    for (int c = 0; c < 3; ++c) {
        SCANERF_REQUIRE(log2dim[c] >= 0 && log2dim[c] <= 10, "voxelize_mesh: log2dim[%d]=%d", c, log2dim[c]);
        a.l2d[c] = log2dim[c];
        a.bmin[c] = block_corner[c];
        a.bsize[c] = block_size[c];
    }
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_vox_init, dim3(1), dim3(64), 0, st, scratch6);
    if (F > 0) hipLaunchKernelGGL(k_vox_faces, dim3(stream_grid((int64_t)F * 64, 256)), dim3(256), 0, st, a);
    if (init_out) {
        const int64_t total = (int64_t)1 << (a.l2d[0] + a.l2d[1] + a.l2d[2]);
        hipLaunchKernelGGL(k_vox_outside, dim3(stream_grid(total, 256)), dim3(256), 0, st, a);
    }
    return check_launch("voxelize_mesh");
}

// loss.hip -- photometric loss of the training step and its gradient w.r.t. the per-ray outputs, two launches.
//
// Reference: criterions.py:90,142-144 (MSE over the valid rays' rgb) + tile.py:999 (0.01 * l2_reg_specular, the mean
// over valid rays x 3 channels of sum_i w_i |c_s,i|^2, hashgrid/__init__.py:593).  The reference builds this with
// ~20 torch ops on [B,3] tensors and autograd; at 65 536 rays the host-side launch gaps of those ops cost more than the
// arithmetic, so the fused training step computes loss and dL/d(out_ray) directly.
#include "common.h"

using namespace scanerf;

namespace {

constexpr int kBlocks = 256, kThreads = 256;

// per-block partial sums: [block][3] = (sum of squared rgb errors, sum of the w*|c_s|^2 column, number of valid rays)
__global__ void __launch_bounds__(kThreads) k_loss_partials(const float *__restrict__ out_ray, const float *__restrict__ target,
                                                            const uint8_t *__restrict__ valid, int B, float *__restrict__ partials)
{
    float se = 0.0f, w2 = 0.0f, n = 0.0f;
    for (int r = blockIdx.x * kThreads + threadIdx.x; r < B; r += kBlocks * kThreads) {
        if (valid && !valid[r]) continue;
        const float4 o = *reinterpret_cast<const float4 *>(out_ray + (size_t)r * SCANERF_RAY_OUT);
        const float dx = o.x - target[3 * r], dy = o.y - target[3 * r + 1], dz = o.z - target[3 * r + 2];
        se += dx * dx + dy * dy + dz * dz;
        w2 += out_ray[(size_t)r * SCANERF_RAY_OUT + 14];
        n += 1.0f;
    }
    __shared__ float red[3][kThreads];
    red[0][threadIdx.x] = se; red[1][threadIdx.x] = w2; red[2][threadIdx.x] = n;
    __syncthreads();
    for (int s = kThreads / 2; s > 0; s >>= 1) {  // fixed tree: deterministic
        if ((int)threadIdx.x < s)
#pragma unroll
            for (int k = 0; k < 3; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x < 3) partials[blockIdx.x * 3 + threadIdx.x] = red[threadIdx.x][0];
}

__global__ void __launch_bounds__(kThreads) k_loss_grad(const float *__restrict__ out_ray, const float *__restrict__ target,
                                                        const uint8_t *__restrict__ valid, int B, float reg,
                                                        const float *__restrict__ partials, float *__restrict__ grad_out,
                                                        float *__restrict__ loss)
{
    __shared__ float tot[3];
    if (threadIdx.x < 3) {  // every block adds the 256 partials in the same order
        float s = 0.0f;
        for (int b = 0; b < kBlocks; ++b) s += partials[b * 3 + threadIdx.x];
        tot[threadIdx.x] = s;
    }
    __syncthreads();
    const float nv3 = 3.0f * tot[2];
    if (blockIdx.x == 0 && threadIdx.x == 0) *loss = nv3 > 0.0f ? (tot[0] + reg * tot[1]) / nv3 : 0.0f;
    const float ginv = nv3 > 0.0f ? 1.0f / nv3 : 0.0f;
    for (int r = blockIdx.x * kThreads + threadIdx.x; r < B; r += gridDim.x * kThreads) {
        float4 g0 = make_float4(0, 0, 0, 0), g3 = make_float4(0, 0, 0, 0);
        if (!valid || valid[r]) {
            const float4 o = *reinterpret_cast<const float4 *>(out_ray + (size_t)r * SCANERF_RAY_OUT);
            g0.x = 2.0f * (o.x - target[3 * r]) * ginv;
            g0.y = 2.0f * (o.y - target[3 * r + 1]) * ginv;
            g0.z = 2.0f * (o.z - target[3 * r + 2]) * ginv;
            g3.z = reg * ginv;  // column 14
        }
        float4 *g = reinterpret_cast<float4 *>(grad_out + (size_t)r * SCANERF_RAY_OUT);
        g[0] = g0;
        g[1] = make_float4(0, 0, 0, 0);
        g[2] = make_float4(0, 0, 0, 0);
        g[3] = g3;
    }
}

}  // namespace

SCANERF_API int scanerf_photometric_loss_scratch_floats(void) { return kBlocks * 3; }

SCANERF_API int scanerf_photometric_loss_grad(const float *out_ray, const float *target, const uint8_t *ray_valid,
                                              float reg_weight, float *grad_out, float *loss, float *scratch, int B,
                                              scanerf_stream_t stream)
{
    SCANERF_REQUIRE(B >= 0, "photometric_loss_grad: B=%d", B);
    SCANERF_REQUIRE(out_ray && target && grad_out && loss && scratch, "photometric_loss_grad: null pointer");
    SCANERF_REQUIRE(((uintptr_t)out_ray & 15) == 0 && ((uintptr_t)grad_out & 15) == 0,
                    "photometric_loss_grad: out_ray / grad_out must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_loss_partials, dim3(kBlocks), dim3(kThreads), 0, st, out_ray, target, ray_valid, B, scratch);
    hipLaunchKernelGGL(k_loss_grad, dim3(stream_grid(B > 0 ? B : 1, kThreads, kBlocks)), dim3(kThreads), 0, st, out_ray, target,
                       ray_valid, B, reg_weight, scratch, grad_out, loss);
    return check_launch("photometric_loss_grad");
}

// Device check of the DPP row operations of csrc/render_t16.h (row_shr / row_shl / row_ror within the 16-lane rows) against a
// host evaluation.  hipcc --offload-arch=gfx950 -o dpp_rows_test dpp_rows_test.hip && ./dpp_rows_test
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
template <int N> __device__ float row_shr(float v, float fill) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, fill), __builtin_bit_cast(int, v), 0x110 + N, 0xf, 0xf, false)); }
template <int N> __device__ float row_shl(float v, float fill) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, fill), __builtin_bit_cast(int, v), 0x100 + N, 0xf, 0xf, false)); }
template <int N> __device__ float row_ror(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + N, 0xf, 0xf, false)); }
__global__ void k(const float* in, float* out) {
    float fi = in[threadIdx.x];
    float incl = fi;
    incl *= row_shr<1>(incl, 1.0f); incl *= row_shr<2>(incl, 1.0f); incl *= row_shr<4>(incl, 1.0f); incl *= row_shr<8>(incl, 1.0f);
    float excl = row_shr<1>(incl, 1.0f);
    float rs = fi;
    rs += row_shl<1>(rs, 0.0f); rs += row_shl<2>(rs, 0.0f); rs += row_shl<4>(rs, 0.0f); rs += row_shl<8>(rs, 0.0f);
    float mx = fi; mx = fmaxf(mx, row_ror<8>(mx)); mx = fmaxf(mx, row_ror<4>(mx)); mx = fmaxf(mx, row_ror<2>(mx)); mx = fmaxf(mx, row_ror<1>(mx));
    out[threadIdx.x] = incl; out[64 + threadIdx.x] = excl; out[128 + threadIdx.x] = rs; out[192 + threadIdx.x] = mx;
}
int main() {
    float h[64], o[256], *di, *dout;
    for (int i = 0; i < 64; ++i) h[i] = 0.5f + 0.01f * ((i * 37) % 64);
    hipMalloc(&di, 256); hipMalloc(&dout, 1024);
    hipMemcpy(di, h, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, dout);
    hipMemcpy(o, dout, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int r = 0; r < 4; ++r) {
        float p = 1.0f, mx = 0.0f;
        for (int c = 0; c < 16; ++c) mx = fmaxf(mx, h[16 * r + c]);
        for (int c = 0; c < 16; ++c) {
            const float ex = p;
            p *= h[16 * r + c];
            float suf = 0.0f;
            for (int j = 15; j >= c; --j) suf += h[16 * r + j];
            const int l = 16 * r + c;
            if (fabsf(o[l] - p) > 1e-5f * p || fabsf(o[64 + l] - ex) > 1e-5f * ex || fabsf(o[128 + l] - suf) > 1e-5f * suf || o[192 + l] != mx) {
                ++bad;
                if (bad < 8) printf("lane %d: incl %g (want %g) excl %g (%g) suffix %g (%g) max %g (%g)\n", l, o[l], p, o[64 + l], ex, o[128 + l], suf, o[192 + l], mx);
            }
        }
    }
    printf("%d lanes wrong\n", bad);
    return bad != 0;
}

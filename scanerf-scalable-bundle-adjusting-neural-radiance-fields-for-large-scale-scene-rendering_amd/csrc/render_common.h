// render_common.h -- layout of the packed decoder image shared by the pack kernel and the
// fused render kernels (forward and backward).
//
// The decoder is network.ShallowMLP (network.py:151-190); its parameters arrive as the
// reference's flat render-time blob (hashgrid/include/decoder.h:48-67, rendering.py:101-112):
// per layer [bias(out), W^T (in-major, out fastest)], layers in state-dict order.
//
// MLP on the matrix cores, fp32 exact (v_mfma_f32_32x32x2_f32).  A wave works on a tile of
// 32 samples and computes every layer TRANSPOSED: H^T[n][s] = sum_k W[n][k] X^T[k][s], with
// the weights as the A operand (from LDS) and the activations as the B operand.  The
// accumulator of one layer (column = sample on the lane, row = unit in the register) is
// then, after the activation, directly the B operand of the next layer: activations never
// leave registers and never touch LDS.
//
//   lane l: sample s = l & 31, half h = l >> 5
//   accumulator register g of block b holds unit  n = 32*b + nmap(g,h),
//   nmap(g,h) = (g & 3) + 8*(g >> 2) + 4*h            (the 32x32 C/D register map)
//   MFMA step r consumes one input per half: k = kmap_layer(r, h).
#pragma once
#include "common.h"

namespace scanerf {

// blob offsets (floats)
constexpr int BLOB_S0 = 0;       // 64 bias + 32x64
constexpr int BLOB_S1 = 2112;    // 64 + 64x64
constexpr int BLOB_SIG = 6272;   // 1 + 32x1
constexpr int BLOB_DIF = 6305;   // 3 + 32x3
constexpr int BLOB_TINT = 6404;  // 3 + 32x3
constexpr int BLOB_D0 = 6503;    // 64 + 48x64
constexpr int BLOB_D1 = 9639;    // 64 + 64x64
constexpr int BLOB_D2 = 13799;   // 3 + 64x3
static_assert(BLOB_D2 + 3 + 192 == SCANERF_PARAMSIZE, "blob layout");

// packed image offsets (floats).  A-images are [block][group of 4 steps][lane][4].
constexpr int PK_L0 = 0;        // 2 x 4 x 256   (K=32; weight_feature folded in)
constexpr int PK_L1 = 2048;     // 2 x 8 x 256   (K=64)
constexpr int PK_D0H = 6144;    // 2 x 4 x 256   (K=32: the H[32:64] half of the 48 inputs)
constexpr int PK_D0S = 8192;    // 2 x 2 x 256   (K=16: the SH half)
constexpr int PK_D1 = 9216;     // 2 x 8 x 256   (K=64)
constexpr int PK_BIAS = 13312;  // [layer 4][block 2][half 2][16]
constexpr int PK_HEAD = 13568;  // [half 2][g 16][8]: sigma, diffuse xyz, tint xyz, 0
constexpr int PK_D2 = 13824;    // [half 2][block 2][g 16][4]: rgb rows of Directional_MLP.mlp.4, 0
constexpr int PK_HB = 14080;    // head biases: sigma, dif3, tint3, 0, d2 xyz, 0...
constexpr int PK_TOTAL = 14096;

__host__ __device__ constexpr int nmap(int g, int h) { return (g & 3) + 8 * (g >> 2) + 4 * h; }

// Transposed A-images for the backward pass (dX^T = W^T dY^T): same scheme with the roles
// of input and output swapped.  [block of 32 INPUT units][group][lane][4]
constexpr int PKT_L1 = 0;       // 2 x 8 x 256: A[i = input k][step -> output n]
constexpr int PKT_D0H = 4096;   // 1 x 8 x 256: inputs = H[32:64] (32 rows), K = 64 outputs
constexpr int PKT_D1 = 6144;    // 2 x 8 x 256
constexpr int PKT_L0 = 10240;   // 1 x 8 x 256: inputs = 32 features (weight_feature folded), K = 64
constexpr int PKT_TOTAL = 12288;

}  // namespace scanerf

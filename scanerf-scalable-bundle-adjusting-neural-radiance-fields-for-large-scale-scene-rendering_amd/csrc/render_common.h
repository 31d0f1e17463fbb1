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

// packed image offsets (floats).  A-images are [block][group of 4 steps][lane slot][4] with a
// group stride of 264 floats and the upper half-wave shifted by one 16-byte slot
// (slot = lane + (lane >> 5)).  The forward reads one float4 per lane per group (conflict-free
// either way); the padding makes the TRANSPOSED walk of the same image -- lane = input unit,
// step = output unit, one ds_read_b32 per MFMA, used by the backward chain dX^T = W^T dY^T --
// hit 32 distinct banks, so no second (transposed) copy of the weights is needed in LDS.
constexpr int PK_GRP = 264;
constexpr int PK_L0 = 0;                      // 2 x 4 groups (K=32; weight_feature folded in)
constexpr int PK_L1 = PK_L0 + 8 * PK_GRP;     // 2 x 8 groups (K=64)
constexpr int PK_D0H = PK_L1 + 16 * PK_GRP;   // 2 x 4 groups (K=32: the H[32:64] half of the 48 inputs)
constexpr int PK_D0S = PK_D0H + 8 * PK_GRP;   // 2 x 2 groups (K=16: the SH half)
constexpr int PK_D1 = PK_D0S + 4 * PK_GRP;    // 2 x 8 groups (K=64)
constexpr int PK_BIAS = PK_D1 + 16 * PK_GRP;  // [layer 4][block 2][half 2][16]
constexpr int PK_HEAD = PK_BIAS + 256;        // [half 2][g 16][8]: sigma, diffuse xyz, tint xyz, 0
constexpr int PK_D2 = PK_HEAD + 256;          // [half 2][block 2][g 16][4]: rgb rows of Directional_MLP.mlp.4, 0
constexpr int PK_HB = PK_D2 + 256;            // head biases: sigma, dif3, tint3, 0, d2 xyz, 0...
constexpr int PK_TOTAL = PK_HB + 16;
static_assert(PK_TOTAL % 4 == 0, "image is copied as float4");

// ---- f16 split ("h3") image, see render_h3.h (bytes from the start of the h3 part of the workspace / LDS copy)
constexpr int H3_SUB = 1088;
constexpr int H3_L0 = 0;                       // 2 blocks x 2 k-steps (weight_feature folded in)
constexpr int H3_L1 = H3_L0 + 8 * H3_SUB;      // 2 x 4
constexpr int H3_HEAD = H3_L1 + 16 * H3_SUB;   // 1 x 2: rows 0-3 sigma,dif; 8-10 tint; replicas at +4
constexpr int H3_D0 = H3_HEAD + 4 * H3_SUB;    // 2 x 3: k-steps 0,1 = H[32:64], k-step 2 = SH (slot 8h+j = SH[8h+j])
constexpr int H3_D1 = H3_D0 + 12 * H3_SUB;     // 2 x 4
constexpr int H3_D2 = H3_D1 + 16 * H3_SUB;     // 1 x 4: rows 0-2 rgb; replicas at +4
constexpr int H3_IMG_BYTES = H3_D2 + 8 * H3_SUB;
// f32 tail: accumulator start values
constexpr int H3_BIAS = H3_IMG_BYTES;          // [layer 4][block 2][half 2][16] f32, as PK_BIAS
constexpr int H3_HB = H3_BIAS + 256 * 4;       // head accumulator start [16]: regs 0-3 sigma,dif; 4-6 tint
constexpr int H3_D2B = H3_HB + 16 * 4;         // rgb accumulator start [16]: regs 0-2
constexpr int H3_BYTES = H3_D2B + 16 * 4;
static_assert(H3_BYTES % 16 == 0 && H3_IMG_BYTES % 256 == 0, "h3 image is copied as float4");
constexpr int H3_FLOATS = H3_BYTES / 4;

// ---- 16-sample-tile images of the backward kernel render_bwd_t16.hip, see render_t16.h (bytes from the start of the t16 part)
constexpr int T16_SUB = 1024;
constexpr int T16_PAIR = 2 * T16_SUB;
// forward image: pairs [block b][k-step t]
constexpr int T16_L0 = 0;                               // 4 x 1   (weight_feature folded in; slot (q, j) = x-stash position 8q + j)
constexpr int T16_L1 = T16_L0 + 4 * T16_PAIR;           // 4 x 2
constexpr int T16_HEAD = T16_L1 + 8 * T16_PAIR;         // 2 x 1   rows 4q+g: block 0 = (sigma, dif xyz), block 1 = (tint xyz, 0), every q
constexpr int T16_D0 = T16_HEAD + 2 * T16_PAIR;         // 4 x 2   k-step 0 = H[32:64], k-step 1 = SH (slot (q, j) = SH[8q + j], q < 2)
constexpr int T16_D1 = T16_D0 + 8 * T16_PAIR;           // 4 x 2
constexpr int T16_D2 = T16_D1 + 8 * T16_PAIR;           // 1 x 2   rows 4q+g = (rgb xyz, 0), every q
constexpr int T16_FWD_BYTES = T16_D2 + 2 * T16_PAIR;    // 32 pairs
// transposed image: sub-images [input block b_in][k-step t of the layer's OUTPUT units]
constexpr int T16T_D2 = T16_FWD_BYTES;                  // 4 x 1   k = narrow rows (8..10 = rgb)
constexpr int T16T_D1 = T16T_D2 + 4 * T16_SUB;          // 4 x 2
constexpr int T16T_D0 = T16T_D1 + 8 * T16_SUB;          // 2 x 2   input = H[32:64] (dH blocks 2, 3)
constexpr int T16T_HEAD = T16T_D0 + 4 * T16_SUB;        // 2 x 1   k = narrow rows (0..6 = sigma, dif, tint); dH blocks 0, 1
constexpr int T16T_L1 = T16T_HEAD + 2 * T16_SUB;        // 4 x 2
constexpr int T16T_L0 = T16T_L1 + 8 * T16_SUB;          // 2 x 2   rows = x-stash positions (see t16_pos)
constexpr int T16_BIAS = T16T_L0 + 4 * T16_SUB;         // f32 [L0 64][L1 64][D0 64][D1 64][headA 4][headB 4][D2 4][pad 4]
constexpr int T16_BYTES = T16_BIAS + (256 + 16) * 4;
static_assert(T16_BYTES % 16 == 0, "t16 image is copied as float4");
constexpr int T16_FLOATS = T16_BYTES / 4;


// ---- images of the split-gradient variant of that kernel ("t16s", render_t16.h): ONE image serves the forward recompute and
// the transposed products (sub-images of two 512-byte halves with XOR-swizzled lane slots), hi and lo parts; the narrow layers'
// transposed sub-images (hi, lo) as extra pairs; the same f32 tail
constexpr int S16T_D2 = T16_FWD_BYTES;                  // 4 pairs: [input block b_in], k = narrow rows (8..10 = rgb)
constexpr int S16T_HEAD = S16T_D2 + 4 * T16_PAIR;       // 2 pairs: dH blocks 0, 1; k = narrow rows (0..6 = sigma, dif, tint)
constexpr int S16_BIAS = S16T_HEAD + 2 * T16_PAIR;
constexpr int S16_BYTES = S16_BIAS + (256 + 16) * 4;
static_assert(S16_BYTES % 16 == 0, "t16s image is copied as float4");
constexpr int S16_FLOATS = S16_BYTES / 4;

// packed workspace = [fp32 image PK_TOTAL floats][h3 image H3_FLOATS floats][t16 images T16_FLOATS floats][t16s images S16_FLOATS floats]
constexpr int WS_T16 = PK_TOTAL + H3_FLOATS;
constexpr int WS_S16 = WS_T16 + T16_FLOATS;
constexpr int WS_FLOATS = WS_S16 + S16_FLOATS;

__host__ __device__ constexpr int nmap(int g, int h) { return (g & 3) + 8 * (g >> 2) + 4 * h; }

// inverse of nmap within a 32-unit block: unit i5 -> (register g, half h)
__host__ __device__ constexpr int nmap_g(int i5) { return (i5 & 3) + 4 * (i5 >> 3); }
__host__ __device__ constexpr int nmap_h(int i5) { return (i5 >> 2) & 1; }

// Coarse-to-fine mask -> the levels the kernels leave out: level l only together with l ^ 2, the level the other half-wave
// handles in the same register pair (encode8), so that the decision is uniform per wave.
inline unsigned pair_masked_levels(unsigned masked)
{
    return masked & (((masked >> 2) & 0x3333u) | ((masked << 2) & 0xccccu));
}

}  // namespace scanerf

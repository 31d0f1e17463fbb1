// render_t16.h -- decoder images and matrix-core primitives of the 16-sample-tile backward kernel
// (render_bwd_t16.hip): v_mfma_f32_16x16x32_f16, two waves per SIMD.
//
// Why a second tiling.  The 32-sample-tile kernel (render_bwd_h3.hip) needs 512 registers and 150 KB of LDS per
// 4-wave workgroup: one wave per SIMD, every LDS / memory / MFMA latency exposed, VALU issue at half rate
// (one wave alone issues a vector instruction every 4 cycles; two waves share the SIMD at one per 2).  With
// 16-sample tiles every per-sample array halves, so a wave fits 256 registers and 8 waves share one CU.
//
// Lane map of v_mfma_f32_16x16x32_f16 (D = A B + C, A 16x32, B 32x16, C/D 16x16):
//     lane l: c = l & 15, q = l >> 4
//     A: lane holds A[row c][k-slot 8q + j], j = 0..7        (one h8 = 4 registers)
//     B: lane holds B[k-slot 8q + j][col c]
//     D: lane holds D[row 4q + g][col c], g = 0..3           (one v4f)
// The decoder is evaluated transposed, as in the other fused kernels: rows = units, columns = samples, weights are
// the A operand (from LDS), activations the B operand.  A layer's output is 4 row blocks b of 16 units:
// block b, lane (c, q), register g = unit 16b + 4q + g of sample c.  The B operand of k-step t of the NEXT layer
// is blocks 2t and 2t+1 converted in place: slot (q, j) = unit ku(t, q, j) = 32t + 16(j >> 2) + 4q + (j & 3),
// so activations never move between lanes; the weight images store W[n][ku(...)] at that slot.
//
// Arithmetic.  Forward recompute: split f16 ("h3", render_h3.h): every operand hi + lo, three MFMAs per term,
// f32 accumulate -- f32-equivalent results.  Gradient products (dX = W^T dY, dW = dY X^T): ONE f16 MFMA on
// the hi parts (W^T from its own transposed image, dY under the workgroup's power-of-two scale); nothing in the
// path's contract holds gradients to 1e-4, the error against the oracle is reported by the tests
// (tests/test_gpu_parity.py, test_gpu_fullsize.py).
//
// LDS images (bytes):
//   forward image   32 pairs (hi, lo) of 1 KB sub-images: lane l's 16 B at 16 l (ds_read_b128, conflict-free)
//   transposed image 30 sub-images, hi only: A operand of dX = W^T dY, rows = input units, k = output units
//   bias tail       f32 accumulator start values (natural unit order: block b, lane q reads 16 B at 16b + 4q)
#pragma once
#include "render_device.h"

typedef float v4f __attribute__((ext_vector_type(4)));
typedef _Float16 t16_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 t16_h4 __attribute__((ext_vector_type(4)));
typedef _Float16 t16_h2 __attribute__((ext_vector_type(2)));
typedef float t16_f2 __attribute__((ext_vector_type(2)));

namespace scanerf {

__host__ __device__ constexpr int t16_ku(int t, int q, int j) { return 32 * t + 16 * (j >> 2) + 4 * q + (j & 3); }
// x-stash position (render.hip: 16h + 2jj + f holds feature f of level 4(jj>>1) + 2h + (jj&1)) -> decoder input index 2 level + f
__host__ __device__ constexpr int t16_pos_to_input(int pos)
{
    const int h = pos >> 4, jj = (pos & 15) >> 1, f = pos & 1;
    return 2 * (4 * (jj >> 1) + 2 * h + (jj & 1)) + f;
}
// row m of dX block e of the first layer (input "unit" 16e + m) <-> x-stash position: lane (c, q) of block e holds
// register g = position 8q + 4e + g, i.e. exactly the 8 positions 8q .. 8q+7 the lane loaded
__host__ __device__ constexpr int t16_l0_row_to_pos(int e, int m) { return 8 * (m >> 2) + 4 * e + (m & 3); }

// ---- operands
struct T16HL {
    t16_h8 hi, lo;
};
// k-step operand from two accumulator blocks (slots j < 4 from `a`, j >= 4 from `b`), split hi + lo
//
// T16_MIX_SPLIT (default): lo = (f16)(x - (float)hi) as ONE instruction per element -- v_fma_mixlo_f16 / v_fma_mixhi_f16 take the
// f16 hi part as it is, multiply by -1.0, add the f32 x and round the (exact) f32 difference to f16 into one half of the
// destination -- instead of v_cvt_f32_f16 + v_sub_f32 per element and a v_cvt_pk_f16_f32 per pair: 12 instead of 24 vector
// instructions per 8-element operand, 29 operands per tile.  The same two roundings as the plain form (the difference is exact in
// f32), so the same bits.  The compiler does not select these forms by itself (it folds fma(h, -1, x) back into a subtraction, and
// prefers cvt_pk for a pair of truncations), hence inline asm; what the assembler cannot know is that a matrix instruction must not
// read a register within two wait states of the vector instruction that wrote it (the hazard recogniser pads that case for
// instructions it sees: `s_nop` after a v_mov feeding an MFMA), so the lo registers pass through one `s_nop 1` tied to them.
#ifndef T16_MIX_SPLIT
#define T16_MIX_SPLIT 1
#endif
__device__ __forceinline__ T16HL t16_split(const v4f &a, const v4f &b)
{
    T16HL o;
#if T16_MIX_SPLIT
    uint32_t lo32[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const t16_f2 x = p < 2 ? t16_f2{ a[2 * p], a[2 * p + 1] } : t16_f2{ b[2 * p - 4], b[2 * p - 3] };
        const t16_h2 hi = __builtin_convertvector(x, t16_h2);
        const uint32_t hb = __builtin_bit_cast(uint32_t, hi);
        uint32_t lb;
        asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lb) : "v"(hb), "v"(x[0]));
        asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lb) : "v"(hb), "v"(x[1]));
        lo32[p] = lb;
        o.hi[2 * p] = hi[0];
        o.hi[2 * p + 1] = hi[1];
    }
    asm("s_nop 1" : "+v"(lo32[0]), "+v"(lo32[1]), "+v"(lo32[2]), "+v"(lo32[3]));   // (see above: two wait states before an MFMA may read them)
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const t16_h2 lo = __builtin_bit_cast(t16_h2, lo32[p]);
        o.lo[2 * p] = lo[0];
        o.lo[2 * p + 1] = lo[1];
    }
#else
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const t16_f2 x = p < 2 ? t16_f2{ a[2 * p], a[2 * p + 1] } : t16_f2{ b[2 * p - 4], b[2 * p - 3] };
        const t16_h2 hi = __builtin_convertvector(x, t16_h2);
        const t16_f2 back = __builtin_convertvector(hi, t16_f2);
        const t16_f2 r = { x[0] - back[0], x[1] - back[1] };
        const t16_h2 lo = __builtin_convertvector(r, t16_h2);
        o.hi[2 * p] = hi[0];
        o.hi[2 * p + 1] = hi[1];
        o.lo[2 * p] = lo[0];
        o.lo[2 * p + 1] = lo[1];
    }
#endif
#if SCANERF_GUARDS
    asm volatile("s_nop 1" : "+v"(o.hi), "+v"(o.lo));  // operand guard (render_h3.h, "operand hazard")
#endif
    return o;
}
// hi part only (gradient operands)
__device__ __forceinline__ t16_h8 t16_hi(const v4f &a, const v4f &b)
{
    t16_h8 o;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const t16_f2 x = p < 2 ? t16_f2{ a[2 * p], a[2 * p + 1] } : t16_f2{ b[2 * p - 4], b[2 * p - 3] };
        const t16_h2 hi = __builtin_convertvector(x, t16_h2);
        o[2 * p] = hi[0];
        o[2 * p + 1] = hi[1];
    }
#if SCANERF_GUARDS
    asm volatile("s_nop 1" : "+v"(o));
#endif
    return o;
}
__device__ __forceinline__ t16_h4 t16_hi4(const v4f &a)
{
    const t16_h2 p0 = __builtin_convertvector(t16_f2{ a[0], a[1] }, t16_h2), p1 = __builtin_convertvector(t16_f2{ a[2], a[3] }, t16_h2);
    return t16_h4{ p0[0], p0[1], p1[0], p1[1] };
}

// ---- row operations: a "row" = the 16 lanes of a group (lane & 15 = sample) = one DPP row.  One vector instruction each,
// where __shfl_* within 16 lanes goes through the LDS crossbar (ds_bpermute: an LDS round trip per step of a dependent scan).
// (tools/probe/dpp_rows_test.hip checks the three against a host evaluation.)
template <int N>
__device__ __forceinline__ float row_shr(float v, float fill)   // lane c <- v of lane c - N (fill where c < N)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, fill), __builtin_bit_cast(int, v), 0x110 + N, 0xf, 0xf, false));
}
template <int N>
__device__ __forceinline__ float row_shl(float v, float fill)   // lane c <- v of lane c + N (fill where c + N > 15)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, fill), __builtin_bit_cast(int, v), 0x100 + N, 0xf, 0xf, false));
}
template <int N>
__device__ __forceinline__ float row_ror(float v)               // lane c <- v of lane (c - N) mod 16
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + N, 0xf, 0xf, false));
}

__device__ __forceinline__ v4f t16_mfma(const t16_h8 &a, const t16_h8 &b, const v4f &c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
// Rounds 1-2 made every group of MFMAs a closed scheduling region followed by wait states (1e-5 .. 3e-3 of the forward's tiles
// came out wrong in sample columns 16-31 otherwise).  Compiled out since round 3 (common.h SCANERF_GUARDS): without packed-f32
// instructions in the kernel (-fno-slp-vectorize) nothing differs between launches with the MFMAs scheduled freely.
#if SCANERF_GUARDS
#define T16_REGION_BEGIN() __builtin_amdgcn_sched_barrier(0)
#define T16_REGION_END()       \
    asm volatile("s_nop 3");   \
    __builtin_amdgcn_sched_barrier(0)
#else
#define T16_REGION_BEGIN()
#define T16_REGION_END()
#endif

// forward layer: u[b] += W[b] X over KS k-steps (three products per term, small ones first).  `img` + `base` =
// the layer's first pair, `lo16` = 16 * lane.
template <int NB, int KS>
__device__ __forceinline__ void t16_layer(v4f u[NB], const char *img, int base, int lo16, const T16HL B[KS])
{
#ifndef T16_GROUP
#define T16_GROUP 2
#endif
    constexpr int G = NB < T16_GROUP ? NB : T16_GROUP;  // blocks per group: their A operands (hi, lo) are the registers in flight
#pragma unroll
    for (int b0 = 0; b0 < NB; b0 += G)
#pragma unroll
        for (int t = 0; t < KS; ++t) {
            t16_h8 ahi[G], alo[G];
#pragma unroll
            for (int b = 0; b < G; ++b) {
                const char *p = img + base + ((b0 + b) * KS + t) * T16_PAIR + lo16;
                ahi[b] = *reinterpret_cast<const t16_h8 *>(p);
                alo[b] = *reinterpret_cast<const t16_h8 *>(p + T16_SUB);
            }
            T16_REGION_BEGIN();
#pragma unroll
            for (int b = 0; b < G; ++b) u[b0 + b] = t16_mfma(alo[b], B[t].hi, u[b0 + b]);
#pragma unroll
            for (int b = 0; b < G; ++b) u[b0 + b] = t16_mfma(ahi[b], B[t].lo, u[b0 + b]);
#pragma unroll
            for (int b = 0; b < G; ++b) u[b0 + b] = t16_mfma(ahi[b], B[t].hi, u[b0 + b]);
            T16_REGION_END();
        }
}
// transposed product: dx[b_in] += W^T[b_in] dY over KS k-steps of the output units (hi parts only)
template <int NBI, int KS>
__device__ __forceinline__ void t16_chain(v4f dx[NBI], const char *img, int base, int lo16, const t16_h8 dY[KS])
{
#pragma unroll
    for (int t = 0; t < KS; ++t) {
        t16_h8 a[NBI];
#pragma unroll
        for (int b = 0; b < NBI; ++b) a[b] = *reinterpret_cast<const t16_h8 *>(img + base + (b * KS + t) * T16_SUB + lo16);
        T16_REGION_BEGIN();
#pragma unroll
        for (int b = 0; b < NBI; ++b) dx[b] = t16_mfma(a[b], dY[t], dx[b]);
        T16_REGION_END();
    }
}
// ---- "t16s": the same layers with the gradient products split as well (hi + lo of W^T and of dY, three MFMAs per term) ----
// A operand of a forward product from the swizzled image: `p` = sub-image + L.pos8
__device__ __forceinline__ t16_h8 s16_lda(const char *p)
{
    const t16_h4 a = *reinterpret_cast<const t16_h4 *>(p), b = *reinterpret_cast<const t16_h4 *>(p + 512);
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}
template <int NB, int KS>
__device__ __forceinline__ void s16_layer(v4f u[NB], const char *img, int base, int pos8, const T16HL B[KS])
{
    constexpr int G = NB < T16_GROUP ? NB : T16_GROUP;
#pragma unroll
    for (int b0 = 0; b0 < NB; b0 += G)
#pragma unroll
        for (int t = 0; t < KS; ++t) {
            t16_h8 ahi[G], alo[G];
#pragma unroll
            for (int b = 0; b < G; ++b) {
                const char *p = img + base + ((b0 + b) * KS + t) * T16_PAIR + pos8;
                ahi[b] = s16_lda(p);
                alo[b] = s16_lda(p + T16_SUB);
            }
            T16_REGION_BEGIN();
#pragma unroll
            for (int b = 0; b < G; ++b) u[b0 + b] = t16_mfma(alo[b], B[t].hi, u[b0 + b]);
#pragma unroll
            for (int b = 0; b < G; ++b) u[b0 + b] = t16_mfma(ahi[b], B[t].lo, u[b0 + b]);
#pragma unroll
            for (int b = 0; b < G; ++b) u[b0 + b] = t16_mfma(ahi[b], B[t].hi, u[b0 + b]);
            T16_REGION_END();
        }
}
__device__ __forceinline__ t16_h4 t16_tr4(const char *p);
// Transposed product dx[b_in] += W^T[b_in] dY from the FORWARD image of the layer (`base`, KSIN k-steps per output block), over
// KS k-steps of its output units.  A operand of (b_in, t): lane (m, q) needs W[n = 32t + 16(j >> 2) + 4q + (j & 3)][i = 16 b_in + m]:
// rows n = 32t + 4q + r (r = 0..3) of output block 2t, then of block 2t + 1; in the image those are lanes (m' = 4q + r, q' = p)
// of the pair (block, input k-step b_in >> 1), half b_in & 1 -- 8 bytes each, which ds_read_b64_tr_b16 delivers transposed.
template <int NBI, int KS, int KSIN>
__device__ __forceinline__ void s16_chain(v4f dx[NBI], const char *img, int base, int trp, const T16HL dY[KS])
{
#pragma unroll
    for (int t = 0; t < KS; ++t)
#pragma unroll
        for (int bi = 0; bi < NBI; ++bi) {
            const char *a0 = img + base + ((2 * t) * KSIN + (bi >> 1)) * T16_PAIR + (bi & 1) * 512 + trp;
            const char *a1 = a0 + KSIN * T16_PAIR;
            const t16_h8 ahi = __builtin_shufflevector(t16_tr4(a0), t16_tr4(a1), 0, 1, 2, 3, 4, 5, 6, 7);
            const t16_h8 alo = __builtin_shufflevector(t16_tr4(a0 + T16_SUB), t16_tr4(a1 + T16_SUB), 0, 1, 2, 3, 4, 5, 6, 7);
            dx[bi] = t16_mfma(alo, dY[t].hi, dx[bi]);
            dx[bi] = t16_mfma(ahi, dY[t].lo, dx[bi]);
            dx[bi] = t16_mfma(ahi, dY[t].hi, dx[bi]);
#ifndef T16_FREE_CHAIN   // (-DT16_FREE_CHAIN: experiment -- no bound on the operands in flight)
            if (bi & 1) __builtin_amdgcn_sched_barrier(0);   // (two input blocks' operands in flight at a time)
#endif
        }
}
// ... of a narrow layer (heads, rgb): its own transposed pairs (hi, lo), lane l's 16 B at 16 l, one k-step
template <int NBI>
__device__ __forceinline__ void s16_chain_narrow(v4f dx[NBI], const char *img, int base, int lo16, const T16HL &dY)
{
#pragma unroll
    for (int b = 0; b < NBI; ++b) {
        const char *p = img + base + b * T16_PAIR + lo16;
        const t16_h8 ahi = *reinterpret_cast<const t16_h8 *>(p), alo = *reinterpret_cast<const t16_h8 *>(p + T16_SUB);
        T16_REGION_BEGIN();
        dx[b] = t16_mfma(alo, dY.hi, dx[b]);
        dx[b] = t16_mfma(ahi, dY.lo, dx[b]);
        dx[b] = t16_mfma(ahi, dY.hi, dx[b]);
        T16_REGION_END();
    }
}

__device__ __forceinline__ v4f t16_ld4(const char *img, int byte_off)
{
    const float4 v = *reinterpret_cast<const float4 *>(img + byte_off);
    return v4f{ v.x, v.y, v.z, v.w };
}
__device__ __forceinline__ v4f t16_bias(const char *img, int layer, int b, int q, int bias_base = T16_BIAS)
{
    return t16_ld4(img, bias_base + (layer * 64 + 16 * b + 4 * q) * 4);
}

// ---- staging image of one wave's tile for the sample-reduction products (dW = dY X^T): a matrix of 64 units x 16
// samples of f16 kept as [sample s][16 chunks of 4 units] with chunk position  chunk ^ g(s),
//     g(s) = (s1 << 3) | (s3 << 2) | (s2 << 1) | s0        (s3..s0 = bits of s)
// The writer (lane = sample c, its 4 registers of block b = chunk 4b + q: one ds_write_b64) hits 16 distinct
// chunk positions per 16-lane group; the owner's transposed read (ds_read_b64_tr_b16: 4 samples x 16 units per
// 16-lane group, lane 4a+p supplies sample a's chunk 4b'+p) hits 32 distinct bank pairs per half-wave.
constexpr int T16_STAGE_MAT = 16 * 128;       // bytes per matrix (Y or X)
constexpr int T16_STAGE_WAVE = 2 * T16_STAGE_MAT;
__host__ __device__ constexpr int t16_g(int s) { return (((s >> 1) & 1) << 3) | (((s >> 3) & 1) << 2) | (((s >> 2) & 1) << 1) | (s & 1); }

typedef short t16_s4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) t16_s4 t16_lds_s4;
__device__ __forceinline__ t16_h4 t16_tr4(const char *p)
{
    const t16_s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((t16_lds_s4 *)(p));
    return __builtin_bit_cast(t16_h4, v);
}
struct T16Lane {
    int lo16;   // 16 * lane: this lane's 16 B of an operand sub-image
    int w1;     // staging write: block b of this lane's sample goes to  mat + (w1 ^ (32 b))
    int r1, r2; // staging reads: operand of unit block b', tile pair P, matrix M:  tr(P*2*WAVE + M*MAT + (r1 ^ (32 b'))) ++ tr(... r2 ...)
    int pos8;   // t16s image: this lane's 8-byte slot of a half sub-image (s16_pos)
    int trp;    // t16s image: the slot this lane ADDRESSES in a transposed read (s16_lda_T)
};
// t16s image: lane l's two 8-byte halves of a sub-image sit at  s16_pos(l) * 8  and  512 + s16_pos(l) * 8.  The XOR makes
// both walks of the image conflict-free for 8-byte accesses (32 lanes = 64 banks per pass): the forward's (lanes 0-31 /
// 32-63 each cover one 256-byte run) and the transposed one (the 32 lanes of a pass address the slots of lanes
// {4q + r + 16p: q in a pair of groups, r, p = 0..3}, which the XOR spreads over 32 distinct slots mod 32).
__host__ __device__ constexpr int s16_pos(int l) { return l ^ ((l >> 5) << 3); }
__device__ __forceinline__ T16Lane t16_lane(int lane, int stage_wave = T16_STAGE_WAVE)
{
    T16Lane L;
    const int c = lane & 15, q = lane >> 4, a = (lane >> 2) & 3, p = lane & 3;
    L.lo16 = lane * 16;
    L.w1 = c * 128 + ((q ^ t16_g(c)) << 3);
    const int s1 = 8 * (q & 1) + a, s2 = s1 + 4;  // samples this lane addresses in the two transposed reads
    L.r1 = (q >> 1) * stage_wave + s1 * 128 + ((p ^ t16_g(s1)) << 3);
    L.r2 = (q >> 1) * stage_wave + s2 * 128 + ((p ^ t16_g(s2)) << 3);
    L.pos8 = s16_pos(lane) * 8;
    L.trp = s16_pos(4 * q + a + 16 * p) * 8;   // row a of the group's 4 x 16 block = image lane (m' = 4q + a, q' = p)
    return L;
}
__device__ __forceinline__ void t16_stage_put(char *mat, const T16Lane &L, int b, const t16_h4 &v)
{
    *reinterpret_cast<t16_h4 *>(mat + (L.w1 ^ (32 * b))) = v;
}
// operand (A or B alike) for units 16b' + (lane & 15), samples of tiles 2P (q < 2) and 2P+1 (q >= 2): slot (q, j) = sample 8(q&1) + j
__device__ __forceinline__ t16_h8 t16_stage_get(const char *pair_mat, const T16Lane &L, int b)
{
    const t16_h4 x0 = t16_tr4(pair_mat + (L.r1 ^ (32 * b))), x1 = t16_tr4(pair_mat + (L.r2 ^ (32 * b)));
    return __builtin_shufflevector(x0, x1, 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ float t16_sum8(const t16_h8 &v, float acc)
{
    const t16_h2 one = { (_Float16)1.0f, (_Float16)1.0f };
#pragma unroll
    for (int p = 0; p < 4; ++p) acc = __builtin_amdgcn_fdot2(t16_h2{ v[2 * p], v[2 * p + 1] }, one, acc, false);
    return acc;
}

// launcher of the pack kernel (render_bwd_t16.hip)
int launch_pack_decoder_t16(const float *blob, const float *wf, char *out, hipStream_t st);

// one ray's split SH operand as a 64-byte row: [hi: SH 0..15 as f16][lo: the f16 residuals] -- the bits t16_split gives in registers
__device__ __forceinline__ void s16_sh_row(char *row, const float d[3], float eps)
{
    float sh[16];
    ray_sh(d, sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]), sh, eps);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const T16HL o = t16_split(v4f{ sh[8 * q], sh[8 * q + 1], sh[8 * q + 2], sh[8 * q + 3] },
                                  v4f{ sh[8 * q + 4], sh[8 * q + 5], sh[8 * q + 6], sh[8 * q + 7] });
        *reinterpret_cast<t16_h8 *>(row + 16 * q) = o.hi;
        *reinterpret_cast<t16_h8 *>(row + 32 + 16 * q) = o.lo;
    }
}

// ---- the decoder FORWARD on one 16-sample tile from the t16s image (the forward recompute of render_bwd_t16.hip / decoder.hip's
// backward as a function): lane (c, q) holds decoder inputs 16 (q & 1) + 4 (q >> 1) + {0..3} in xa and those + 8 in xb, and the
// sample's view direction; the outputs are valid in the lanes with q == 0.  ~95 VGPRs: four waves per SIMD.
// SH_ROW: the split SH operand of the sample's ray is read from `sh_row` (this lane's 16 bytes of the hi part; the lo part 32
// bytes further: a 64-byte row [hi 16][lo 16] per ray written by s16_sh_row below; lanes with q >= 2 point at a row of zeros)
// instead of being evaluated from d -- per tile the harmonics cost ~100 vector instructions for 16 samples, four lanes each.
// `gate(sigma)` (wave-uniform) is asked after the heads whether the directional half is needed at all: the render-time kernel
// answers no when every live sample of the tile has an opacity of exactly zero (1 - exp(-sigma delta) == 0: empty space inside
// occupied cells) -- their colours are multiplied by that zero, so the three directional layers (54 of the tile's 96 MFMAs and
// half its vector work) change nothing.  specular is then returned as zero.
struct S16NoGate {
    __device__ __forceinline__ bool operator()(float) const { return true; }
};
// FOLD (render-time inference, SCANERF_INFER_FOLDED): the image was packed from a blob whose three Gaussian-activated layers
// (Spatial_MLP.mlp.0, Directional_MLP.mlp.0 / .2: weights AND biases) carry the activation's constant sqrt(50 log2 e), so
// G(u) = exp2(-(u')^2) is a multiply and an exponential instead of two multiplies and an exponential (48 activations per lane and tile).
template <bool FOLD>
__device__ __forceinline__ float s16_gauss(float u)
{
    if constexpr (FOLD) return __builtin_amdgcn_exp2f(-(u * u));
    else return gauss_fast(u);
}
template <bool SH_ROW = false, class Gate = S16NoGate, bool FOLD = false>
__device__ __forceinline__ SampleOut decode_tile_s16(const char *lds, int lane, const v4f &xa, const v4f &xb, const float d[3], float eps,
                                                     const char *sh_row = nullptr, Gate gate = Gate())
{
    const int q = lane >> 4, pos8 = s16_pos(lane) * 8;
    SampleOut so;
    T16HL HB[2];
    {
        const T16HL xB = t16_split(xa, xb);
        v4f act[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) act[b] = t16_bias(lds, 0, b, q, S16_BIAS);
        s16_layer<4, 1>(act, lds, T16_L0, pos8, &xB);
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int g = 0; g < 4; ++g) act[b][g] = s16_gauss<FOLD>(act[b][g]);
        const T16HL aB[2] = { t16_split(act[0], act[1]), t16_split(act[2], act[3]) };
        v4f hh[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) hh[b] = t16_bias(lds, 1, b, q, S16_BIAS);
        s16_layer<4, 2>(hh, lds, T16_L1, pos8, aB);
        HB[0] = t16_split(hh[0], hh[1]);
        HB[1] = t16_split(hh[2], hh[3]);
    }
    {
        v4f hd[2] = { t16_ld4(lds, S16_BIAS + 256 * 4), t16_ld4(lds, S16_BIAS + 260 * 4) };
        s16_layer<2, 1>(hd, lds, T16_HEAD, pos8, &HB[0]);
        so.sigma = softplus_fast(hd[0][0]);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            so.dif[k] = sigmoid_fast(hd[0][1 + k]);
            so.tint[k] = sigmoid_fast(hd[1][k]);
        }
    }
    if (!gate(so.sigma)) {
        so.spec[0] = so.spec[1] = so.spec[2] = 0.0f;
        return so;
    }
    T16HL cB[2];
    {
        T16HL shB;
        if constexpr (SH_ROW) {
            shB.hi = *reinterpret_cast<const t16_h8 *>(sh_row);
            shB.lo = *reinterpret_cast<const t16_h8 *>(sh_row + 32);
        } else {
            const float dnorm = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
            float sh[16];
            ray_sh(d, dnorm, sh, eps);
            v4f s0, s1;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s0[j] = q == 0 ? sh[j] : (q == 1 ? sh[8 + j] : 0.0f);
                s1[j] = q == 0 ? sh[4 + j] : (q == 1 ? sh[12 + j] : 0.0f);
            }
            shB = t16_split(s0, s1);
        }
        v4f v0[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) v0[b] = t16_bias(lds, 2, b, q, S16_BIAS);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const char *p1 = lds + T16_D0 + (b * 2 + 1) * T16_PAIR + pos8, *p0 = lds + T16_D0 + (b * 2) * T16_PAIR + pos8;
            const t16_h8 shi = s16_lda(p1), slo = s16_lda(p1 + T16_SUB);
            const t16_h8 ahi = s16_lda(p0), alo = s16_lda(p0 + T16_SUB);
            v0[b] = t16_mfma(slo, shB.hi, v0[b]);
            v0[b] = t16_mfma(shi, shB.lo, v0[b]);
            v0[b] = t16_mfma(shi, shB.hi, v0[b]);
            v0[b] = t16_mfma(alo, HB[1].hi, v0[b]);
            v0[b] = t16_mfma(ahi, HB[1].lo, v0[b]);
            v0[b] = t16_mfma(ahi, HB[1].hi, v0[b]);
        }
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int g = 0; g < 4; ++g) v0[b][g] = s16_gauss<FOLD>(v0[b][g]);
        cB[0] = t16_split(v0[0], v0[1]);
        cB[1] = t16_split(v0[2], v0[3]);
    }
    {
        v4f v1[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) v1[b] = t16_bias(lds, 3, b, q, S16_BIAS);
        s16_layer<4, 2>(v1, lds, T16_D1, pos8, cB);
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int g = 0; g < 4; ++g) v1[b][g] = s16_gauss<FOLD>(v1[b][g]);
        cB[0] = t16_split(v1[0], v1[1]);
        cB[1] = t16_split(v1[2], v1[3]);
    }
    {
        v4f r[1] = { t16_ld4(lds, S16_BIAS + 264 * 4) };
        s16_layer<1, 2>(r, lds, T16_D2, pos8, cB);
#pragma unroll
        for (int k = 0; k < 3; ++k) so.spec[k] = sigmoid_fast(r[0][k]);
    }
    return so;
}

}  // namespace scanerf

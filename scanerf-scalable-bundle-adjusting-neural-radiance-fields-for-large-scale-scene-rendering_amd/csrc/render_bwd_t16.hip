// render_bwd_t16.hip -- fused per-ray volume rendering, backward, 16-sample tiles on v_mfma_f32_16x16x32_f16:
// 8 waves per workgroup, TWO WAVES PER SIMD (render_t16.h explains why and gives the lane maps and images).
//
// Same adjoint as render_bwd_h3.hip (hashgrid/__init__.py:512-596 under autograd in the reference).  Structure:
//   * one wave per ray, its 16-sample tiles walked last -> first (compositing adjoint = register carry);
//   * the 8 waves of a workgroup walk their rays in lock step; every weight-gradient block (16 x 16) is OWNED by one
//     wave, which sums it over the 8 waves' tiles (two tiles = one K = 32 MFMA) from their staged operands
//     (28 accumulator registers per wave); one partial row per workgroup, reduced by k_reduce_dw;
//   * forward recompute in split-f16 (three products per term); what the backward steps need from it is kept as
//     f16 (hi parts of the activations, G'(u) of the three Gaussian layers): 48 registers, nothing is recomputed twice;
//   * gradient products on ONE f16 MFMA per term under the workgroup's power-of-two scale (render_bwd_h3.hip) -- "t16"; or,
//     SPLIT = true ("t16s", the library's default and bench.py's headline): every gradient product split hi + lo like the
//     recompute, pre-activations held in f32 and activations formed again where they are used: f32-equivalent gradients;
//   * the table-gradient scatter records are emitted here (scatter.hip), 4 levels per lane; t16s: waves 0-3 at the tile's end,
//     waves 4-7 from a copy of dX parked in LDS, behind the next tile's compositing ("skewed emission", below);
//   * POSE = true: also the per-ray sums of the pose refinement (g_dnorm, g_rowsum) and, from the forward's Jacobian stash
//     (render_device.h jst_pack), dL/d(rays_o), dL/d(rays_d) through the sample positions.
// Needs the forward's x-stash (without one the caller falls back to render_bwd_h3.hip, which re-gathers).
// What bounds it, measured: DESIGN.md 4.2a (cycle stamps: -DT16_STAMPS + tools/bwd_stamps.py).
#include <stdlib.h>

#include "render_bwd_common.h"
#include "render_t16.h"

using namespace scanerf;

// timing experiments only (results are wrong): -DT16_NO_BARRIER drops the per-step workgroup barriers, -DT16_NO_WGRAD the
// weight-gradient products, -DT16_NO_EMIT the records / dfeat
#ifdef T16_NO_BARRIER
#define STEP_BARRIER() __builtin_amdgcn_sched_barrier(0)
#else
#define STEP_BARRIER() __syncthreads()
#endif

// -DT16_STAMPS (investigation builds, tools/build_variant.py; timing only): every wave sums the shader cycles it spends in each
// interval of the tile loop (23 intervals: emission, forward recompute, compositing, and for each of the 10 workgroup barriers the
// work before it and the wait in it) and writes the sums to row gridDim.x + blockIdx.x of dw_partial (32 floats per wave;
// tools/bwd_stamps.py reads them).  s_memtime needs lgkmcnt(0): the stamps sit where the kernel drains the LDS anyway (barriers).
#ifdef T16_STAMPS
#define STAMP(i)                                                             \
    do {                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                   \
        const uint32_t now_ = (uint32_t)__builtin_amdgcn_s_memtime();        \
        stamps[i] += now_ - tlast;                                           \
        tlast = now_;                                                        \
        __builtin_amdgcn_sched_barrier(0);                                   \
    } while (0)
#else
#define STAMP(i)
#endif

// where the t16s kernel issues the NEXT tile's input loads: 0 = at the tile's end (before the record stores), 2..5 = right behind
// barrier A2..A5 (earlier: the loads of the second wave of a SIMD otherwise queue behind the first wave's record stores)
// (measured, plan + backward at configs[1], same box: 0: 5.53-5.56 ms, 5: 5.45, 4: 5.44-5.45, 3: 5.42-5.48)
#ifndef T16_LOAD_AT
#define T16_LOAD_AT 4
#endif

// -DT16_MIN_WAVES=3 (investigation builds): the register budget of THREE waves per SIMD (168) for the same 8-wave workgroup -- what
// the code would have to fit before a 12-wave workgroup could share one image (profiles/r06_bwd_3waves.txt)
#ifndef T16_MIN_WAVES
#define T16_MIN_WAVES 2
#endif

namespace {

#ifdef T16_LIBM_HEADS   // A/B only (tools/ab_heads.sh): the forms of rounds 1-5 -- libm's log1pf / expf and an IEEE division on the tile's chain
#define T16_SOFTPLUS(x) softplus_(x)
#define T16_DSOFTPLUS(x) ((x) > 20.0f ? 1.0f : sigmoid_fast(x))
#define T16_EXPNEG(x) expf(-(x))
#define T16_DIV(a, b) ((a) / (b))
#else
#define T16_SOFTPLUS(x) softplus_fast(x)
#define T16_DSOFTPLUS(x) sigmoid_fast(x)
#define T16_EXPNEG(x) __builtin_amdgcn_exp2f(-1.4426950408889634f * (x))
#define T16_DIV(a, b) ((a) * __builtin_amdgcn_rcpf(b))
#endif
constexpr int kThreads = 512;
constexpr int kWaves = 8;
// LDS carve (SPLIT = the t16s variant: its own images, staging of hi AND lo parts)
template <bool SPLIT>
struct Lds {
    static constexpr int kImg = SPLIT ? S16_BYTES : T16_BYTES;
    static constexpr int kBias = SPLIT ? S16_BIAS : T16_BIAS;
    static constexpr int kStageWave = SPLIT ? 2 * T16_STAGE_WAVE : T16_STAGE_WAVE;   // {Y, X} (, {Y lo, X lo})
    static constexpr int kRes = kImg;                             // resolutions [16][4] i32
    static constexpr int kDinit = kRes + 256;                     // 8 waves x 64 f32: Dir layer-0 accumulator start of the wave's ray (unit order)
    static constexpr int kSh = kDinit + kWaves * 256;             // 8 waves x SH[16]
    static constexpr int kMx = kSh + kWaves * 64;                 // 8 floats
    static constexpr int kStage = kMx + 64;                       // 8 waves x staging
    static constexpr int kPark = kStage + kWaves * kStageWave;    // (SPLIT, a.park) waves 4-7: a tile's dX, 64 lanes x 32 B per wave
    static constexpr int kParkBytes = 4 * 2048;
    static constexpr int kCursor = kPark;                         // record cursors (fused scatter producer only); + kParkBytes when parking
    static_assert(kStage % 16 == 0 && kCursor % 16 == 0, "LDS carve alignment");
};

// ------------------------------------------------------------------ pack: blob -> t16 / t16s images
__device__ __forceinline__ float pk_W(const float *blob, int base, int n_out, int n, int k) { return blob[base + n_out + k * n_out + n]; }
// element j of lane (m, q) of forward pair `pair`: W[n][unit of slot (q, j)]
__device__ __forceinline__ float t16_fwd_weight(const float *blob, const float *wf, int pair, int lane, int j)
{
    const int m = lane & 15, q = lane >> 4;
    int pp = pair, layer, ks;
    if (pp < 4) { layer = 0; ks = 1; }
    else if ((pp -= 4) < 8) { layer = 1; ks = 2; }
    else if ((pp -= 8) < 2) { layer = 2; ks = 1; }
    else if ((pp -= 2) < 8) { layer = 3; ks = 2; }
    else if ((pp -= 8) < 8) { layer = 4; ks = 2; }
    else { pp -= 8; layer = 5; ks = 2; }
    const int b = pp / ks, t = pp % ks;
    const int n = 16 * b + m, ku = t16_ku(t, q, j);
    float w = 0.0f;
    if (layer == 0) {
        const int k = t16_pos_to_input(8 * q + j);
        w = pk_W(blob, BLOB_S0, 64, n, k) * wf[k];
    } else if (layer == 1) w = pk_W(blob, BLOB_S1, 64, n, ku);
    else if (layer == 2) {  // heads on H[:32] (k-step 0 of H): rows replicated in every q
        const int r = m & 3;
        if (b == 0) w = r == 0 ? pk_W(blob, BLOB_SIG, 1, 0, ku) : pk_W(blob, BLOB_DIF, 3, r - 1, ku);
        else if (r < 3) w = pk_W(blob, BLOB_TINT, 3, r, ku);
    } else if (layer == 3) {
        if (t == 0) w = pk_W(blob, BLOB_D0, 64, n, ku);                      // input k = H[32 + k], slot unit ku(0, q, j) - 0
        else if (q < 2) w = pk_W(blob, BLOB_D0, 64, n, 32 + 8 * q + j);      // SH part
    } else if (layer == 4) w = pk_W(blob, BLOB_D1, 64, n, ku);
    else {
        const int r = m & 3;
        if (r < 3) w = pk_W(blob, BLOB_D2, 3, r, ku);
    }
    return w;
}
// element j of lane (m, q) of a TRANSPOSED sub-image of `layer` (input block bi, k-step t of the layer's output units):
// W[n = ku(t, q, j)][i = 16 bi + m]
__device__ __forceinline__ float t16_T_weight(const float *blob, const float *wf, int layer, int bi, int t, int lane, int j)
{
    const int m = lane & 15, q = lane >> 4;
    const int n = t16_ku(t, q, j), i = 16 * bi + m;
    float w = 0.0f;
    if (layer == 5) {           // narrow rows 8..10 = rgb
        if (n >= 8 && n < 11) w = pk_W(blob, BLOB_D2, 3, n - 8, i);
    } else if (layer == 4) w = pk_W(blob, BLOB_D1, 64, n, i);
    else if (layer == 3) w = pk_W(blob, BLOB_D0, 64, n, i);      // i = 16 b_in + m in 0..31 <-> H[32 + i]
    else if (layer == 2) {      // narrow rows 0..6 = sigma, dif, tint; i = H unit 0..31
        if (n == 0) w = pk_W(blob, BLOB_SIG, 1, 0, i);
        else if (n < 4) w = pk_W(blob, BLOB_DIF, 3, n - 1, i);
        else if (n < 7) w = pk_W(blob, BLOB_TINT, 3, n - 4, i);
    } else if (layer == 1) w = pk_W(blob, BLOB_S1, 64, n, i);
    else {
        const int k = t16_pos_to_input(t16_l0_row_to_pos(bi, m));
        w = pk_W(blob, BLOB_S0, 64, n, k) * wf[k];
    }
    return w;
}
__device__ __forceinline__ float t16_tail_value(const float *blob, int t)   // f32 tail: [L0 64][L1 64][D0 64][D1 64][headA 4][headB 4][D2 4][pad 4]
{
    if (t < 256) {
        const int bases[4] = { BLOB_S0, BLOB_S1, BLOB_D0, BLOB_D1 };
        return blob[bases[t >> 6] + (t & 63)];
    }
    if (t < 260) return t == 256 ? blob[BLOB_SIG] : blob[BLOB_DIF + t - 257];
    if (t < 264) return t < 263 ? blob[BLOB_TINT + t - 260] : 0.0f;
    if (t < 268) return t < 267 ? blob[BLOB_D2 + t - 264] : 0.0f;
    return 0.0f;
}

__global__ void __launch_bounds__(256) k_pack_decoder_t16(const float *__restrict__ blob, const float *__restrict__ wf,
                                                          char *__restrict__ out)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < 32 * 512) {  // forward image: one thread per (pair, lane, j)
        const int pair = e >> 9, lane = (e >> 3) & 63, j = e & 7;
        const float w = t16_fwd_weight(blob, wf, pair, lane, j);
        const _Float16 hi = (_Float16)w, lo = (_Float16)(w - (float)hi);
        char *p = out + pair * T16_PAIR + lane * 16 + j * 2;
        *reinterpret_cast<_Float16 *>(p) = hi;
        *reinterpret_cast<_Float16 *>(p + T16_SUB) = lo;
    } else if (e < 32 * 512 + 30 * 512) {  // transposed image (hi only)
        const int f = e - 32 * 512;
        const int sub = f >> 9, lane = (f >> 3) & 63, j = f & 7;
        int ss = sub, layer, ks;
        if (ss < 4) { layer = 5; ks = 1; }
        else if ((ss -= 4) < 8) { layer = 4; ks = 2; }
        else if ((ss -= 8) < 4) { layer = 3; ks = 2; }
        else if ((ss -= 4) < 2) { layer = 2; ks = 1; }
        else if ((ss -= 2) < 8) { layer = 1; ks = 2; }
        else { ss -= 8; layer = 0; ks = 2; }
        const float w = t16_T_weight(blob, wf, layer, ss / ks, ss % ks, lane, j);
        *reinterpret_cast<_Float16 *>(out + T16_FWD_BYTES + sub * T16_SUB + lane * 16 + j * 2) = (_Float16)w;
    } else if (e < 62 * 512 + 272) {
        const int t = e - 62 * 512;
        reinterpret_cast<float *>(out + T16_BIAS)[t] = t16_tail_value(blob, t);
    }
}

// t16s images (render_common.h S16_*): the forward pairs in the swizzled two-half layout (render_t16.h s16_pos), the narrow
// layers' transposed sub-images as (hi, lo) pairs, the f32 tail
__global__ void __launch_bounds__(256) k_pack_decoder_s16(const float *__restrict__ blob, const float *__restrict__ wf,
                                                          char *__restrict__ out)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < 32 * 512) {
        const int pair = e >> 9, lane = (e >> 3) & 63, j = e & 7;
        const float w = t16_fwd_weight(blob, wf, pair, lane, j);
        const _Float16 hi = (_Float16)w, lo = (_Float16)(w - (float)hi);
        char *p = out + pair * T16_PAIR + (j >> 2) * 512 + s16_pos(lane) * 8 + (j & 3) * 2;
        *reinterpret_cast<_Float16 *>(p) = hi;
        *reinterpret_cast<_Float16 *>(p + T16_SUB) = lo;
    } else if (e < 32 * 512 + 6 * 512) {
        const int f = e - 32 * 512;
        const int sub = f >> 9, lane = (f >> 3) & 63, j = f & 7;
        const float w = sub < 4 ? t16_T_weight(blob, wf, 5, sub, 0, lane, j) : t16_T_weight(blob, wf, 2, sub - 4, 0, lane, j);
        const _Float16 hi = (_Float16)w, lo = (_Float16)(w - (float)hi);
        char *p = out + S16T_D2 + sub * T16_PAIR + lane * 16 + j * 2;
        *reinterpret_cast<_Float16 *>(p) = hi;
        *reinterpret_cast<_Float16 *>(p + T16_SUB) = lo;
    } else if (e < 38 * 512 + 272) {
        const int t = e - 38 * 512;
        reinterpret_cast<float *>(out + S16_BIAS)[t] = t16_tail_value(blob, t);
    }
}

// ------------------------------------------------------------------ device helpers
// A lane index the optimiser cannot trace back.  LDS addresses derived from the plain lane index are loop invariants: the
// ~60 distinct ones this kernel uses (operand slots in three 64 KB windows, bias rows, the XOR-swizzled staging slots of
// every block) get hoisted out of the tile loop and held -- or spilled -- for its whole duration.  Derived from an opaque
// copy they are recomputed where they are used (a handful of VALU per step).
__device__ __forceinline__ int fresh(int v)
{
    asm volatile("" : "+v"(v));
    return v;
}
__device__ __forceinline__ T16Lane fresh_lane(const T16Lane &L)
{
    T16Lane r;
    r.lo16 = fresh(L.lo16);
    r.w1 = fresh(L.w1);
    r.r1 = fresh(L.r1);
    r.r2 = fresh(L.r2);
    r.pos8 = fresh(L.pos8);
    r.trp = fresh(L.trp);
    return r;
}
template <int N>
__device__ __forceinline__ void zero4(v4f (&v)[N])
{
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = v4f{ 0, 0, 0, 0 };
}
// Gaussian activation of a block and its derivative factor G'(u) = -100 u G(u) as f16
__device__ __forceinline__ void act_deriv(v4f &u, t16_h4 &dg)
{
    v4f d;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float a = gauss_fast(u[g]);
        d[g] = -100.0f * u[g] * a;
        u[g] = a;
    }
    dg = t16_hi4(d);
}
__device__ __forceinline__ v4f mul_dg(const v4f &a, const t16_h4 &dg)
{
    return v4f{ a[0] * (float)dg[0], a[1] * (float)dg[1], a[2] * (float)dg[2], a[3] * (float)dg[3] };
}
__device__ __forceinline__ t16_h4 lo4(const t16_h8 &v) { return __builtin_shufflevector(v, v, 0, 1, 2, 3); }
__device__ __forceinline__ t16_h4 hi4(const t16_h8 &v) { return __builtin_shufflevector(v, v, 4, 5, 6, 7); }

// Weight-gradient blocks owned by this wave: acc[i] += sum over the 4 tile pairs of dY[yb] X[xb0 + i]^T (operands read back
// transposed from the pairs' staging images).  ROWSUM: also accumulate this lane's row sums of dY (bias gradients).
// SPLIT: hi and lo parts of both operands staged ({Y, X, Y lo, X lo} per wave), three products per term.
template <int NX, bool ROWSUM, int XSTRIDE = 1, bool SPLIT = false>
__device__ __forceinline__ void wgrad(v4f *acc, float &rowsum, const char *stage, const T16Lane &L, int yb, int x_mat_off, int xb0)
{
#ifdef T16_NO_WGRAD
    return;
#endif
    constexpr int kWave = SPLIT ? 2 * T16_STAGE_WAVE : T16_STAGE_WAVE, kLo = 2 * T16_STAGE_MAT;
#pragma unroll
    for (int P = 0; P < 4; ++P) {
        const char *pm = stage + P * 2 * kWave;
        const t16_h8 a = t16_stage_get(pm, L, yb);
        t16_h8 b[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) b[i] = t16_stage_get(pm + x_mat_off, L, xb0 + i * XSTRIDE);
        if (ROWSUM) rowsum = t16_sum8(a, rowsum);
        if constexpr (SPLIT) {
            const t16_h8 alo = t16_stage_get(pm + kLo, L, yb);
            if (ROWSUM) rowsum = t16_sum8(alo, rowsum);
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                const t16_h8 blo = t16_stage_get(pm + x_mat_off + kLo, L, xb0 + i * XSTRIDE);
                acc[i] = t16_mfma(alo, b[i], acc[i]);
                acc[i] = t16_mfma(a, blo, acc[i]);
                acc[i] = t16_mfma(a, b[i], acc[i]);
            }
#ifndef T16_FREE_WGRAD   // (-DT16_FREE_WGRAD: experiment -- let the scheduler hoist the next pair's operand reads over this pair's products)
            __builtin_amdgcn_sched_barrier(0);   // (bounds the operands in flight: the pairs' reads are not hoisted over each other)
#endif
        } else {
            T16_REGION_BEGIN();
#pragma unroll
            for (int i = 0; i < NX; ++i) acc[i] = t16_mfma(a, b[i], acc[i]);
            T16_REGION_END();
        }
    }
}
// a split operand (hi, lo) of this lane's sample into blocks b, b + 1 of a staged matrix and of its lo twin
__device__ __forceinline__ void stage_put2(char *mat, const T16Lane &L, int b, const T16HL &v)
{
    t16_stage_put(mat, L, b, __builtin_shufflevector(v.hi, v.hi, 0, 1, 2, 3));
    t16_stage_put(mat, L, b + 1, __builtin_shufflevector(v.hi, v.hi, 4, 5, 6, 7));
    t16_stage_put(mat + 2 * T16_STAGE_MAT, L, b, __builtin_shufflevector(v.lo, v.lo, 0, 1, 2, 3));
    t16_stage_put(mat + 2 * T16_STAGE_MAT, L, b + 1, __builtin_shufflevector(v.lo, v.lo, 4, 5, 6, 7));
}
// G'(u) = -100 u G(u) of a block, f32, from its pre-activation (the activation is formed again: 4 exponentials against 4 more
// registers held across the weight-gradient products)
__device__ __forceinline__ v4f gauss_deriv(const v4f &u)
{
    v4f d;
#pragma unroll
    for (int g = 0; g < 4; ++g) d[g] = -100.0f * u[g] * gauss_fast(u[g]);
    return d;
}
// the Gaussian activations of a layer (64 units from their pre-activations), split and staged as blocks 0..3 of X and X lo
__device__ __forceinline__ void stage_act(char *stX, const T16Lane &L, const v4f u[4])
{
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        v4f a0, a1;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            a0[g] = gauss_fast(u[2 * t][g]);
            a1[g] = gauss_fast(u[2 * t + 1][g]);
        }
        stage_put2(stX, L, 2 * t, t16_split(a0, a1));
    }
}

// REC: format of the scatter records (scatter_common.h): 0 = Rec (16 bytes), 1 = Rec8 (8 bytes), 2 = Rec12 (12 bytes).  POSE: also the two per-ray sums the pose
// refinement needs (render_bwd_common.h g_dnorm / g_rowsum; csrc/render_bwd_h3.hip is the other kernel that produces them)
// SPLIT ("t16s"): the gradient products split as well -- dY, W^T (read transposed out of the forward image) and both operands of
// the weight gradients as hi + lo, three MFMAs per term; G'(u) in f32 (pre-activations are held, activations recomputed); f32
// records.  Same structure, same barriers; fp32-equivalent gradients (tests/test_gpu_parity.py).
template <int DT, int REC, bool POSE, bool SPLIT = false>
__global__ void __launch_bounds__(kThreads, T16_MIN_WAVES) k_render_bwd_t16(BwdArgs a)
{
    using LD = Lds<SPLIT>;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    int *lres = reinterpret_cast<int *>(lds + LD::kRes);
    uint32_t *cursor = reinterpret_cast<uint32_t *>(lds + LD::kCursor + (SPLIT && a.park ? LD::kParkBytes : 0));
    float *shbuf = reinterpret_cast<float *>(lds + LD::kSh);
    float *mxbuf = reinterpret_cast<float *>(lds + LD::kMx);
    {
        const float4 *src = reinterpret_cast<const float4 *>(a.f.packed + (SPLIT ? WS_S16 : WS_T16));
        float4 *dst = reinterpret_cast<float4 *>(lds);
        for (int i = threadIdx.x; i < LD::kImg / 16; i += kThreads) dst[i] = src[i];
        if (threadIdx.x < 64) {
            const int lv = threadIdx.x >> 2, c = threadIdx.x & 3;
            lres[threadIdx.x] = c < 3 ? a.f.resolutions[3 * lv + c] : 0;
        }
        float4 *stz = reinterpret_cast<float4 *>(lds + LD::kStage);  // finite contents wherever a step leaves a block unwritten
        for (int i = threadIdx.x; i < kWaves * LD::kStageWave / 16; i += kThreads) stz[i] = make_float4(0, 0, 0, 0);
        if (a.recs) {
            const int nbins = 16 * a.bins.NB;
            for (int i = threadIdx.x; i < nbins; i += kThreads)
                cursor[i] = a.bin_starts[i] + a.bin_rowprefix[(size_t)i * a.bins.W + blockIdx.x];
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: ray index, per-ray loads and branches go scalar
    const char *stage = lds + LD::kStage;
    char *stY = lds + LD::kStage + wv * LD::kStageWave, *stX = stY + T16_STAGE_MAT;   // (SPLIT: their lo twins 2 matrices further)
    const int S = a.f.S, nt16 = (S + 15) >> 4;
    const uint32_t plan_skip = a.recs ? __builtin_amdgcn_readfirstlane(*skip_word(a.recs)) : 0u;   // levels the plan left out
    __builtin_amdgcn_s_setreg(1 | (23 << 6), 1);  // hwreg(HW_REG_MODE, 23, 1) = FP16_OVFL: f16 conversions saturate

    // ---- ownership of the weight-gradient blocks
    const int rb = wv >> 1, cb = wv & 1;
    v4f gW_D1[2], gW_L1[2], gW_D0[2], gW_L0[1], gW_nar[1];   // gW_D0: [0] = H part (block cb), [1] = SH part (cb == 0 only)
    zero4(gW_D1); zero4(gW_L1); zero4(gW_D0); zero4(gW_L0); zero4(gW_nar);
    float gB_D1 = 0.0f, gB_L1 = 0.0f, gB_D0 = 0.0f, gB_L0 = 0.0f, gB_nar = 0.0f;  // (cb == 0 / wave 0) lane = unit, this lane's 8 samples of every pair
    float gmax = 0.0f;
    int K = 0;                             // gradient scale 2^K of the workgroup (identical in its 8 waves)
    float sc = 1.0f, isc = 1.0f;

#ifdef T16_STAMPS
    uint32_t stamps[23];
#pragma unroll
    for (int i = 0; i < 23; ++i) stamps[i] = 0;
    uint32_t tlast = (uint32_t)__builtin_amdgcn_s_memtime();
#endif
#ifdef T16_WG_STAGGER   // experiment: workgroups start T16_WG_STAGGER x 64 cycles apart in 8 phases (their record bursts then do not coincide chip-wide)
    for (int i = 0; i < (int)((blockIdx.x >> 3) & 7u); ++i) __builtin_amdgcn_s_sleep(T16_WG_STAGGER);
#endif
#ifdef T16_SKEW_START   // timing experiments only (with -DT16_NO_BARRIER): the second wave of every SIMD starts T16_SKEW_START x 8 128 cycles late
    if (wv >= 4)
        for (int i = 0; i < T16_SKEW_START; ++i) __builtin_amdgcn_s_sleep(127);
#endif
    const int ngroups_all = (a.f.B + kWaves - 1) / kWaves;
    for (int grp = blockIdx.x; grp < ngroups_all; grp += gridDim.x) {
        const int ray = kWaves * grp + wv;
        const bool active = ray < a.f.B && !(a.f.ray_valid && !a.f.ray_valid[ray]);  // wave-uniform
        const int rayc = active ? ray : 0;  // inactive waves run on ray 0's geometry with zero inputs and gradients
        if (ray < a.f.B && !active && a.dfeat)  // invalid ray: zero feature gradients
            for (int s = lane >> 2; s < S; s += 16)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
                    reinterpret_cast<float2 *>(a.dfeat)[(size_t)(4 * jj + (lane & 3)) * a.f.B * S + (size_t)ray * S + s] = make_float2(0, 0);
        float o[3], d[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            o[k] = a.f.rays_o[3 * rayc + k];
            d[k] = a.f.rays_d[3 * rayc + k];
        }
        const float dnorm = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        const float *go = a.grad_out + (size_t)rayc * SCANERF_RAY_OUT;
        const float *fo = a.f.out_ray + (size_t)rayc * SCANERF_RAY_OUT;
        {   // per ray: SH (published for the owners of the SH part) and the Dir layer-0 accumulator start, parked in LDS
            const int ln = fresh(lane), c = ln & 15, q = ln >> 4;
            const T16Lane L = t16_lane(ln, LD::kStageWave);
            float sh[16];
            ray_sh(d, dnorm, sh);
            v4f s0, s1;  // B operand of the SH k-step: slot (q, j) = SH[8q + j] for q < 2, zero above
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s0[j] = q == 0 ? sh[j] : (q == 1 ? sh[8 + j] : 0.0f);
                s1[j] = q == 0 ? sh[4 + j] : (q == 1 ? sh[12 + j] : 0.0f);
            }
            const T16HL shB = t16_split(s0, s1);
            v4f dinit[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) dinit[b] = t16_bias(lds, 2, b, q, LD::kBias);
            // k-step 1 of the D0 pairs
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const char *p = lds + T16_D0 + (b * 2 + 1) * T16_PAIR + (SPLIT ? L.pos8 : L.lo16);
                const t16_h8 ahi = SPLIT ? s16_lda(p) : *reinterpret_cast<const t16_h8 *>(p);
                const t16_h8 alo = SPLIT ? s16_lda(p + T16_SUB) : *reinterpret_cast<const t16_h8 *>(p + T16_SUB);
                T16_REGION_BEGIN();
                dinit[b] = t16_mfma(alo, shB.hi, dinit[b]);
                dinit[b] = t16_mfma(ahi, shB.lo, dinit[b]);
                dinit[b] = t16_mfma(ahi, shB.hi, dinit[b]);
                T16_REGION_END();
            }
            if (c == 0) {
                float4 *dp = reinterpret_cast<float4 *>(lds + LD::kDinit + wv * 256);
#pragma unroll
                for (int b = 0; b < 4; ++b) dp[4 * b + q] = make_float4(dinit[b][0], dinit[b][1], dinit[b][2], dinit[b][3]);
            }
            if (lane < 16) {
                float v = 0.0f;
#pragma unroll
                for (int j = 0; j < 16; ++j) v = lane == j ? sh[j] : v;
                shbuf[wv * 16 + lane] = v;
            }
        }
        float Rcarry = 0.0f;              // sum of a_j w_j over all later tiles of the ray
        // POSE: sum over the ray's samples of dL/d(delta) * dist (this lane's samples; -> dL/d|d|) and of the Directional layer-0
        // pre-activation gradient (under the scale 2^K; -> dL/dSH), reduce-scattered over the 16 lanes of a group every tile so that
        // ONE register carries it (16 accumulators per lane cost 80 spilled registers)
        float pose_dn = 0.0f, pose_rs = 0.0f;   // pose_rs: lane (c, q) owns unit 16 (c >> 2) + 4 q + (c & 3)
        // ... and, with the forward's position Jacobians (f.jstash), the gradient through the sample positions: this lane's
        // levels of the tile being emitted, chained to the ray through the contraction
        float pose_go[3] = { 0, 0, 0 }, pose_gd[3] = { 0, 0, 0 };

        // Feature gradients of one tile: dfeat rows and / or the scatter records (scatter.hip) into the ranges the plan
        // reserved, ONE LEVEL PER CALL.  e0 / e1 = dX blocks 0 / 1: register g of block e = x-stash position 8q + 4e + g =
        // feature g & 1 of this lane's level jj = 2e + (g >> 1); level(q, jj) = 4(j8 >> 1) + 2(q >> 1) + (j8 & 1), j8 = 4(q & 1) + jj.
        //
        // THE FOUR LEVELS OF A TILE ARE EMITTED AT FOUR POINTS OF THE NEXT TILE.  The 8.6 GB of records are the kernel's
        // memory traffic; stored in one piece at the end of a tile they arrive at the L2s in bursts (the 8 waves of every
        // workgroup run in lock step), every new line evicts a dirty one, and the waves wait for the chip's write bandwidth
        // while nothing computes: removing the barriers or the weight-gradient products did not change the kernel's time.
        // Spread over the next tile's layers the same stores overlap its matrix work.
        auto emit_level = [&](int tile_e, int jj, const v4f &e0, const v4f &e1, const float pe[3]) {
#ifdef T16_NO_EMIT   // timing experiments only (results are wrong): no records, no dfeat
            return;
#endif
            const int ln = fresh(lane), c = ln & 15, q = ln >> 4;
            const int s = tile_e * 16 + c;
            if (tile_e < 0 || !(s < S) || !active) return;
            const int j8 = 4 * (q & 1) + jj, level = 4 * (j8 >> 1) + 2 * (q >> 1) + (j8 & 1);
#ifdef T16_MERGE_EMU   // timing experiment only (results are wrong): the record COUNT a run-merged emission of the coarse levels would leave
            {          // (a run of k samples in one cell = 8 single-entry records instead of 4k: levels 0-3 have runs of ~8 / 6 / 4.5 / 3.4)
                const int kk = level == 0 ? 4 : (level == 1 ? 3 : (level <= 3 ? 2 : 1));
                if (kk > 1 && (c % kk) != 0) return;
            }
#endif
            const float gx = jj == 0 ? e0[0] : (jj == 1 ? e0[2] : (jj == 2 ? e1[0] : e1[2]));
            const float gy = jj == 0 ? e0[1] : (jj == 1 ? e0[3] : (jj == 2 ? e1[1] : e1[3]));
            if (a.dfeat) reinterpret_cast<float2 *>(a.dfeat)[(size_t)level * a.f.B * S + (size_t)ray * S + s] = make_float2(gx, gy);
            if (a.recs && !((plan_skip >> level) & 1u)) {   // (a masked level's gradient is exactly zero: no records, as planned)
                const uint32_t mask = (uint32_t)a.f.T - 1u;
                const int4 r = *reinterpret_cast<const int4 *>(lres + 4 * level);
                const int32_t rr[3] = { r.x, r.y, r.z };
                Pairs pr;
#ifdef SCANERF_BWD_EXPERIMENTS
                if ((a.f.dbg & 15) == 6) {  // no index arithmetic: made-up pairs
#pragma unroll
                    for (int q2 = 0; q2 < 4; ++q2) {
                        pr.idx0[q2] = (uint32_t)(ln * 977 + q2 * 131 + level * 7919 + s * 31) & mask;
                        pr.wyz[q2] = 0.25f;
                    }
                    pr.xm = 1u;
                    pr.tx = 0.5f;
                } else
#endif
                make_pairs(pe, rr, mask, pr);
                gmax = fmaxf(gmax, fmaxf(fabsf(gx), fabsf(gy)));
                uint32_t *cl = cursor + level * a.bins.NB;
                float *gl = a.grad_features + (size_t)level * a.f.T * 2;
#ifdef SCANERF_BWD_EXPERIMENTS
                if (REC == 1 && (a.f.dbg & 15) == 5) emit_pairs8<5>(pr, gx, gy, cl, a.bins.bucket_log, rec_capacity(a.bins.capacity, 1), a.recs, gl);
                else
#endif
                if (REC == 1) emit_pairs8(pr, gx, gy, cl, a.bins.bucket_log, rec_capacity(a.bins.capacity, 1), a.recs, gl);
                else if (REC == 2) emit_pairs12(pr, gx, gy, cl, a.bins.bucket_log, rec_capacity(a.bins.capacity, 2), a.recs, gl);
                else if ((a.f.dbg & 15) == 0) emit_pairs(pr, gx, gy, cl, a.bins.bucket_log, a.bins.capacity, a.recs, gl);
#ifdef SCANERF_BWD_EXPERIMENTS
                else if ((a.f.dbg & 15) == 1) emit_pairs<1>(pr, gx, gy, cl, a.bins.bucket_log, a.bins.capacity, a.recs, gl);
                else if ((a.f.dbg & 15) == 2) emit_pairs<2>(pr, gx, gy, cl, a.bins.bucket_log, a.bins.capacity, a.recs, gl);
                else if ((a.f.dbg & 15) == 3) emit_pairs<3>(pr, gx, gy, cl, a.bins.bucket_log, a.bins.capacity, a.recs, gl);
                else if ((a.f.dbg & 15) == 4) emit_pairs<4>(pr, gx, gy, cl, a.bins.bucket_log, a.bins.capacity, a.recs, gl);
                else emit_pairs<8>(pr, gx, gy, cl, a.bins.bucket_log, a.bins.capacity, a.recs, gl);
#endif
            }
            SCANERF_STORE_GUARD();  // the records' / dfeat's data registers are about to be reused by matrix results
        };
        v4f pdx0 = { 0, 0, 0, 0 }, pdx1 = { 0, 0, 0, 0 };   // the previous tile's dX, emitted during this one
        float ppe[3] = { 0, 0, 0 };                         // ... and its samples' contracted positions (computed once per tile)
        int ptile = -1;
        // SKEWED EMISSION (SPLIT, a.park).  All 8 waves storing their 16 x 64 records at the same moment is 128 fully divergent
        // stores = ~8 000 cycles of the CU's address path (one 64-B line per clock) while nothing else runs: the interval was
        // 27 % of the kernel, the younger wave of every SIMD finished it 4 400 cycles after the older one and the older ones
        // waited that long in barrier S (tools/bwd_stamps.py).  Waves 4-7 (the second wave of every SIMD) therefore PARK their
        // tile's dX in LDS (8 KB; the registers to carry it through the forward recompute do not exist) and emit
        // its records behind the next tile's compositing, right before barrier S; waves 0-3 emit at the tile's end as before.
        // Each SIMD then always has one wave in vector work while the other's stores drain, and the bursts halve.
        const bool park_wave = SPLIT && a.park && wv >= 4;   // wave-uniform
        float pz = 0.0f;                                     // sample depth of the parked tile (its position is formed again)
        auto emit_parked = [&]() {
            if constexpr (SPLIT) {
                if (ptile < 0) return;
                const int lp = fresh(lane);
                const float4 *pk = reinterpret_cast<const float4 *>(lds + LD::kPark + (wv - 4) * 2048);
                const float4 p0 = pk[lp], p1 = pk[64 + lp];
                const v4f e0 = { p0.x, p0.y, p0.z, p0.w }, e1 = { p1.x, p1.y, p1.z, p1.w };
                float pe[3] = { 0, 0, 0 };
                if (a.recs) contract_point(a.f, o, d, pz, pe);
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) emit_level(ptile, jj, e0, e1, pe);
                ptile = -1;
            }
        };

        // A tile's inputs are loaded ONE TILE AHEAD, before the previous tile's records are stored: vector-memory operations
        // complete in issue order, so a load issued behind the 16 record stores of a tile would not return before they
        // have drained -- at the write rate of the whole chip, the stores' latency would be exposed at the top of every
        // tile (measured: 2.0 of 5.5 ms).  Issued first, the loads (and the HBM misses of the x-stash) have a whole tile to land.
        struct TileIn {
            float z, dist, tT;
            v4f xa, xb;
        };
        auto load_tile = [&](int t) {
            TileIn in;
            const int ln = fresh(lane), c = ln & 15, q = ln >> 4;
            const int s = t * 16 + c;
            const bool live = s < S;
            in.z = live ? a.f.z_vals[(size_t)rayc * S + s] : 0.0f;
            in.dist = live ? a.f.dists[(size_t)rayc * S + s] : 0.0f;
            in.tT = a.tile_T[(size_t)rayc * nt16 + t];
            in.xa = v4f{ 0, 0, 0, 0 };
            in.xb = v4f{ 0, 0, 0, 0 };
            if (active) {
                const float4 *xs = reinterpret_cast<const float4 *>(a.xstash + ((size_t)ray * S + (live ? s : 0)) * 32 + 8 * q);
#ifdef T16_NT_LOADS   // experiment: the x-stash (1 GB, read once) as non-temporal loads, so that it does not displace the records' partially written lines in L2
                const v4f n0 = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(xs)), n1 = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(xs) + 1);
                const float4 q0 = make_float4(n0[0], n0[1], n0[2], n0[3]), q1 = make_float4(n1[0], n1[1], n1[2], n1[3]);
#else
                const float4 q0 = xs[0], q1 = xs[1];
#endif
                in.xa = v4f{ q0.x, q0.y, q0.z, q0.w };
                in.xb = v4f{ q1.x, q1.y, q1.z, q1.w };
            }
            return in;
        };
        TileIn nxt = load_tile(nt16 - 1);
        for (int tile = nt16 - 1; tile >= 0; --tile) {
            STAMP(22);
#ifdef T16_DUMMY_VALU_TOP   // timing experiment: is the kernel bound by vector issue?  N dependent-free vector instructions per tile
            {
                float dmy = 1.0f;
#pragma unroll
                for (int i = 0; i < T16_DUMMY_VALU_TOP; ++i) asm volatile("v_mul_f32 %0, 1.0, %0" : "+v"(dmy));
                asm volatile("" ::"v"(dmy));
            }
#endif
            const int ln = fresh(lane);               // this tile's lane terms (not loop invariants: see fresh())
            const int c = ln & 15, q = ln >> 4;
            T16Lane L = t16_lane(ln, LD::kStageWave);
            const int s = tile * 16 + c;
            const bool live = s < S;
            const float z = nxt.z, dist_i = nxt.dist, tile_T_in = nxt.tT;
            float delta = dist_i * dnorm;
            if (a.f.infinity && s == S - 1) delta = 1e10f;

            // ================= forward recompute =================
            const v4f xa = nxt.xa, xb = nxt.xb;
            t16_h8 xh, a0h[2], c0h[2], c1h[2], Hh[2];   // (!SPLIT) hi parts kept for the weight gradients
            t16_h4 dg0[4], dgv0[4], dgv1[4];           // (!SPLIT) G'(u0), G'(v0), G'(v1) as f16
            v4f ku0[4], khh[4], kv0[4], kv1[4];        // (SPLIT) pre-activations of the three Gaussian layers and H, f32: the
                                                       // activations, their splits and G' are formed again where they are used
            float sigma, dsig_dpre, dif[3], tint[3], spec[3];
            if constexpr (SPLIT) {
                T16HL HB[2];
                {
                    const T16HL xB = t16_split(xa, xb);
                    v4f act[4];   // (u0 is formed again in the backward: not kept)
#pragma unroll
                    for (int b = 0; b < 4; ++b) act[b] = t16_bias(lds, 0, b, q, LD::kBias);
                    s16_layer<4, 1>(act, lds, T16_L0, L.pos8, &xB);
#pragma unroll
                    for (int b = 0; b < 4; ++b)
#pragma unroll
                        for (int g = 0; g < 4; ++g) act[b][g] = gauss_fast(act[b][g]);
                    const T16HL aB[2] = { t16_split(act[0], act[1]), t16_split(act[2], act[3]) };
#pragma unroll
                    for (int b = 0; b < 4; ++b) khh[b] = t16_bias(lds, 1, b, q, LD::kBias);
                    s16_layer<4, 2>(khh, lds, T16_L1, L.pos8, aB);
                    HB[0] = t16_split(khh[0], khh[1]);
                    HB[1] = t16_split(khh[2], khh[3]);
                }
                {   // heads on H[:32]
                    v4f hd[2] = { t16_ld4(lds, LD::kBias + 256 * 4), t16_ld4(lds, LD::kBias + 260 * 4) };
                    s16_layer<2, 1>(hd, lds, T16_HEAD, L.pos8, &HB[0]);
                    sigma = T16_SOFTPLUS(hd[0][0]);
                    dsig_dpre = T16_DSOFTPLUS(hd[0][0]);
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        dif[k] = sigmoid_fast(hd[0][1 + k]);
                        tint[k] = sigmoid_fast(hd[1][k]);
                    }
                }
                T16HL cB[2];
                {   // Directional_MLP.mlp.0: H[32:64] part; SH part + bias in dinit
#pragma unroll
                    for (int b = 0; b < 4; ++b) kv0[b] = t16_ld4(lds, LD::kDinit + wv * 256 + (16 * b + 4 * q) * 4);
#pragma unroll
                    for (int b = 0; b < 4; ++b) {  // k-step 0 of the D0 pairs
                        const char *p = lds + T16_D0 + (b * 2) * T16_PAIR + L.pos8;
                        const t16_h8 ahi = s16_lda(p), alo = s16_lda(p + T16_SUB);
                        T16_REGION_BEGIN();
                        kv0[b] = t16_mfma(alo, HB[1].hi, kv0[b]);
                        kv0[b] = t16_mfma(ahi, HB[1].lo, kv0[b]);
                        kv0[b] = t16_mfma(ahi, HB[1].hi, kv0[b]);
                        T16_REGION_END();
                    }
                    v4f act[4];
#pragma unroll
                    for (int b = 0; b < 4; ++b)
#pragma unroll
                        for (int g = 0; g < 4; ++g) act[b][g] = gauss_fast(kv0[b][g]);
                    cB[0] = t16_split(act[0], act[1]);
                    cB[1] = t16_split(act[2], act[3]);
                }
                {
#pragma unroll
                    for (int b = 0; b < 4; ++b) kv1[b] = t16_bias(lds, 3, b, q, LD::kBias);
                    s16_layer<4, 2>(kv1, lds, T16_D1, L.pos8, cB);
                    v4f act[4];
#pragma unroll
                    for (int b = 0; b < 4; ++b)
#pragma unroll
                        for (int g = 0; g < 4; ++g) act[b][g] = gauss_fast(kv1[b][g]);
                    cB[0] = t16_split(act[0], act[1]);
                    cB[1] = t16_split(act[2], act[3]);
                }
                {
                    v4f r[1] = { t16_ld4(lds, LD::kBias + 264 * 4) };
                    s16_layer<1, 2>(r, lds, T16_D2, L.pos8, cB);
#pragma unroll
                    for (int k = 0; k < 3; ++k) spec[k] = sigmoid_fast(r[0][k]);
                }
            } else {
                T16HL HB[2];
                {
                    const T16HL xB = t16_split(xa, xb);
                    xh = xB.hi;
                    v4f u[4];
#pragma unroll
                    for (int b = 0; b < 4; ++b) u[b] = t16_bias(lds, 0, b, q);
                    t16_layer<4, 1>(u, lds, T16_L0, L.lo16, &xB);
#pragma unroll
                    for (int b = 0; b < 4; ++b) act_deriv(u[b], dg0[b]);
                    const T16HL aB[2] = { t16_split(u[0], u[1]), t16_split(u[2], u[3]) };
                    a0h[0] = aB[0].hi;
                    a0h[1] = aB[1].hi;
                    v4f hh[4];
#pragma unroll
                    for (int b = 0; b < 4; ++b) hh[b] = t16_bias(lds, 1, b, q);
                    t16_layer<4, 2>(hh, lds, T16_L1, L.lo16, aB);
                    HB[0] = t16_split(hh[0], hh[1]);
                    HB[1] = t16_split(hh[2], hh[3]);
                    Hh[0] = HB[0].hi;
                    Hh[1] = HB[1].hi;
                }
                emit_level(ptile, 0, pdx0, pdx1, ppe);
                {   // heads on H[:32]
                    v4f hd[2] = { t16_ld4(lds, T16_BIAS + 256 * 4), t16_ld4(lds, T16_BIAS + 260 * 4) };
                    t16_layer<2, 1>(hd, lds, T16_HEAD, L.lo16, &HB[0]);
                    sigma = T16_SOFTPLUS(hd[0][0]);
                    dsig_dpre = T16_DSOFTPLUS(hd[0][0]);
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        dif[k] = sigmoid_fast(hd[0][1 + k]);
                        tint[k] = sigmoid_fast(hd[1][k]);
                    }
                }
                T16HL cB[2];
                {   // Directional_MLP.mlp.0: H[32:64] part; SH part + bias in dinit
                    v4f v[4];
#pragma unroll
                    for (int b = 0; b < 4; ++b) v[b] = t16_ld4(lds, LD::kDinit + wv * 256 + (16 * b + 4 * q) * 4);
#pragma unroll
                    for (int b = 0; b < 4; ++b) {  // k-step 0 of the D0 pairs
                        const char *p = lds + T16_D0 + (b * 2) * T16_PAIR + L.lo16;
                        const t16_h8 ahi = *reinterpret_cast<const t16_h8 *>(p), alo = *reinterpret_cast<const t16_h8 *>(p + T16_SUB);
                        T16_REGION_BEGIN();
                        v[b] = t16_mfma(alo, HB[1].hi, v[b]);
                        v[b] = t16_mfma(ahi, HB[1].lo, v[b]);
                        v[b] = t16_mfma(ahi, HB[1].hi, v[b]);
                        T16_REGION_END();
                    }
#pragma unroll
                    for (int b = 0; b < 4; ++b) act_deriv(v[b], dgv0[b]);
                    cB[0] = t16_split(v[0], v[1]);
                    cB[1] = t16_split(v[2], v[3]);
                    c0h[0] = cB[0].hi;
                    c0h[1] = cB[1].hi;
                }
                emit_level(ptile, 1, pdx0, pdx1, ppe);
                {
                    v4f v[4];
#pragma unroll
                    for (int b = 0; b < 4; ++b) v[b] = t16_bias(lds, 3, b, q);
                    t16_layer<4, 2>(v, lds, T16_D1, L.lo16, cB);
#pragma unroll
                    for (int b = 0; b < 4; ++b) act_deriv(v[b], dgv1[b]);
                    cB[0] = t16_split(v[0], v[1]);
                    cB[1] = t16_split(v[2], v[3]);
                    c1h[0] = cB[0].hi;
                    c1h[1] = cB[1].hi;
                }
                emit_level(ptile, 2, pdx0, pdx1, ppe);
                {
                    v4f r[1] = { t16_ld4(lds, T16_BIAS + 264 * 4) };
                    t16_layer<1, 2>(r, lds, T16_D2, L.lo16, cB);
#pragma unroll
                    for (int k = 0; k < 3; ++k) spec[k] = sigmoid_fast(r[0][k]);
                }
            }

            if constexpr (!SPLIT) emit_level(ptile, 3, pdx0, pdx1, ppe);
            STAMP(0);

            // ================= compositing: recompute and adjoint (16-lane scans, identical in the 4 lane groups) =================
            // (the density head and the opacity on v_exp / v_log / v_rcp, not the library's log1pf / expf and an IEEE division: ~65 vector
            // instructions per lane and tile on the tile's dependent chain; plan + backward 5.25 -> 5.17 ms same-box, tools/ab_heads.sh)
            const float ex = live ? T16_EXPNEG(sigma * delta) : 1.0f;  // 1 - alpha
            const float alpha = 1.0f - ex;
            const float fi = 1.0f - alpha + 1e-6f;
            float incl = fi;   // inclusive prefix product over the row (DPP row shifts, identity shifted in)
            incl *= row_shr<1>(incl, 1.0f);
            incl *= row_shr<2>(incl, 1.0f);
            incl *= row_shr<4>(incl, 1.0f);
            incl *= row_shr<8>(incl, 1.0f);
            const float excl = row_shr<1>(incl, 1.0f);
            const float Ti = tile_T_in * excl;
            const float w = alpha * Ti;
            float gD[3], gS[3], gTi[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float pre = fo[5 + k] + fo[8 + k];  // diffuse + specular before the clamp
                const float grgb = (pre >= 0.0f && pre <= 1.0f) ? go[k] : 0.0f;
                gD[k] = go[5 + k] + grgb;
                gS[k] = go[8 + k] + grgb;
                gTi[k] = go[11 + k];
            }
            const float gDepth = go[3], gTl = go[4], gW2 = go[14], Tl = fo[4];
            float ai = gDepth * z;
#pragma unroll
            for (int k = 0; k < 3; ++k) ai += gD[k] * dif[k] + gS[k] * tint[k] * spec[k] + gTi[k] * tint[k];
            const float aw = live ? ai * w : 0.0f;
            float rs = aw;  // inclusive suffix sum inside the tile
            rs += row_shl<1>(rs, 0.0f);
            rs += row_shl<2>(rs, 0.0f);
            rs += row_shl<4>(rs, 0.0f);
            rs += row_shl<8>(rs, 0.0f);
            const float suffix = Rcarry + rs - aw;
            // the tile's total = lane 0's suffix sum (the four rows hold the same samples), through a scalar register
            Rcarry += __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, rs)));
            float dalpha = Ti * ai - T16_DIV(suffix + ((s < S - 1) ? gTl * Tl : 0.0f), fi);
            if (!live) dalpha = 0.0f;
            const float dsigma = dalpha * delta * ex;
            if (POSE && active && q == 0)  // (the infinity sample's delta is a constant)
                pose_dn += dalpha * sigma * ex * ((a.f.infinity && s == S - 1) ? 0.0f : dist_i);
            float gh[7], gs3[3];  // gradients w.r.t. the head / rgb pre-activations
            gh[0] = dsigma * dsig_dpre;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                gh[1 + k] = w * gD[k] * dif[k] * (1.0f - dif[k]);
                gh[4 + k] = w * (gS[k] * spec[k] + gTi[k]) * tint[k] * (1.0f - tint[k]);
                gs3[k] = (w * gS[k] * tint[k] + gW2 * w * 2.0f * spec[k]) * spec[k] * (1.0f - spec[k]);
            }
            if (!active) {  // select, not multiply: the inputs of an inactive wave may be anything
#pragma unroll
                for (int k = 0; k < 7; ++k) gh[k] = 0.0f;
#pragma unroll
                for (int k = 0; k < 3; ++k) gs3[k] = 0.0f;
            }
            {   // this tile's largest |gradient|, published for the workgroup's scale
                float mx = 0.0f;
#pragma unroll
                for (int k = 0; k < 7; ++k) mx = fmaxf(mx, fabsf(gh[k]));
#pragma unroll
                for (int k = 0; k < 3; ++k) mx = fmaxf(mx, fabsf(gs3[k]));
                mx = fmaxf(mx, row_ror<8>(mx));
                mx = fmaxf(mx, row_ror<4>(mx));
                mx = fmaxf(mx, row_ror<2>(mx));
                mx = fmaxf(mx, row_ror<1>(mx));
                if (lane == 0) mxbuf[wv] = mx;
            }
            if (park_wave) emit_parked();   // the previous tile's records (see "skewed emission")
            STAMP(1);
            STEP_BARRIER();  // ---- S: tile maxima visible; every wave is done with the previous tile's staged operands
            STAMP(2);
            {   // gradient scale: keep the workgroup's largest |gradient| * 2^K in [2^2, 2^6): with single f16 operands the
                // 8 tiles' smaller gradients need the room BELOW the maximum (full precision down to 2^-19 of it, subnormal to
                // 2^-29), the chains' growth through G' <= 6 and the weights the 2^10 above it
                const float4 m0 = reinterpret_cast<const float4 *>(mxbuf)[0], m1 = reinterpret_cast<const float4 *>(mxbuf)[1];
                const float mx = fmaxf(fmaxf(fmaxf(m0.x, m0.y), fmaxf(m0.z, m0.w)), fmaxf(fmaxf(m1.x, m1.y), fmaxf(m1.z, m1.w)));
                const float ms = mx * sc;
                if (mx > 0.0f && mx < 3.0e38f && (ms >= 64.0f || ms < 4.0f)) {
                    int e;
                    frexpf(mx, &e);  // mx = f * 2^e, f in [0.5, 1)
                    int Kn = 6 - e;  // mx * 2^K in [32, 64)
                    Kn = Kn > K + 100 ? K + 100 : (Kn < K - 100 ? K - 100 : Kn);  // rescale factor stays an f32 power of two
                    Kn = Kn > 100 ? 100 : (Kn < -100 ? -100 : Kn);
                    const float r = ldexpf(1.0f, Kn - K);
                    K = Kn;
                    sc = ldexpf(1.0f, K);
                    isc = ldexpf(1.0f, -K);
#pragma unroll
                    for (int i = 0; i < 2; ++i) { gW_D1[i] *= r; gW_L1[i] *= r; }
                    gW_D0[0] *= r; gW_D0[1] *= r; gW_L0[0] *= r; gW_nar[0] *= r;
                    gB_D1 *= r; gB_L1 *= r; gB_D0 *= r; gB_L0 *= r; gB_nar *= r;
                    if (POSE) pose_rs *= r;
                }
#pragma unroll
                for (int k = 0; k < 7; ++k) gh[k] *= sc;
#pragma unroll
                for (int k = 0; k < 3; ++k) gs3[k] *= sc;
            }
            v4f dx[2];
            zero4(dx);
            if constexpr (SPLIT) {
                constexpr int kLo = 2 * T16_STAGE_MAT;
                const v4f zero = { 0, 0, 0, 0 };
                // (tried: the two waves of a SIMD running every interval's weight-gradient products and chain in opposite orders,
                // so that one's vector work meets the other's matrix / LDS work: 5.13 -> 5.18 ms, not kept)
                // ================= narrow layers: heads (32 -> 7) and rgb (64 -> 3) =================
                L = fresh_lane(L);
                // one 16-row block: rows 0-3 sigma, dif; 4-6 tint; 8-10 rgb  (lane group q holds rows 4q .. 4q+3)
                T16HL narS;   // B operand of the transposed products (second block of the k-step = zeros)
                {
                    v4f nar;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float r1 = g < 3 ? gh[4 + g] : 0.0f, r2 = g < 3 ? gs3[g] : 0.0f;
                        nar[g] = q == 0 ? gh[g] : (q == 1 ? r1 : (q == 2 ? r2 : 0.0f));
                    }
                    narS = t16_split(nar, zero);
                    t16_stage_put(stY, L, 0, lo4(narS.hi));
                    t16_stage_put(stY + kLo, L, 0, lo4(narS.lo));
                    stage_put2(stY, L, 2, t16_split(khh[0], khh[1]));   // X operand of the heads' weight gradient: H[:32] in blocks 2, 3 of Y
                    stage_act(stX, L, kv1);     // X operand of the rgb layer's weight gradient: c1 = G(v1)
                }
                STAMP(3);
                STEP_BARRIER();  // ---- A1
                STAMP(4);
                if (wv == 0) wgrad<1, true, 1, true>(gW_nar, gB_nar, stage, L, 0, 0, 2);          // heads: x = H[0:16]
                else if (wv == 1) { float dummy = 0.0f; wgrad<1, false, 1, true>(gW_nar, dummy, stage, L, 0, 0, 3); }   // heads: x = H[16:32]
                else if (wv < 6) { float dummy = 0.0f; wgrad<1, false, 1, true>(gW_nar, dummy, stage, L, 0, T16_STAGE_MAT, wv - 2); }  // rgb: x = c1 block wv-2
                // dv1 = (W_rgb^T gs3) * G'(v1)
                T16HL dyS[2];
                {
                    v4f dc[4];
                    zero4(dc);
                    s16_chain_narrow<4>(dc, lds, S16T_D2, L.lo16, narS);
#pragma unroll
                    for (int b = 0; b < 4; ++b) dc[b] *= gauss_deriv(kv1[b]);
                    dyS[0] = t16_split(dc[0], dc[1]);
                    dyS[1] = t16_split(dc[2], dc[3]);
                }
                STAMP(5);
                STEP_BARRIER();  // ---- B1
                STAMP(6);
                // ================= Directional_MLP.mlp.2 (64 -> 64) =================
                L = fresh_lane(L);
                stage_put2(stY, L, 0, dyS[0]);
                stage_put2(stY, L, 2, dyS[1]);
                stage_act(stX, L, kv0);
                STAMP(7);
                STEP_BARRIER();  // ---- A2
                STAMP(8);
                if (T16_LOAD_AT == 2 && tile > 0) nxt = load_tile(tile - 1);   // (early: see load_tile)
                if (cb == 0) wgrad<2, true, 1, true>(gW_D1, gB_D1, stage, L, rb, T16_STAGE_MAT, 0);
                else { float dummy = 0.0f; wgrad<2, false, 1, true>(gW_D1, dummy, stage, L, rb, T16_STAGE_MAT, 2); }
                {
                    v4f dc[4];
                    zero4(dc);
                    s16_chain<4, 2, 2>(dc, lds, T16_D1, L.trp, dyS);
#pragma unroll
                    for (int b = 0; b < 4; ++b) dc[b] *= gauss_deriv(kv0[b]);   // dv0
                if (POSE) {  // 16 values x 16 samples -> lane c keeps the tile's sum of value c (= block c >> 2, register c & 3)
                    float v[16];
#pragma unroll
                    for (int i = 0; i < 16; ++i) v[i] = dc[i >> 2][i & 3];
#pragma unroll
                    for (int off = 8, half = 8; off > 0; off >>= 1, half >>= 1) {
                        const bool up = (c & off) != 0;
#pragma unroll
                        for (int j = 0; j < half; ++j) {
                            const float mine = up ? v[j + half] : v[j], send = up ? v[j] : v[j + half];
                            v[j] = mine + __shfl_xor(send, off, 16);
                        }
                    }
                    pose_rs += v[0];
                }
                    dyS[0] = t16_split(dc[0], dc[1]);
                    dyS[1] = t16_split(dc[2], dc[3]);
                }
                STAMP(9);
                STEP_BARRIER();  // ---- B2
                STAMP(10);
#ifdef T16_DUMMY_VALU_MID
                {
                    float dmy = 1.0f;
#pragma unroll
                    for (int i = 0; i < T16_DUMMY_VALU_MID; ++i) asm volatile("v_mul_f32 %0, 1.0, %0" : "+v"(dmy));
                    asm volatile("" ::"v"(dmy));
                }
#endif
                // ================= Directional_MLP.mlp.0 (32 of its 48 inputs; the SH part per ray) =================
                L = fresh_lane(L);
                stage_put2(stY, L, 0, dyS[0]);
                stage_put2(stY, L, 2, dyS[1]);
                stage_put2(stX, L, 0, t16_split(khh[2], khh[3]));
                {   // the SH part of the layer's input is constant along the ray: staged as 16 more "units" (block 2 of X)
                    const float4 shq = *reinterpret_cast<const float4 *>(shbuf + wv * 16 + 4 * q);
                    const T16HL shS = t16_split(v4f{ shq.x, shq.y, shq.z, shq.w }, zero);
                    t16_stage_put(stX, L, 2, lo4(shS.hi));
                    t16_stage_put(stX + kLo, L, 2, lo4(shS.lo));
                }
                STAMP(11);
                STEP_BARRIER();  // ---- A3
                STAMP(12);
                if (T16_LOAD_AT == 3 && tile > 0) nxt = load_tile(tile - 1);   // (early: see load_tile)
                if (cb == 0) wgrad<2, true, 2, true>(gW_D0, gB_D0, stage, L, rb, T16_STAGE_MAT, 0);   // x = H[32:48] and SH
                else { float dummy = 0.0f; wgrad<1, false, 1, true>(gW_D0, dummy, stage, L, rb, T16_STAGE_MAT, 1); }   // x = H[48:64]
                {
                    v4f dH[4];
                    zero4(dH);
                    s16_chain<2, 2, 2>(&dH[2], lds, T16_D0, L.trp, dyS);        // dH[32:64] = W_D0[:, :32]^T dv0 (input k-step 0 of the D0 pairs)
                    s16_chain_narrow<2>(&dH[0], lds, S16T_HEAD, L.lo16, narS);  // dH[0:32] = heads^T gh
                    dyS[0] = t16_split(dH[0], dH[1]);
                    dyS[1] = t16_split(dH[2], dH[3]);
                }
                STAMP(13);
                STEP_BARRIER();  // ---- B3
                STAMP(14);
                // ================= Spatial_MLP.mlp.2 (64 -> 64, linear) =================
                L = fresh_lane(L);
                stage_put2(stY, L, 0, dyS[0]);
                stage_put2(stY, L, 2, dyS[1]);
                {   // u0 = W0 x + b0 again (12 MFMAs against 16 registers held through the whole tile)
                    const T16HL xB = t16_split(xa, xb);
#pragma unroll
                    for (int b = 0; b < 4; ++b) ku0[b] = t16_bias(lds, 0, b, q, LD::kBias);
                    s16_layer<4, 1>(ku0, lds, T16_L0, L.pos8, &xB);
                }
                stage_act(stX, L, ku0);
                STAMP(15);
                STEP_BARRIER();  // ---- A4
                STAMP(16);
                if (T16_LOAD_AT == 4 && tile > 0) nxt = load_tile(tile - 1);   // (early: see load_tile)
                if (cb == 0) wgrad<2, true, 1, true>(gW_L1, gB_L1, stage, L, rb, T16_STAGE_MAT, 0);
                else { float dummy = 0.0f; wgrad<2, false, 1, true>(gW_L1, dummy, stage, L, rb, T16_STAGE_MAT, 2); }
                {
                    v4f dc[4];
                    zero4(dc);
                    s16_chain<4, 2, 2>(dc, lds, T16_L1, L.trp, dyS);
#pragma unroll
                    for (int b = 0; b < 4; ++b) dc[b] *= gauss_deriv(ku0[b]);   // du0
                    dyS[0] = t16_split(dc[0], dc[1]);
                    dyS[1] = t16_split(dc[2], dc[3]);
                }
                STAMP(17);
                STEP_BARRIER();  // ---- B4
                STAMP(18);
                // ================= Spatial_MLP.mlp.0 (32 -> 64) =================
                L = fresh_lane(L);
                stage_put2(stY, L, 0, dyS[0]);
                stage_put2(stY, L, 2, dyS[1]);
                stage_put2(stX, L, 0, t16_split(xa, xb));
                STAMP(19);
                STEP_BARRIER();  // ---- A5
                STAMP(20);
                if (T16_LOAD_AT == 5 && tile > 0) nxt = load_tile(tile - 1);   // (early: see load_tile)
                if (cb == 0) wgrad<1, true, 1, true>(gW_L0, gB_L0, stage, L, rb, T16_STAGE_MAT, 0);
                else { float dummy = 0.0f; wgrad<1, false, 1, true>(gW_L0, dummy, stage, L, rb, T16_STAGE_MAT, 1); }
                s16_chain<2, 2, 1>(dx, lds, T16_L0, L.trp, dyS);
            } else {
                // ================= narrow layers: heads (32 -> 7) and rgb (64 -> 3) =================
                L = fresh_lane(L);
                // one 16-row block: rows 0-3 sigma, dif; 4-6 tint; 8-10 rgb  (lane group q holds rows 4q .. 4q+3)
                t16_h8 narB;   // B operand of the transposed products (second block of the k-step = zeros)
                {
                    v4f nar;
    #pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float r1 = g < 3 ? gh[4 + g] : 0.0f, r2 = g < 3 ? gs3[g] : 0.0f;
                        nar[g] = q == 0 ? gh[g] : (q == 1 ? r1 : (q == 2 ? r2 : 0.0f));
                    }
                    const v4f zero = { 0, 0, 0, 0 };
                    narB = t16_hi(nar, zero);
                    t16_stage_put(stY, L, 0, lo4(narB));
                    t16_stage_put(stY, L, 2, lo4(Hh[0]));   // X operand of the heads' weight gradient: H[:32] in blocks 2, 3 of Y
                    t16_stage_put(stY, L, 3, hi4(Hh[0]));
                    t16_stage_put(stX, L, 0, lo4(c1h[0]));  // X operand of the rgb layer's weight gradient
                    t16_stage_put(stX, L, 1, hi4(c1h[0]));
                    t16_stage_put(stX, L, 2, lo4(c1h[1]));
                    t16_stage_put(stX, L, 3, hi4(c1h[1]));
                }
                STEP_BARRIER();  // ---- A1
                if (wv == 0) wgrad<1, true>(gW_nar, gB_nar, stage, L, 0, 0, 2);          // heads: x = H[0:16]
                else if (wv == 1) { float dummy = 0.0f; wgrad<1, false>(gW_nar, dummy, stage, L, 0, 0, 3); }   // heads: x = H[16:32]
                else if (wv < 6) { float dummy = 0.0f; wgrad<1, false>(gW_nar, dummy, stage, L, 0, T16_STAGE_MAT, wv - 2); }  // rgb: x = c1 block wv-2
                // dv1 = (W_rgb^T gs3) * G'(v1)
                t16_h8 dyB[2];
                {
                    v4f dc[4];
                    zero4(dc);
                    t16_chain<4, 1>(dc, lds, T16T_D2, L.lo16, &narB);
    #pragma unroll
                    for (int b = 0; b < 4; ++b) dc[b] = mul_dg(dc[b], dgv1[b]);
                    dyB[0] = t16_hi(dc[0], dc[1]);
                    dyB[1] = t16_hi(dc[2], dc[3]);
                }
                STEP_BARRIER();  // ---- B1
                // ================= Directional_MLP.mlp.2 (64 -> 64) =================
                L = fresh_lane(L);
    #pragma unroll
                for (int t = 0; t < 2; ++t) {
                    t16_stage_put(stY, L, 2 * t, lo4(dyB[t]));
                    t16_stage_put(stY, L, 2 * t + 1, hi4(dyB[t]));
                    t16_stage_put(stX, L, 2 * t, lo4(c0h[t]));
                    t16_stage_put(stX, L, 2 * t + 1, hi4(c0h[t]));
                }
                STEP_BARRIER();  // ---- A2
                if (cb == 0) wgrad<2, true>(gW_D1, gB_D1, stage, L, rb, T16_STAGE_MAT, 0);
                else { float dummy = 0.0f; wgrad<2, false>(gW_D1, dummy, stage, L, rb, T16_STAGE_MAT, 2); }
                {
                    v4f dc[4];
                    zero4(dc);
                    t16_chain<4, 2>(dc, lds, T16T_D1, L.lo16, dyB);
    #pragma unroll
                    for (int b = 0; b < 4; ++b) dc[b] = mul_dg(dc[b], dgv0[b]);   // dv0
                    if (POSE) {  // 16 values x 16 samples -> lane c keeps the tile's sum of value c (= block c >> 2, register c & 3)
                        float v[16];
    #pragma unroll
                        for (int i = 0; i < 16; ++i) v[i] = dc[i >> 2][i & 3];
    #pragma unroll
                        for (int off = 8, half = 8; off > 0; off >>= 1, half >>= 1) {
                            const bool up = (c & off) != 0;
    #pragma unroll
                            for (int j = 0; j < half; ++j) {
                                const float mine = up ? v[j + half] : v[j], send = up ? v[j] : v[j + half];
                                v[j] = mine + __shfl_xor(send, off, 16);
                            }
                        }
                        pose_rs += v[0];
                    }
                    dyB[0] = t16_hi(dc[0], dc[1]);
                    dyB[1] = t16_hi(dc[2], dc[3]);
                }
                STEP_BARRIER();  // ---- B2
                // ================= Directional_MLP.mlp.0 (32 of its 48 inputs; the SH part per ray) =================
                L = fresh_lane(L);
    #pragma unroll
                for (int t = 0; t < 2; ++t) {
                    t16_stage_put(stY, L, 2 * t, lo4(dyB[t]));
                    t16_stage_put(stY, L, 2 * t + 1, hi4(dyB[t]));
                }
                t16_stage_put(stX, L, 0, lo4(Hh[1]));
                t16_stage_put(stX, L, 1, hi4(Hh[1]));
                {   // the SH part of the layer's input is constant along the ray: staged as 16 more "units" (block 2 of X)
                    const float4 shq = *reinterpret_cast<const float4 *>(shbuf + wv * 16 + 4 * q);
                    t16_stage_put(stX, L, 2, t16_hi4(v4f{ shq.x, shq.y, shq.z, shq.w }));
                }
                STEP_BARRIER();  // ---- A3
                if (cb == 0) wgrad<2, true, 2>(gW_D0, gB_D0, stage, L, rb, T16_STAGE_MAT, 0);   // x = H[32:48] and SH
                else { float dummy = 0.0f; wgrad<1, false>(gW_D0, dummy, stage, L, rb, T16_STAGE_MAT, 1); }   // x = H[48:64]
                {
                    v4f dH[4];
                    zero4(dH);
                    t16_chain<2, 2>(&dH[2], lds, T16T_D0, L.lo16, dyB);     // dH[32:64] = W_D0[:, :32]^T dv0
                    t16_chain<2, 1>(&dH[0], lds, T16T_HEAD, L.lo16, &narB);  // dH[0:32] = heads^T gh
                    dyB[0] = t16_hi(dH[0], dH[1]);
                    dyB[1] = t16_hi(dH[2], dH[3]);
                }
                STEP_BARRIER();  // ---- B3
                // ================= Spatial_MLP.mlp.2 (64 -> 64, linear) =================
                L = fresh_lane(L);
    #pragma unroll
                for (int t = 0; t < 2; ++t) {
                    t16_stage_put(stY, L, 2 * t, lo4(dyB[t]));
                    t16_stage_put(stY, L, 2 * t + 1, hi4(dyB[t]));
                    t16_stage_put(stX, L, 2 * t, lo4(a0h[t]));
                    t16_stage_put(stX, L, 2 * t + 1, hi4(a0h[t]));
                }
                STEP_BARRIER();  // ---- A4
                if (cb == 0) wgrad<2, true>(gW_L1, gB_L1, stage, L, rb, T16_STAGE_MAT, 0);
                else { float dummy = 0.0f; wgrad<2, false>(gW_L1, dummy, stage, L, rb, T16_STAGE_MAT, 2); }
                {
                    v4f dc[4];
                    zero4(dc);
                    t16_chain<4, 2>(dc, lds, T16T_L1, L.lo16, dyB);
    #pragma unroll
                    for (int b = 0; b < 4; ++b) dc[b] = mul_dg(dc[b], dg0[b]);   // du0
                    dyB[0] = t16_hi(dc[0], dc[1]);
                    dyB[1] = t16_hi(dc[2], dc[3]);
                }
                STEP_BARRIER();  // ---- B4
                // ================= Spatial_MLP.mlp.0 (32 -> 64) =================
                L = fresh_lane(L);
    #pragma unroll
                for (int t = 0; t < 2; ++t) {
                    t16_stage_put(stY, L, 2 * t, lo4(dyB[t]));
                    t16_stage_put(stY, L, 2 * t + 1, hi4(dyB[t]));
                }
                t16_stage_put(stX, L, 0, lo4(xh));
                t16_stage_put(stX, L, 1, hi4(xh));
                STEP_BARRIER();  // ---- A5
                if (cb == 0) wgrad<1, true>(gW_L0, gB_L0, stage, L, rb, T16_STAGE_MAT, 0);
                else { float dummy = 0.0f; wgrad<1, false>(gW_L0, dummy, stage, L, rb, T16_STAGE_MAT, 1); }
                t16_chain<2, 2>(dx, lds, T16T_L0, L.lo16, dyB);
            }
            dx[0] *= isc;
            dx[1] *= isc;
            if (POSE && a.g_raypos && live && active) {
                // position path (here, where only dX is live): this lane's four levels' Jacobians (12 loads in flight together),
                // contracted with its dX.  The forward's lane 32 h + (s & 31) wrote its 8 levels 4 (j >> 1) + 2 h + (j & 1) into
                // the 32-sample tile's [8][4][64] block of packed entries; register g of dX block e = feature g & 1 of level jj = 2e + (g >> 1).
                uint4 jw[4];
#pragma unroll
                for (int lv = 0; lv < 4; ++lv) {
                    const int j8l = 4 * (q & 1) + lv, lvl = 4 * (j8l >> 1) + 2 * (q >> 1) + (j8l & 1);
                    const int hh = (lvl >> 1) & 1, j = 2 * (lvl >> 2) + (lvl & 1);
                    const uint32_t *jr = a.f.jstash + (((size_t)ray * ((S + 31) >> 5) + (s >> 5)) * 8 + j) * (4 * 64) + 32 * hh + (s & 31);
                    jw[lv] = make_uint4(jr[0], jr[64], jr[128], jr[192]);
                }
                float gxk[3] = { 0, 0, 0 };
#pragma unroll
                for (int lv = 0; lv < 4; ++lv) {
                    // entry = (df0/dx, df0/dy, df0/dz, df1/dx, df1/dy, df1/dz) * 2^(E - 19) (render_device.h jst_pack)
                    float jq[6], jsc;
                    jst_unpack(jw[lv].x, jw[lv].y, jw[lv].z, jw[lv].w, jq, jsc);
                    const float fx = dx[lv >> 1][2 * (lv & 1)] * jsc, fy = dx[lv >> 1][2 * (lv & 1) + 1] * jsc;
                    gxk[0] += fx * jq[0] + fy * jq[3];
                    gxk[1] += fx * jq[1] + fy * jq[4];
                    gxk[2] += fx * jq[2] + fy * jq[5];
                }
                if (a.f.contract_mode == 1) {  // through contract_bg: r g + (g . x) r' sign(x_m) e_m, r = (2 - 1/|x|_inf) / |x|_inf
                    float xk[3];
#pragma unroll
                    for (int k = 0; k < 3; ++k) xk[k] = (o[k] + z * d[k] - a.f.min_bbox[k]) * a.f.inv_size4[k] - 2.0f;
                    const float ax = fabsf(xk[0]), ay = fabsf(xk[1]), az = fabsf(xk[2]);
                    const float linf = fmaxf(ax, fmaxf(ay, az));
                    const int mk = (ax >= ay && ax >= az) ? 0 : (ay >= az ? 1 : 2);   // torch.max: the first maximum
                    const float il = 1.0f / linf, r = (2.0f - il) * il, dr = (-2.0f + 2.0f * il) * il * il;
                    const float dot = gxk[0] * xk[0] + gxk[1] * xk[1] + gxk[2] * xk[2];
#pragma unroll
                    for (int k = 0; k < 3; ++k) gxk[k] *= r;
                    const float extra = dot * dr * (xk[mk] < 0.0f ? -1.0f : 1.0f);
                    gxk[0] += mk == 0 ? extra : 0.0f;
                    gxk[1] += mk == 1 ? extra : 0.0f;
                    gxk[2] += mk == 2 ? extra : 0.0f;
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float gk = gxk[k] * a.f.inv_size4[k];
                    pose_go[k] += gk;
                    pose_gd[k] += z * gk;
                }
            }
            // (no barrier here: the next tile's staging writes come after its barrier S)
            STAMP(21);

            if (!(SPLIT && T16_LOAD_AT) && tile > 0) nxt = load_tile(tile - 1);   // before this tile's record stores (see load_tile)
            if constexpr (SPLIT) {
                // ================= feature gradients: emitted here (the split kernel has no registers to carry them into the
                // next tile: 11 held across its forward recompute cost more in spills than the overlap gains)
                if (park_wave) {
                    const int lp = fresh(lane);
                    float4 *pk = reinterpret_cast<float4 *>(lds + LD::kPark + (wv - 4) * 2048);
                    pk[lp] = make_float4(dx[0][0], dx[0][1], dx[0][2], dx[0][3]);
                    pk[64 + lp] = make_float4(dx[1][0], dx[1][1], dx[1][2], dx[1][3]);
                    pz = z;
                    ptile = tile;
                } else {
                    float pe[3] = { 0, 0, 0 };
                    if (a.recs) contract_point(a.f, o, d, z, pe);
                    // (tried: the cursor round trips of two or four levels in flight before the first record is stored -- 24 / 48 spilled
                    // registers, 5.2 -> 5.6 ms; the four levels go one after the other)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) emit_level(tile, jj, dx[0], dx[1], pe);
                }
            } else {
                // ================= feature gradients: emitted during the next tile (emit_level) =================
                pdx0 = dx[0];
                pdx1 = dx[1];
                ptile = tile;
                if (a.recs) contract_point(a.f, o, d, z, ppe);
            }
        }
        if constexpr (!SPLIT) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) emit_level(ptile, jj, pdx0, pdx1, ppe);   // the ray's first tile
        } else if (park_wave) emit_parked();   // the ray's first tile
        if (POSE && active) {  // the ray's pose-gradient sums
            const int ln = fresh(lane), c = ln & 15, q = ln >> 4;
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) pose_dn += __shfl_xor(pose_dn, off, 16);
            float *rsum = a.g_rowsum + (size_t)ray * 2 * 64;   // [2][64]: the second partial row is the h3 kernel's other half-wave
            const int unit = 16 * (c >> 2) + 4 * q + (c & 3);
            rsum[unit] = pose_rs * isc;
            rsum[64 + unit] = 0.0f;
            if (a.g_raypos) {
#pragma unroll
                for (int k = 0; k < 3; ++k)
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) {
                        pose_go[k] += __shfl_xor(pose_go[k], off, 64);
                        pose_gd[k] += __shfl_xor(pose_gd[k], off, 64);
                    }
                if (ln == 0) {
                    float2 *gr = reinterpret_cast<float2 *>(a.g_raypos + (size_t)ray * 6);
                    gr[0] = make_float2(pose_go[0], pose_go[1]);
                    gr[1] = make_float2(pose_go[2], pose_gd[0]);
                    gr[2] = make_float2(pose_gd[1], pose_gd[2]);
                }
            }
            if (ln == 0) {
                const int ncol = (S + 31) >> 5;                 // [B][ceil(S/32)] partials: all of it in column 0
                a.g_dnorm[(size_t)ray * ncol] = pose_dn;
                for (int t = 1; t < ncol; ++t) a.g_dnorm[(size_t)ray * ncol + t] = 0.0f;
            }
            SCANERF_STORE_GUARD();
        }
    }

#ifdef T16_STAMPS
    STAMP(22);
    if (lane == 0) {
        float *so = a.dw_partial + (size_t)(gridDim.x + blockIdx.x) * SCANERF_PARAMSIZE + wv * 32;
#pragma unroll
        for (int i = 0; i < 23; ++i) so[i] = (float)stamps[i];
    }
#endif
    if (a.recs) {  // launch-wide max |dL/dfeature| for the fixed-point scale of the accumulate pass
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) gmax = fmaxf(gmax, __shfl_xor(gmax, off, 64));
        if (lane == 0 && gmax > 0.0f) atomicMax(a.maxbits, __float_as_uint(gmax));
    }
    // ---- flush: the workgroup's partial sums in blob order, ONE row per workgroup (every entry has exactly one owner)
    const int c = lane & 15, q = lane >> 4;
    float *out = a.dw_partial + (size_t)blockIdx.x * SCANERF_PARAMSIZE;
    auto put64 = [&](const v4f &acc, int base, int cblk) {   // 64-output layer: acc = dW[n = 16rb + 4q + g][k = 16 cblk + c]
#pragma unroll
        for (int g = 0; g < 4; ++g) out[base + 64 + (16 * cblk + c) * 64 + 16 * rb + 4 * q + g] = acc[g] * isc;
    };
    put64(gW_D1[0], BLOB_D1, 2 * cb);
    put64(gW_D1[1], BLOB_D1, 2 * cb + 1);
    put64(gW_L1[0], BLOB_S1, 2 * cb);
    put64(gW_L1[1], BLOB_S1, 2 * cb + 1);
    put64(gW_D0[0], BLOB_D0, cb);
    if (cb == 0) put64(gW_D0[1], BLOB_D0, 2);   // SH part: input index 32 + c
    {
        const int kin = t16_pos_to_input(t16_l0_row_to_pos(cb, c));
#pragma unroll
        for (int g = 0; g < 4; ++g) out[BLOB_S0 + 64 + kin * 64 + 16 * rb + 4 * q + g] = gW_L0[0][g] * isc;
    }
    auto rowtotal = [&](float v) {  // sum over the 4 lane groups: lane = unit
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        return v;
    };
    if (cb == 0) {
        float v;
        v = rowtotal(gB_D1) * isc;
        if (q == 0) out[BLOB_D1 + 16 * rb + c] = v;
        v = rowtotal(gB_L1) * isc;
        if (q == 0) out[BLOB_S1 + 16 * rb + c] = v;
        v = rowtotal(gB_L0) * isc;
        if (q == 0) out[BLOB_S0 + 16 * rb + c] = v;
        v = rowtotal(gB_D0) * isc;
        if (q == 0) out[BLOB_D0 + 16 * rb + c] = v;
    }
    if (wv < 2) {  // heads: acc = d[row 4q + g][H unit 16 wv + c]; rows 0 sigma, 1-3 dif, 4-6 tint
        const int k = 16 * wv + c;
        if (q == 0) {
            out[BLOB_SIG + 1 + k] = gW_nar[0][0] * isc;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) out[BLOB_DIF + 3 + k * 3 + ch] = gW_nar[0][1 + ch] * isc;
        } else if (q == 1) {
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) out[BLOB_TINT + 3 + k * 3 + ch] = gW_nar[0][ch] * isc;
        }
    } else if (wv < 6 && q == 2) {  // rgb layer: rows 8-10, column = c1 unit 16(wv-2) + c
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) out[BLOB_D2 + 3 + (16 * (wv - 2) + c) * 3 + ch] = gW_nar[0][ch] * isc;
    }
    if (wv == 0) {  // narrow biases = row sums of the narrow block
        const float v = rowtotal(gB_nar) * isc;
        if (q == 0) {
            if (c == 0) out[BLOB_SIG] = v;
            else if (c < 4) out[BLOB_DIF + c - 1] = v;
            else if (c < 7) out[BLOB_TINT + c - 4] = v;
            else if (c >= 8 && c < 11) out[BLOB_D2 + c - 8] = v;
        }
    }
}

}  // namespace

namespace scanerf {

int launch_pack_decoder_t16(const float *blob, const float *wf, char *out, hipStream_t st)
{
    hipLaunchKernelGGL(k_pack_decoder_t16, dim3((62 * 512 + 272 + 255) / 256), dim3(256), 0, st, blob, wf, out);
    // ... and the images of its split-gradient variant, right behind (render_common.h WS_S16)
    hipLaunchKernelGGL(k_pack_decoder_s16, dim3((38 * 512 + 272 + 255) / 256), dim3(256), 0, st, blob, wf, out + T16_BYTES);
    return 0;
}

int launch_render_bwd_t16(const BwdArgs &a_in, int feat_dtype, int blocks, size_t lds_extra, hipStream_t st, bool split)
{
    size_t lds_bytes = (size_t)(split ? Lds<true>::kCursor : Lds<false>::kCursor) + lds_extra;
    BwdArgs a = a_in;
    // skewed emission (the t16s kernel's waves 4-7 park a tile's dX in LDS): whenever its 8 KB fit next to the record cursors
    // (T <= 2^20 entries per level at 2^13-entry buckets); SCANERF_BWD_PARK=0 switches it off (A/B timing)
    a.park = split && a.recs && lds_bytes + Lds<true>::kParkBytes <= 160 * 1024 && tune_int("SCANERF_BWD_PARK", 1) != 0;
    if (a.park) lds_bytes += Lds<true>::kParkBytes;
    SCANERF_REQUIRE(lds_bytes <= 160 * 1024, "render_backward(t16): %zu B of LDS needed (table too large for the fused scatter)", lds_bytes);
    SCANERF_REQUIRE(a.xstash, "render_backward(t16): needs the x-stash");
    SCANERF_REQUIRE((a.g_dnorm != nullptr) == (a.g_rowsum != nullptr), "render_backward(t16): g_dnorm and g_rowsum come together");
    SCANERF_REQUIRE(!a.g_raypos || (a.g_dnorm && a.f.jstash), "render_backward(t16): g_raypos needs g_dnorm / g_rowsum and the forward's jstash");
#define SCANERF_LAUNCH_BWD(DT, R8, PO, SP)                                                                         \
    {                                                                                                              \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_render_bwd_t16<DT, R8, PO, SP>),      \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);            \
        SCANERF_REQUIRE(e == hipSuccess, "render_backward(t16): cannot reserve %zu B of LDS: %s", lds_bytes,        \
                        hipGetErrorString(e));                                                                     \
        hipLaunchKernelGGL((k_render_bwd_t16<DT, R8, PO, SP>), dim3(blocks), dim3(kThreads), lds_bytes, st, a);     \
    }
    (void)feat_dtype;  // the table is only read through the x-stash here
    const int rec = a.recs ? a.bins.rec8 : 0;   // record format as the plan decided (scatter_common.h fused_rec8)
    SCANERF_REQUIRE(!(split && rec == 1) && !(!split && rec == 2), "render_backward(t16): record format %d does not belong to this arithmetic", rec);
    if (split) {
        if (a.g_dnorm) {
            if (rec == 2) SCANERF_LAUNCH_BWD(SCANERF_F32, 2, true, true)
            else SCANERF_LAUNCH_BWD(SCANERF_F32, 0, true, true)
        } else if (rec == 2) SCANERF_LAUNCH_BWD(SCANERF_F32, 2, false, true)
        else SCANERF_LAUNCH_BWD(SCANERF_F32, 0, false, true)
    } else if (a.g_dnorm) {
        if (rec == 1) SCANERF_LAUNCH_BWD(SCANERF_F32, 1, true, false)
        else SCANERF_LAUNCH_BWD(SCANERF_F32, 0, true, false)
    } else if (rec == 1) SCANERF_LAUNCH_BWD(SCANERF_F32, 1, false, false)
    else SCANERF_LAUNCH_BWD(SCANERF_F32, 0, false, false)
#undef SCANERF_LAUNCH_BWD
    return 0;
}

}  // namespace scanerf

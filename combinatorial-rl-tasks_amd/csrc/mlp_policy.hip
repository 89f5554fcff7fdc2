// mlp_policy.hip -- ZoneEnvModel + actor forward on bf16 MFMA (gfx950), SURVEY.md 8(f) row 1.
//
// Reference (all float32 torch):  main/src/env_model.py:48-79
//     zone_net_:  Linear(8+F, h) ReLU Linear(h, h) ReLU Linear(h, h)   on every [obs, zone row]
//     zone_emb = zone_net_(...).sum(dim=1) / n_zones ;  combine_net_: Linear(8+h, h) on [obs, zone_emb]
// main/src/flat_model.py:24-37 + policy_network.py:12-53 (Box action space):
//     enc_: Linear(h, h) ReLU ;  mu = 2 (sigmoid(mu_(x)) - 0.5) ;  std = sigmoid(std_(x)) + 1e-3
// h = 185 (utils/agent.py:17), padded here to 192 = 6 MFMA tiles.
//
// Formulation.  v_mfma_f32_32x32x16_bf16 leaves a 32x32 result with its COLUMN on the lane and its
// ROWS in the 16 accumulator registers, and such a tile can feed the next MFMA directly -- as the B
// operand of  Y = A X  or the A operand of  Z = X^T B  -- when the other operand's k order follows the
// accumulator order (cdna_hip_programming.md section 3, "an accumulator tile as the next MFMA's
// operand").  So activations never leave registers:
//   kernel 1 (k_mlp_zone1; k_mlp_zone is its predecessor), a wave per 64 envs, batch = the envs' 64 Z zone rows,
//   32 per tile:
//     X1   = relu(W1 X0)            features in registers, zone row on the lane     (6 MFMA / tile)
//     H2^T = relu(X1^T W2^T)        zone row in registers, feature on the lane      (72 MFMA / tile)
//     mean over the zone rows of an env = one more product P += S relu(H2^T) with a 0/1 selection
//     matrix S[env slot][zone row] (12 MFMA / tile), accumulated per 32 envs; tiles are zone-major in
//     k_mlp_zone1 (tile t = zone t of 32 envs), which makes S the identity.
//     The third zone_net_ layer has no activation, so the mean commutes with it and moves into
//     kernel 2 -- 25x fewer rows for that layer (float reassociation only).
//   kernel 2 (k_mlp_head), batch = envs:  c = Wc [W3 mean(H2) + b3; obs] (one folded layer);
//     a = relu(Wa c); [mu; std] = Wh a  -- a chain of Y = A X products, each taking the previous
//     accumulator as its B operand.
// Biases ride in a padded k slot: a constant-1 feature is carried through every layer (slot 15 of the
// 16-wide input, feature h of every hidden layer), and the bias is that slot's weight column.
// W2 / W1 fragments live in LDS (78 KB per workgroup); kernel 2 stages one layer's fragments at a time.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <cstring>

#include "mlp_policy.hpp"
#include "mlp_head_out.hpp"

namespace zenvk {
namespace {

// The 16-bit element type of every operand.  This file is compiled twice: as it stands (bf16: float32's range, 8
// significant bits) and through mlp_policy_f16.hip with MLP_ELEM_F16 (float16: 11 significant bits -- the same kernels
// at the same speed with an eighth of the rounding error, ZENV_MLP_F16 -- and float16's range, which zenv_mlp_load
// bounds at load time for the zone layers and k_mlp_head watches at run time).  The names below keep "bf16".
#ifndef MLP_ELEM_F16
#define MLP_ELEM_F16 0
#endif
#if MLP_ELEM_F16
typedef _Float16 elem_t;
#define MLP_SFX "f16"
#else
typedef __bf16 elem_t;
#define MLP_SFX "bf16"
#endif
typedef __attribute__((__vector_size__(8 * sizeof(elem_t)))) elem_t bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
constexpr int kWave = 64;
constexpr int NT = kMlpNT, KS = kMlpKS, HP = kMlpHP;
#ifndef MLP_KERNEL
#define MLP_KERNEL 1      // 1: k_mlp_zone1 (one pipelined wave per SIMD; shipped), 0: k_mlp_zone (its predecessor, a
#endif                    //    wave pair per SIMD -- compiled only on request, for comparison)

// Diagnostic builds only (scripts/probes/k4_stamps.hip): s_memtime stamps of one wave of block 0, per row tile.
#ifdef MLP_STAMP
#ifndef MLP_STAMP_SLOTS
#define MLP_STAMP_SLOTS 0xFFFF      // bit s: stamp slot s is recorded
#endif
__device__ unsigned long long *g_k4_stamps;
#define KSTAMP(slot)                                                                                       \
    do {                                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
        if (blockIdx.x == 0 && wave == (MLP_STAMP) && stamp_it < 64 && (MLP_STAMP_SLOTS >> (slot)) & 1) {  \
            unsigned long long t_;                                                                         \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                    \
            if (lane == 0) g_k4_stamps[stamp_it * 16 + (slot)] = t_;                                       \
            if ((slot) == 0) {      /* the 100 MHz clock beside it: what frequency do the ticks run at? */ \
                asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");            \
                if (lane == 0) g_k4_stamps[stamp_it * 16 + 15] = t_;                                       \
            }                                                                                              \
        }                                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
    } while (0)
#define ZT_STAMP_ARGS , wave, stamp_it++
// two stamps per launch around the whole tile loop of block 0 / wave MLP_STAMP (slots 60, 61 of the table): cycles and
// 100 MHz ticks without touching the loop
#define KSTAMP_WHOLE(which)                                                                                \
    do {                                                                                                   \
        if (blockIdx.x == 0 && wave == (MLP_STAMP)) {                                                      \
            unsigned long long t_, r_;                                                                     \
            asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_), "=s"(r_)::"memory"); \
            if (lane == 0) { g_k4_stamps[(60 + (which)) * 16] = t_; g_k4_stamps[(60 + (which)) * 16 + 15] = r_; } \
        }                                                                                                  \
    } while (0)
#else
#define KSTAMP_WHOLE(which) do { } while (0)
#define KSTAMP(slot) do { } while (0)
#define ZT_STAMP_ARGS
#endif

__device__ __forceinline__ bf16x8 as_frag(const uint4 v) { return __builtin_bit_cast(bf16x8, v); }
__device__ __forceinline__ f32x16 mfma(const bf16x8 a, const bf16x8 b, const f32x16 c)
{
#if MLP_ELEM_F16
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
#else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
#endif
}
__device__ __forceinline__ f32x16 zero16()
{
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}
// A 32x32 accumulator tile provides two k-steps of the next product: registers 8s..8s+7 -> step s.
// ReLU is applied AFTER the conversion, on the packed bf16 pairs, as a signed 16-bit max with 0
// (v_pk_max_i16): a bf16 is negative exactly when its bit pattern is a negative int16, and rounding
// to bf16 never changes the sign -- one instruction per two elements instead of two per element.
typedef __attribute__((__vector_size__(8 * sizeof(short)))) short s16x8;
__device__ __forceinline__ bf16x8 relu_bf16(const bf16x8 f)
{
    const s16x8 v = __builtin_bit_cast(s16x8, f);
    const s16x8 z = { 0, 0, 0, 0, 0, 0, 0, 0 };
    return __builtin_bit_cast(bf16x8, __builtin_elementwise_max(v, z));
}
// two floats -> one dword of two bf16 (round to nearest even): ONE v_cvt_pk_bf16_f32.  Converting element
// by element makes the compiler emit a cvt per element plus a v_perm to pair them up.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef elem_t bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_bf16(float a, float b)
{
    const f32x2_t v = { a, b };
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ bf16x8 frag_from_8(float a0, float a1, float a2, float a3, float a4, float a5, float a6,
                                              float a7)
{
    uint4 u;
    u.x = pk_bf16(a0, a1);
    u.y = pk_bf16(a2, a3);
    u.z = pk_bf16(a4, a5);
    u.w = pk_bf16(a6, a7);
    return as_frag(u);
}
__device__ __forceinline__ void acc_to_frags(const f32x16 &acc, bool relu, bf16x8 &f0, bf16x8 &f1)
{
    f0 = frag_from_8(acc[0], acc[1], acc[2], acc[3], acc[4], acc[5], acc[6], acc[7]);
    f1 = frag_from_8(acc[8], acc[9], acc[10], acc[11], acc[12], acc[13], acc[14], acc[15]);
    if (relu) {
        f0 = relu_bf16(f0);
        f1 = relu_bf16(f1);
    }
}
__device__ __forceinline__ bf16x8 frag_from_floats(const float4 lo, const float4 hi)
{
    return frag_from_8(lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w);
}

// ------------------------------------------------------------------------------------------ kernel 1
// raw inputs of one 32-row tile as a lane holds them: lane half 0 the env's obs, half 1 the zone row.
// The SAME three load instructions in both lane halves (no divergent branches: with one branch per half the
// compiler's wait for the loads landed right behind their issue -- a full memory round trip exposed per tile),
// never past the end of a row; the values are only looked at by row_frag(), one tile later.
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));
#if MLP_KERNEL == 0
struct RawRow {
    f32x4u a;       // floats 0-3
    f32x2u b, c;    // floats 4-5; obs: floats 6-7, zone row: its last two floats
    bool valid;
};
template <int ZT, int F>
__device__ __forceinline__ RawRow load_row(const float *__restrict__ obs, const float *__restrict__ zrows, int env0,
                                           int Z, int row, int n_rows, int h)
{
    RawRow x;
    x.valid = row < n_rows;
    const int rr = x.valid ? row : 0;   // any readable row; row_frag() zeroes it
    const float *p = h ? zrows + (size_t)rr * F : obs + (size_t)(env0 + rr / Z) * 8;
    x.a = *reinterpret_cast<const f32x4u *>(p);
    x.b = *reinterpret_cast<const f32x2u *>(p + 4);
    x.c = *reinterpret_cast<const f32x2u *>(p + (h ? F - 2 : 6));
#if defined(MLP_EXP) && (MLP_EXP & 16)     // diagnostic: no global loads for the rows
    x.a = f32x4u{ 1.f, 0.f, 0.f, 0.f }; x.b = f32x2u{ 0.f, 0.f }; x.c = x.b;
#endif
    return x;
}
// the tile's B operand of layer 1: k = 0..7 obs (half 0) / k = 8..14 zone row, k = 15 the bias slot (half 1)
template <int F>
__device__ __forceinline__ bf16x8 row_frag(const RawRow &x, int h)
{
    float v6 = x.c.x, v7 = x.c.y;
    if (h) {
        v6 = F == 7 ? x.c.y : 0.f;
        v7 = 1.0f;
    }
    const bool ok = x.valid;
    return frag_from_8(ok ? x.a.x : 0.f, ok ? x.a.y : 0.f, ok ? x.a.z : 0.f, ok ? x.a.w : 0.f, ok ? x.b.x : 0.f,
                       ok ? x.b.y : 0.f, ok ? v6 : 0.f, ok ? v7 : 0.f);
}

// 8 waves per workgroup, two per SIMD: wave w works on the 64-env group (w & 3) of the workgroup and on
// the output-feature half (w >> 2) -- tiles n = 3 half .. 3 half + 2 of layer 2 and of the pooled mean.  Both
// waves of a pair compute X1 (6 MFMA per 32 rows, 7 % of the work) so that neither needs the other's
// registers; in return a wave fits in 256 registers and the SIMD always has a second wave to issue from
// while one waits for an MFMA result, an LDS fragment or the next rows (one wave per SIMD measured: matrix
// pipe 49 % busy, 29 % of the cycles in s_waitcnt, 30 % in issue stalls).
#ifndef MLP_SPLIT
#define MLP_SPLIT 2                      // waves per 64-env group = waves per SIMD
#endif
constexpr int kZoneWaves = 4 * MLP_SPLIT;
constexpr int NH = NT / MLP_SPLIT;       // output tiles per wave

template <int ZT, int F>
__global__ __launch_bounds__(kZoneWaves * kWave) __attribute__((amdgpu_waves_per_eu(MLP_SPLIT, MLP_SPLIT)))
void k_mlp_zone(MlpImages img, int N, int Z_rt, const float *__restrict__ obs, const float *__restrict__ zone_obs,
                elem_t *__restrict__ pooled)
{
    extern __shared__ uint4 lds[];
    uint4 *w2s = lds;                       // [NT*KS][64]
    uint4 *w1s = lds + NT * KS * kWave;     // [NT][64]
    for (int i = threadIdx.x; i < NT * KS * kWave; i += kZoneWaves * kWave)
        w2s[i] = reinterpret_cast<const uint4 *>(img.w2)[i];
    for (int i = threadIdx.x; i < NT * kWave; i += kZoneWaves * kWave) w1s[i] = reinterpret_cast<const uint4 *>(img.w1)[i];
    __syncthreads();

    const int Z = ZT > 0 ? ZT : Z_rt;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int n0 = NH * (wave >> 2);        // this wave's first output tile
    const int env0 = (blockIdx.x * 4 + (wave & 3)) * kWave;
    if (env0 >= N) return;
    const int n_env = min(kWave, N - env0), n_rows = n_env * Z;
    const float inv_z = 1.0f / (float)Z;
    const float *zrows = zone_obs + (size_t)env0 * Z * F;
    const elem_t one = (elem_t)1.0f, nil = (elem_t)0.0f;
    // layer 1's six fragments stay in registers for the whole kernel (read just in time from LDS they cost an
    // exposed LDS round trip per MFMA)
    bf16x8 w1f[NT];
#pragma unroll
    for (int m = 0; m < NT; ++m) w1f[m] = as_frag(w1s[m * kWave + lane]);

    // (Starting the second wave of each SIMD half a tile late, so that the pair is not in its VALU phases at
    // the same moments, changed nothing: 165.6 us per step at any offset; neither did a static priority for the
    // younger half of the workgroup.)

#ifdef MLP_STAMP
    int stamp_it = 0;
#endif
    // two groups of 32 envs: a group's zone rows all pool into ONE 32-env accumulator tile set
    for (int e_base = 0; e_base < n_env; e_base += 32) {
        const int g_lo = e_base * Z, g_hi = min(e_base + 32, n_env) * Z;     // the group's rows
        f32x16 pool[NH];
#pragma unroll
        for (int n = 0; n < NH; ++n) pool[n] = zero16();
        RawRow nxt = load_row<ZT, F>(obs, zrows, env0, Z, g_lo + r, n_rows, h);

        for (int b = g_lo; b < g_hi; b += 32) {
            KSTAMP(0);
            const bf16x8 x0 = row_frag<F>(nxt, h);
            // the next tile's rows are fetched while this one is in the matrix pipe (fetching three tiles ahead
            // was slower); issued strictly behind the use of the previous ones, so that the wait in front of
            // row_frag() never covers a load younger than a whole tile
            __builtin_amdgcn_sched_barrier(0);
            nxt = load_row<ZT, F>(obs, zrows, env0, Z, b + 32 + r, min(n_rows, g_hi), h);
            // layer 2's first fragments: issued here so that they land during layer 1
            bf16x8 wf[NH + 1][KS];          // statically indexed: two tiles' worth live at a time
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) wf[0][kk] = as_frag(w2s[(n0 * KS + kk) * kWave + lane]);
            // ---- layer 1: X1 = relu(W1 X0), features in registers
            bf16x8 xa[KS];
#if defined(MLP_EXP) && (MLP_EXP & 2)      // diagnostic: no layer 1
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) xa[kk] = x0;
#else
#pragma unroll
            for (int m = 0; m < NT; ++m) {
                const f32x16 acc1 = mfma(w1f[m], x0, zero16());
                acc_to_frags(acc1, true, xa[2 * m], xa[2 * m + 1]);
            }
#endif
            KSTAMP(1);
            // ---- mean over an env's rows = one more product: P += S relu(H2^T), S[env slot][row] = 1 when
            // the row belongs to env e_base + slot.  Lane (slot r, half h) element j of k-step s is tile row
            // 16 s + 8 (j >> 2) + 4 h + (j & 3) -- the accumulator order of H2^T.
            bf16x8 ind[2];
            {
                const int lo = (e_base + r) * Z - b;   // first row of this lane's env, tile-relative
#pragma unroll
                for (int sgm = 0; sgm < 2; ++sgm)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int pos = 16 * sgm + 8 * (j >> 2) + 4 * h + (j & 3);
                        ind[sgm][j] = ((unsigned)(pos - lo) < (unsigned)Z) ? one : nil;
                    }
            }
            // ---- layer 2: H2^T = relu(X1^T W2^T), zone row in registers, feature on the lane; then the pooling.
            // Software-pipelined over the wave's output tiles: region n issues the LDS reads of tile n+1's
            // fragments, then tile n's 12 MFMAs, then tile n-1's conversion + pooling MFMAs.
            f32x16 acc2[NH];
            KSTAMP(2);
#pragma unroll
            for (int n = 0; n <= NH; ++n) {
                if (n < NH) {
                    if (n + 1 < NH) {
#pragma unroll
                        for (int kk = 0; kk < KS; ++kk)
                            wf[n + 1][kk] = as_frag(w2s[((n0 + n + 1) * KS + kk) * kWave + lane]);
                    }
                    acc2[n] = zero16();
#pragma unroll
                    for (int kk = 0; kk < KS; ++kk)
#if defined(MLP_EXP) && (MLP_EXP & 4)      // diagnostic: one fragment per output tile instead of 12
                        acc2[n] = mfma(xa[kk], wf[n][0], acc2[n]);
#else
                        acc2[n] = mfma(xa[kk], wf[n][kk], acc2[n]);   // (two independent chains per tile: slower)
#endif
                }
                if (n > 0) {
#if defined(MLP_EXP) && (MLP_EXP & 1)      // diagnostic: no conversion / pooling product
                    pool[n - 1] = acc2[n - 1];
#else
                    bf16x8 f0, f1;
                    acc_to_frags(acc2[n - 1], true, f0, f1);
                    pool[n - 1] = mfma(ind[0], f0, pool[n - 1]);
                    pool[n - 1] = mfma(ind[1], f1, pool[n - 1]);
#endif
                }
                // region boundary: keeps the scheduler from hoisting every later tile's fragment reads up here
                __builtin_amdgcn_sched_barrier(0);
                KSTAMP(3 + n);
            }
#ifdef MLP_STAMP
            ++stamp_it;
#endif
        }
        // ---- the group's means: accumulator register i of lane half h is env slot (i&3) + 8 (i>>2) + 4 h
#pragma unroll
        for (int n = 0; n < NH; ++n)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int e = e_base + (i & 3) + 8 * (i >> 2) + 4 * h;
                // bf16: kernel 2 feeds it to an MFMA as it is (half the bytes of a float32 mean, both ways).  Even
                // lanes store their own and their neighbour's feature as one dword (2-byte stores were slower).
                const float v = pool[n][i] * inv_z;
                const float vn = __shfl_down(v, 1);
                if (e < n_env && !(r & 1))
                    *reinterpret_cast<uint32_t *>(pooled + (size_t)(env0 + e) * HP + 32 * (n0 + n) + r) = pk_bf16(v, vn);
            }
    }
}

#endif   // MLP_KERNEL == 0

// ------------------------------------------------------------------------------------------ kernel 1, v2
// k_mlp_zone1: ONE wave per SIMD (4 per workgroup, 512 registers each), software-pipelined by hand so that
// everything a matrix instruction does not depend on sits in its shadow.  What the stamps of the two-wave
// kernel showed (scripts/probes/k4_stamps.hip): a SIMD issues ONE vector instruction stream -- a VALU
// instruction costs 4 issue cycles, an MFMA 8 of its 32 -- and the two waves' streams simply added up (the
// layer-1 chain MFMA -> wait -> 16 conversions -> next MFMA ran 1000 cycles per tile, the selection matrix
// 430), so the matrix pipe idled half of the time.  Here:
//   * tiles are ZONE-major: tile t = zone t of the group's 32 envs, so row r of every tile belongs to env
//     slot r -- the selection matrix of the pooling product is the identity, a per-lane constant, and the
//     obs half of the layer-1 operand is loaded once per group;
//   * layer 1 of tile t+1 (6 MFMAs + 96 conversions) is issued inside the layer-2 chains of tile t, and
//     the conversion + pooling of output tile n-1 inside the chain of output tile n (also across tiles);
//   * the pooled accumulators (96 registers) live in AGPRs (inline-asm MFMA, "+a"), the W2 fragments are
//     read from LDS straight into AGPRs by the register allocator; VGPRs hold the activations only.
#ifndef MLP_POOL_VALU
#define MLP_POOL_VALU 0     // 0 (shipped): the mean over an env's zone rows as two more MFMAs per output tile against the identity
#endif                      // (bf16-rounded summands); 1: summed on the vector ALU in float32 -- measured and NOT faster, below
// p += relu(x) in float32: with zone-major tiles the pooling product's selection matrix is the identity, so the product
// is this register-wise sum -- 16 v_max_i32 (ReLU on the float's bits) + 8 v_pk_add_f32 per output tile instead of 16
// conversions + 2 MFMAs, with the layer-1 fragments moved to AGPRs (8 v_accvgpr_write) to make room for 96 float32 sums.
// Round 3, cycles per 32-row tile by two stamps around the whole loop (profiles/r03/k4_whole_loop_cycles.log): MFMA
// pooling 3 687, no pooling at all 3 302 (every MFMA costs its 32 cycles: 12 x 32 = 384), VALU pooling 4 253 (3 919 in
// the compiler's own order): a gap between two MFMAs of one wave hides about three vector / LDS instructions, the 48 + 12
// of this variant do not fit into 13 gaps, and the two pooling MFMAs were the cheaper way to buy two more gaps.
// (Written with __builtin_bit_cast on x[i] the loop compiled to sixteen copies of x[0]: keep the scalar temporaries.)
#if MLP_POOL_VALU
typedef float f32x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void pool_add(f32x16 &p, const f32x16 &x)
{
#pragma unroll
    for (int i = 0; i < 16; i += 2) {
        const float x0 = x[i], x1 = x[i + 1];
        const f32x2v v = { __int_as_float(max(__float_as_int(x0), 0)), __int_as_float(max(__float_as_int(x1), 0)) };
        f32x2v q = { p[i], p[i + 1] };
        q += v;
        p[i] = q.x;
        p[i + 1] = q.y;
    }
}
#endif
#ifndef MLP_POOL_SPARSE
#define MLP_POOL_SPARSE 1   // the pooling product as ONE v_smfmac_f32_32x32x32_bf16 per output tile (K = 32: the identity is 2:4
#endif                      // sparse) instead of two dense K = 16 products: every MFMA of this kernel costs its 32 cycles
// Operand layout of the sparse instruction, found by experiment (scripts/probes/smfmac_layout.hip, profiles/r03/
// smfmac_layout.log): B = the two dense K = 16 fragments back to back (registers 0-3 k-step 0, 4-7 k-step 1; lane half hb
// element e is k = 16 (e >> 3) + 8 hb + (e & 7)); A = lane (row, half ha) holds k in [16 ha, 16 ha + 16) as four groups
// of four with two kept values each (slots 2 g, 2 g + 1), the 2-bit position of slot i in bits [2 i + 1 : 2 i] of the
// index register.  The identity has one non-zero per row, so the result is that of the two dense products bit for bit.
typedef __attribute__((__vector_size__(16 * sizeof(elem_t)))) elem_t bf16x16;
__device__ __forceinline__ void pool_smfmac(f32x16 &p, const bf16x8 a, const bf16x8 b0, const bf16x8 b1, const int idx)
{
    const bf16x16 b = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
    asm volatile("s_nop 1\n\tv_smfmac_f32_32x32x32_" MLP_SFX " %0, %1, %2, %3" : "+a"(p) : "v"(a), "v"(b), "v"(idx));
}
#if !MLP_POOL_SPARSE
__device__ __forceinline__ void pool_mfma(f32x16 &p, const bf16x8 a, const bf16x8 b)
{
    // s_nop: the operands come from VALU instructions the hazard recogniser cannot see through the asm
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_" MLP_SFX " %0, %1, %2, %0" : "+a"(p) : "v"(a), "v"(b));
}
#endif

// Pins a fragment to the accumulation registers (a value defined by an "a"-constrained asm can only be allocated
// there); MFMA A/B operands and ds_read destinations may be AGPRs, so this costs no instruction -- it only keeps
// the weight fragments from competing with the activations for the 256 VGPRs.
__device__ __forceinline__ bf16x8 in_agpr(bf16x8 f)
{
    asm("" : "+a"(f));
    return f;
}

struct ZoneRow {    // one zone row as loaded: floats 0-3, 4-5, and its last two
    f32x4u a;
    f32x2u b, c;
};
template <int F>
__device__ __forceinline__ ZoneRow load_zone_row(const float *__restrict__ rows, int t)
{
    const float *p = rows + (size_t)t * F;
    ZoneRow x;
    x.a = *reinterpret_cast<const f32x4u *>(p);
    x.b = *reinterpret_cast<const f32x2u *>(p + 4);
    x.c = *reinterpret_cast<const f32x2u *>(p + F - 2);
    return x;
}
// layer 1's B operand: lane half 0 the env's obs (k = 0..7, constant over the group's tiles), half 1 the
// zone row (k = 8..14) and the bias slot (k = 15)
// (branch-free: `keep` is all ones in the lanes that carry a zone row -- half 1 of a valid env -- and `obs_sel` the obs
// fragment in half 0, zero elsewhere: one v_and_or_b32 per dword instead of two exec-masked blocks at every tile's head)
__device__ __forceinline__ uint32_t and_or(uint32_t a, uint32_t m, uint32_t o) { return (a & m) | o; }
template <int F>
__device__ __forceinline__ bf16x8 zone_frag(const ZoneRow &x, const uint4 obs_sel, uint32_t keep)
{
    uint4 u;
    u.x = and_or(pk_bf16(x.a.x, x.a.y), keep, obs_sel.x);
    u.y = and_or(pk_bf16(x.a.z, x.a.w), keep, obs_sel.y);
    u.z = and_or(pk_bf16(x.b.x, x.b.y), keep, obs_sel.z);
    u.w = and_or(pk_bf16(F == 7 ? x.c.y : 0.f, 1.0f), keep, obs_sel.w);
    return as_frag(u);
}

constexpr int kZone1Waves = 4;
constexpr int kMlpSplitMaxEnvs = 8192;    // at most this many envs: zone tiles of a 32-env group split over the workgroup's waves
#ifndef MLP_TIE
#define MLP_TIE 1          // fragment pins tied into the accumulation chain (see zone_tile); 0: the round-2 kernel
#endif
#ifndef MLP_GAPS
#define MLP_GAPS 6          // MFMA gaps that carry VALU work in a region ...
#define MLP_PER_GAP 6       // ... and instructions per gap (32 per region: 16 + 16 conversions)
#endif
#ifndef MLP_V_PER_GAP
#define MLP_V_PER_GAP 4     // MLP_POOL_VALU: vector instructions per MFMA gap (48 per region over 12 gaps)
#endif
#ifndef MLP_RGAPS
#define MLP_RGAPS 6         // MFMA gaps behind those that carry the next chain's fragment reads ...
#define MLP_R_PER_GAP 2     // ... and reads per gap (12 per region)
#endif

// One 32-row tile: layer-2 chains on xa, pooling into pool[], and layer 1 of the NEXT tile (operand x0n) into
// xb; x0nn is the operand of the tile after that.  pend = the previous tile's last chain, still to be converted and pooled; wf0 = fragments of output
// tile 0, read by the previous tile.
__device__ __forceinline__ void zone_tile(const uint4 *__restrict__ w2s, int lane, const bf16x8 (&w1f)[NT],
                                          const bf16x8 (&ind)[2], const int ind_idx, const bf16x8 (&xa)[KS],
                                          bf16x8 (&xb)[KS],
                                          const bf16x8 x0n, const bf16x8 x0nn, f32x16 &a1, f32x16 &pend,
                                          f32x16 (&pool)[NT], bf16x8 (&wf0)[KS]
#ifdef MLP_STAMP
                                          , int wave, int stamp_it
#endif
                                          )
{
    bf16x8 wf[2][KS];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) wf[0][kk] = wf0[kk];
    f32x16 acc[2];
    acc[1] = pend;
    // layer 1 of the next tile (operand x0n) runs one region ahead of its conversion -- an MFMA result is 64
    // cycles away; a1 arrives holding its output tile 0, issued by the previous call, and leaves holding output
    // tile 0 of the tile after next (operand x0nn), so that the tile boundary has matrix work in flight
    KSTAMP(0);
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int nn = (n + 1) % NT;
        // the next chain's fragments (placed in the region's late MFMA gaps by the group barriers below)
#pragma unroll
        for (int kk = 0; kk < KS; ++kk)
#if defined(MLP_EXP) && (MLP_EXP & 4)      // diagnostic: one fragment read per chain instead of twelve
            wf[(n + 1) & 1][kk] = kk == 0 ? as_frag(w2s[(nn * KS + kk) * kWave + lane]) : wf[n & 1][kk];
#else
            wf[(n + 1) & 1][kk] = as_frag(w2s[(nn * KS + kk) * kWave + lane]);
#endif

        const f32x16 prev = acc[(n + 1) & 1];                   // the chain before this one (n = 0: pend)
        acc[n & 1] = zero16();
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
#if MLP_TIE
            // the fragment's AGPR pin is tied to the running accumulator, so it sits between the chain's MFMAs kk - 1 and
            // kk and the wait for its ds_read with it: lgkmcnt(11 - kk) in front of MFMA kk.  Untied, all twelve pins
            // (and a lgkmcnt(0)) were hoisted to the head of the region -- the chain's first MFMA waited for the reads
            // issued three MFMAs earlier, with nothing but the two pooling products in the matrix pipe.
            bf16x8 b = wf[n & 1][kk];
            if (kk == 0) asm("" : "+a"(b));
            else asm("" : "+a"(b), "+v"(acc[n & 1]));
            acc[n & 1] = mfma(xa[kk], b, acc[n & 1]);
#else
            acc[n & 1] = mfma(xa[kk], in_agpr(wf[n & 1][kk]), acc[n & 1]);
#endif
        }
#if MLP_POOL_VALU
#if !defined(MLP_EXP) || !(MLP_EXP & 8)    // diagnostic (bit 3): no pooling
        pool_add(pool[(n + NT - 1) % NT], prev);
#else
        asm volatile("" ::"v"(prev));
#endif
        {
            // layer 1's fragments go to the accumulation registers as they are made (v_accvgpr_write: the MFMA's A
            // operand may live there), which is what leaves the 96 float32 sums room among the VGPRs
            bf16x8 t0, t1;
            acc_to_frags(a1, true, t0, t1);
            xb[2 * n] = in_agpr(t0);
            xb[2 * n + 1] = in_agpr(t1);
        }
#else
        {
            bf16x8 f0, f1;
#if defined(MLP_EXP) && (MLP_EXP & 1)      // diagnostic: no conversion of the previous chain
            f0 = xa[0]; f1 = xa[1];
            asm volatile("" ::"v"(prev));
#else
            acc_to_frags(prev, true, f0, f1);
#endif
#if !defined(MLP_EXP) || !(MLP_EXP & 8)    // diagnostic (bit 3): no pooling products
#if MLP_POOL_SPARSE
            pool_smfmac(pool[(n + NT - 1) % NT], ind[0], f0, f1, ind_idx);
#else
            pool_mfma(pool[(n + NT - 1) % NT], ind[0], f0);
            pool_mfma(pool[(n + NT - 1) % NT], ind[1], f1);
#endif
#else
            asm volatile("" ::"v"(f0), "v"(f1));
#endif
        }
#if defined(MLP_EXP) && (MLP_EXP & 2)      // diagnostic: no conversion of layer 1
        xb[2 * n] = xa[2 * n]; xb[2 * n + 1] = xa[2 * n + 1];
        asm volatile("" ::"v"(a1));
#else
        acc_to_frags(a1, true, xb[2 * n], xb[2 * n + 1]);
#endif
        // (materialised here: otherwise the conversions sink into the next tile's block, behind its branch)
        asm volatile("" : "+v"(xb[2 * n]), "+v"(xb[2 * n + 1]));
#endif
        a1 = n + 1 < NT ? mfma(w1f[n + 1], x0n, zero16())       // layer 1, next tile, output tile n + 1
                        : mfma(w1f[0], x0nn, zero16());         // ... and output tile 0 of the tile after it
#if MLP_POOL_VALU && (!defined(MLP_EXP) || !(MLP_EXP & 32))
        // order of the region: 13 MFMAs (12 of the chain + layer 1's), 48 vector instructions (16 + 8 for layer 1's
        // fragments, 24 for the sums) and the next chain's 12 fragment reads: one MFMA ahead (the first vector
        // instruction waits for the previous chain's last result), then 4 + 1 per gap
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
#pragma unroll
        for (int gsl = 0; gsl < 12; ++gsl) {
            __builtin_amdgcn_sched_group_barrier(0x002, MLP_V_PER_GAP, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
#elif !defined(MLP_EXP) || !(MLP_EXP & 32)
        // order of the region: two chain MFMAs ahead of the first conversion (which waits for the previous chain's
        // last result -- the new chain must not queue behind it), then six VALU instructions per MFMA gap
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
#pragma unroll
        for (int gsl = 0; gsl < MLP_GAPS; ++gsl) {
            __builtin_amdgcn_sched_group_barrier(0x002, MLP_PER_GAP, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
        // ... and the next chain's twelve fragment reads two per gap behind them (as a block at the head of the
        // region they cost 48 issue cycles with no MFMA going out: -2.3 us per forward)
#pragma unroll
        for (int gsl = 0; gsl < MLP_RGAPS; ++gsl) {
            __builtin_amdgcn_sched_group_barrier(0x100, MLP_R_PER_GAP, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
#endif
        __builtin_amdgcn_sched_barrier(0);
        KSTAMP(1 + n);
    }
    pend = acc[(NT - 1) & 1];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) wf0[kk] = in_agpr(wf[NT & 1][kk]);   // (read a whole chain ago: no wait)
}

template <int ZT, int F, bool SPLIT>
__global__ __launch_bounds__(kZone1Waves * kWave) __attribute__((amdgpu_waves_per_eu(1, 1)))
void k_mlp_zone1(MlpImages img, int N, int Z_rt, const float *__restrict__ obs, const float *__restrict__ zone_obs,
                 elem_t *__restrict__ pooled, int envs_per_wave)
{
    // Batch layout (launch_mlp_forward picks it by N so that a small batch still spreads over the chip):
    //   !SPLIT: a wave owns envs_per_wave (64 or 32) envs and all their zone tiles;
    //   SPLIT:  the workgroup's four waves share ONE group of 32 envs, wave w takes zones w, w + 4, ...; the partial
    //           sums of waves 1-3 meet wave 0's in LDS (behind the weight images).
    extern __shared__ uint4 lds[];
    uint4 *w2s = lds;                       // [NT*KS][64]
    uint4 *w1s = lds + NT * KS * kWave;     // [NT][64]
    {
        // all of a thread's 18 + 2 loads in flight at once: as a load -> wait -> ds_write loop the staging was 18
        // dependent L2 round trips (~14 us of a 115 us kernel with no wave doing anything else)
        constexpr int kThreads = kZone1Waves * kWave, kPer = NT * KS * kWave / kThreads;
        static_assert(NT * KS * kWave % kThreads == 0, "W2 image divides evenly over the workgroup");
        uint4 t2[kPer], t1[2];
#pragma unroll
        for (int j = 0; j < kPer; ++j) t2[j] = reinterpret_cast<const uint4 *>(img.w2)[threadIdx.x + kThreads * j];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = threadIdx.x + kThreads * j;
            t1[j] = i < NT * kWave ? reinterpret_cast<const uint4 *>(img.w1)[i] : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int j = 0; j < kPer; ++j) w2s[threadIdx.x + kThreads * j] = t2[j];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = threadIdx.x + kThreads * j;
            if (i < NT * kWave) w1s[i] = t1[j];
        }
    }
    __syncthreads();

    const int Z = ZT > 0 ? ZT : Z_rt;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // wave-uniform: the tile loop's bounds are scalar
    const int r = lane & 31, h = lane >> 5;
    constexpr bool split = SPLIT;
    const int env0 = split ? (int)blockIdx.x * 32 : (int)(blockIdx.x * kZone1Waves + wave) * envs_per_wave;
    if (env0 >= N) return;                                    // (split: the whole workgroup, so nobody misses the barrier)
    const int n_env = min(split ? 32 : envs_per_wave, N - env0);
    const float inv_z = 1.0f / (float)Z;
    // this wave's tiles: zones z_first, z_first + z_step, ... (n_tiles of them)
    const int z_first = split ? wave : 0, z_step = split ? kZone1Waves : 1;
    const int n_tiles = split ? (Z - wave + kZone1Waves - 1) / kZone1Waves : Z;     // !SPLIT, ZT > 0: a constant
    float *red = reinterpret_cast<float *>(lds + (NT * KS + NT) * kWave);   // [3][NT * 16][64] floats, split only
#ifdef MLP_STAMP
    int stamp_it = 0;
#endif
    bf16x8 w1f[NT];
#pragma unroll
    for (int m = 0; m < NT; ++m) w1f[m] = in_agpr(as_frag(w1s[m * kWave + lane]));
    // the identity as the pooling product's A operand: lane (slot r, half h) element j of k-step s is tile row
    // 16 s + 8 (j >> 2) + 4 h + (j & 3) -- the accumulator order of H2^T
    bf16x8 ind[2];
    int ind_idx = 0;
#if MLP_POOL_SPARSE
    {
        // the same identity in 2:4 form: tile row r sits at k = 16 (r >> 4) + 8 ((r >> 2) & 1) + 4 ((r >> 3) & 1) + (r & 3),
        // i.e. lane half r >> 4, group g, position r & 3 (slot 2 g carries the 1, its partner a 0 at another position)
        const int g = 2 * ((r >> 2) & 1) + ((r >> 3) & 1), m = r & 3;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            ind[0][i] = (h == (r >> 4) && i == 2 * g) ? (elem_t)1.0f : (elem_t)0.0f;
            const int v = i == 2 * g ? m : i == 2 * g + 1 ? ((m + 1) & 3) : (i & 1);
            ind_idx |= v << (2 * i);
        }
        ind[1] = ind[0];
    }
#else
#pragma unroll
    for (int sgm = 0; sgm < 2; ++sgm)
#pragma unroll
        for (int j = 0; j < 8; ++j)
            ind[sgm][j] = (16 * sgm + 8 * (j >> 2) + 4 * h + (j & 3) == r) ? (elem_t)1.0f : (elem_t)0.0f;
#endif

    KSTAMP_WHOLE(0);
    for (int e_base = 0; e_base < n_env; e_base += 32) {
        const bool valid = e_base + r < n_env;
        const int env = env0 + e_base + (valid ? r : 0);
        const float *rows = zone_obs + (size_t)env * Z * F;
        uint4 obs_frag;
        {
            const float4 *o = reinterpret_cast<const float4 *>(obs + (size_t)env * 8);
            const float4 a = o[0], b = o[1];
            obs_frag = make_uint4(pk_bf16(a.x, a.y), pk_bf16(a.z, a.w), pk_bf16(b.x, b.y), pk_bf16(b.z, b.w));
            if (!valid) obs_frag = make_uint4(0u, 0u, 0u, 0u);
#if MLP_ELEM_F16
            // float16: zenv_mlp_load has bounded the zone layers' activations for observations up to kMlpF16ObsBound (the
            // zone rows are bounded by construction); an env beyond it -- a robot driven 190 m out of the arena -- voids
            // that bound and is reported like an overflow
            const float om = fmaxf(fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))),
                                   fmaxf(fmaxf(fabsf(b.x), fabsf(b.y)), fmaxf(fabsf(b.z), fabsf(b.w))));
            if (valid && !(om <= kMlpF16ObsBound) && img.range_flag) *img.range_flag = 1;
#endif
        }
        const uint32_t keep = (valid && h != 0) ? 0xFFFFFFFFu : 0u;
        const uint4 obs_sel = h == 0 ? obs_frag : make_uint4(0u, 0u, 0u, 0u);
        f32x16 pool[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            pool[n] = zero16();
#if !MLP_POOL_VALU
            asm volatile("" : "+a"(pool[n]));      // AGPR-resident from here on (see pool_mfma)
#endif
        }
        f32x16 pend = zero16();
        bf16x8 wf0[KS];
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) wf0[kk] = in_agpr(as_frag(w2s[kk * kWave + lane]));
        const auto zone_of = [&](int tile) { return z_first + z_step * min(tile, max(n_tiles - 1, 0)); };
        // prologue: layer 1 of tile 0 (not overlapped: once per Z tiles)
        ZoneRow nxt = load_zone_row<F>(rows, min(zone_of(0), Z - 1));
        bf16x8 xa[KS], xb[KS];
        bf16x8 x0n;
        f32x16 a1;
        {
            const bf16x8 x0 = zone_frag<F>(nxt, obs_sel, keep);
            nxt = load_zone_row<F>(rows, min(zone_of(1), Z - 1));
#pragma unroll
            for (int m = 0; m < NT; ++m) {
                const f32x16 acc1 = mfma(w1f[m], x0, zero16());
                acc_to_frags(acc1, true, xa[2 * m], xa[2 * m + 1]);
#if MLP_POOL_VALU
                xa[2 * m] = in_agpr(xa[2 * m]);
                xa[2 * m + 1] = in_agpr(xa[2 * m + 1]);
#endif
            }
            x0n = zone_frag<F>(nxt, obs_sel, keep);                           // tile 1
            nxt = load_zone_row<F>(rows, min(zone_of(2), Z - 1));
            a1 = mfma(w1f[0], x0n, zero16());
        }
        for (int t = 0; t < n_tiles; t += 2) {
            {
                const bf16x8 x0nn = zone_frag<F>(nxt, obs_sel, keep);         // tile t + 2
                __builtin_amdgcn_sched_barrier(0);
#if !defined(MLP_EXP) || !(MLP_EXP & 64)   // diagnostic (bit 6): no zone-row loads inside the tile loop
                nxt = load_zone_row<F>(rows, zone_of(t + 3));
#endif
                zone_tile(w2s, lane, w1f, ind, ind_idx, xa, xb, x0n, x0nn, a1, pend, pool, wf0 ZT_STAMP_ARGS);
                x0n = x0nn;
            }
            if (t + 1 < n_tiles) {
                const bf16x8 x0nn = zone_frag<F>(nxt, obs_sel, keep);         // tile t + 3
                __builtin_amdgcn_sched_barrier(0);
#if !defined(MLP_EXP) || !(MLP_EXP & 64)
                nxt = load_zone_row<F>(rows, zone_of(t + 4));
#endif
                zone_tile(w2s, lane, w1f, ind, ind_idx, xb, xa, x0n, x0nn, a1, pend, pool, wf0 ZT_STAMP_ARGS);
                x0n = x0nn;
            }
        }
        // the last chain of the last tile (a wave without tiles -- split, Z < 4 -- has nothing pending)
#if MLP_POOL_VALU
        if (n_tiles > 0) pool_add(pool[NT - 1], pend);
#else
        if (n_tiles > 0) {
            bf16x8 f0, f1;
            acc_to_frags(pend, true, f0, f1);
#if MLP_POOL_SPARSE
            pool_smfmac(pool[NT - 1], ind[0], f0, f1, ind_idx);
#else
            pool_mfma(pool[NT - 1], ind[0], f0);
            pool_mfma(pool[NT - 1], ind[1], f1);
#endif
        }
        asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");     // asm MFMA results -> v_accvgpr_read
#endif
        if (split) {
            // waves 1-3 leave their partial sums in LDS ([register][lane]: conflict-free), wave 0 adds them in wave order
            if (wave > 0) {
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int i = 0; i < 16; ++i) red[((wave - 1) * NT * 16 + n * 16 + i) * kWave + lane] = pool[n][i];
            }
            __syncthreads();
            if (wave > 0) return;
#pragma unroll
            for (int w = 0; w < kZone1Waves - 1; ++w)
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int i = 0; i < 16; ++i) pool[n][i] += red[(w * NT * 16 + n * 16 + i) * kWave + lane];
        }
        // ---- the group's means: accumulator register i of lane half h is env slot (i&3) + 8 (i>>2) + 4 h
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int e = e_base + (i & 3) + 8 * (i >> 2) + 4 * h;
                const float v = pool[n][i] * inv_z;
                const float vn = __shfl_down(v, 1);
                if (e < n_env && !(r & 1))
                    *reinterpret_cast<uint32_t *>(pooled + (size_t)(env0 + e) * HP + 32 * n + r) = pk_bf16(v, vn);
            }
    }
    KSTAMP_WHOLE(1);
}

// ------------------------------------------------------------------------------------------ kernel 2
// (Fetching the next layer's fragments into registers during the current layer's products and committing them
// to a second LDS buffer afterwards was slower than plain staging: 48 us against 34 us.)
// 8 waves per workgroup (two per SIMD), 32 envs per wave.  The layers' fragment images are staged into two LDS
// buffers by LDS-DMA (global_load_lds_dwordx4: no VGPR destination), the next layer's image in flight while the
// current layer is in the matrix pipe; a raw s_barrier + counted waits, because __syncthreads() would drain the
// DMA at once (cdna_hip_programming.md, "Pipelining across barriers").
constexpr int kHeadWaves = 8;
constexpr int kImgFrags = NT * (KS + 1);      // the largest image (combine_net_), in 1 KiB fragments

__device__ __forceinline__ void stage_issue(uint4 *dst, const void *src, int n_frags)
{
    const int lane = threadIdx.x & (kWave - 1);
    for (int i = threadIdx.x; i < n_frags * kWave; i += kHeadWaves * kWave)
        // the LDS destination is wave-uniform base + lane * 16; the global source is per lane
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(reinterpret_cast<const uint4 *>(src) + i),
                                         (__attribute__((address_space(3))) void *)(dst + (i - lane)), 16, 0, 0);
}
// every DMA this wave issued has landed, and every wave of the workgroup is here: the staged buffer may be read
// and the other buffer may be overwritten
__device__ __forceinline__ void stage_fence()
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// One hidden layer of the head: xout = [relu] (W xin [+ W_obs xo]), W's fragments staged in `buf` (NK k-steps per
// output tile, the last one the obs part when NK > KS).  The conversion of output tile m-1 is issued between the
// MFMAs of tile m (it used to be one block of 96 VALU instructions behind all six chains, with the matrix pipe idle
// and all six accumulators live); xin and xout are different registers.
template <int NK, bool RELU>
__device__ __forceinline__ void head_layer(const uint4 *buf, int lane, const bf16x8 (&xin)[KS], const bf16x8 xo,
                                           bf16x8 (&xout)[KS], float &mx)
{
    f32x16 acc[2];
#pragma unroll
    for (int m = 0; m <= NT; ++m) {
        if (m < NT) {
            acc[m & 1] = zero16();
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) acc[m & 1] = mfma(as_frag(buf[(m * NK + kk) * kWave + lane]), xin[kk], acc[m & 1]);
            if (NK > KS) acc[m & 1] = mfma(as_frag(buf[(m * NK + KS) * kWave + lane]), xo, acc[m & 1]);
        }
        if (m > 0) {
#if MLP_ELEM_F16
            // float16: the largest value about to be rounded (behind a ReLU only the positive side matters)
#pragma unroll
            for (int i = 0; i < 16; ++i) mx = fmaxf(mx, RELU ? acc[(m - 1) & 1][i] : fabsf(acc[(m - 1) & 1][i]));
#endif
            acc_to_frags(acc[(m - 1) & 1], RELU, xout[2 * (m - 1)], xout[2 * (m - 1) + 1]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

__global__ __launch_bounds__(kHeadWaves * kWave) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_mlp_head(MlpImages img, int N, const float *__restrict__ obs, const elem_t *__restrict__ pooled,
                float *__restrict__ mu, float *__restrict__ stdv, float *__restrict__ value,
                float *__restrict__ value_sigma, MlpAction act)
{
    extern __shared__ uint4 wl2[];          // two buffers of kImgFrags KiB
    uint4 *const bufA = wl2, *const bufB = wl2 + kImgFrags * kWave;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    // every wave of the workgroup takes part in the staging, also one without envs
    const int first = (int)(blockIdx.x * kHeadWaves + wave) * 32;
    const int env = min(first + r, max(N - 1, 0));
    const bool valid = first + r < N;
    const bool critic = img.wv1 != nullptr;
    float value_out = 0.f;
    float mx = 0.f;                         // float16 build: see head_layer

    stage_issue(bufB, img.wc, NT * (KS + 1));
    bf16x8 xa[KS], xb[KS], xo;              // the activations ping-pong between xa and xb
    {
        // the mean is stored in bf16, 8 consecutive features = one fragment
        const uint4 *pr = reinterpret_cast<const uint4 *>(pooled + (size_t)env * HP + 8 * h);
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) xa[kk] = as_frag(pr[2 * kk]);
        const float4 *o = reinterpret_cast<const float4 *>(obs + (size_t)env * 8);
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        xo = h == 0 ? frag_from_floats(o[0], o[1]) : frag_from_floats(z4, z4);
    }
    // ---- c = Wc [W3 mean(H2) + b3; obs]: zone_net_.4 (after the mean) and combine_net_ have no activation between
    // them, so the host folds them into ONE layer (pack_images: Wc' = Wc_emb W3 in float64)              xa -> xb
    stage_fence();
    stage_issue(bufA, critic ? img.wv1 : img.wa, NT * KS);
    head_layer<KS + 1, false>(bufB, lane, xa, xo, xb, mx);
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) xa[kk] = xb[kk];          // (register renaming: the layers below read c from xa)
    // ---- value = Wv2 relu(Wv1 c)   (critic, flat_model.py:43-47), from the same embedding c     xa -> xb
    if (critic) {
        stage_fence();
        stage_issue(bufB, img.wv2, KS);
        head_layer<KS, true>(bufA, lane, xa, xo, xb, mx);
        stage_fence();
        stage_issue(bufA, img.wa, NT * KS);
        f32x16 hv = zero16();
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) hv = mfma(as_frag(bufB[kk * kWave + lane]), xb[kk], hv);
        value_out = hv[0];                                      // critic.2, or critic_mu (distributional)
        if (h == 0 && valid) {
            value[env] = value_out;
            if (value_sigma) value_sigma[env] = softplus03(hv[1]) + 1e-3f;   // flat_model.py:57-60
        }
    }
    // ---- a = relu(Wa c)   (actor.enc_)                                                          xa -> xb
    stage_fence();
    stage_issue(bufB, img.wh, KS);
    head_layer<KS, true>(bufA, lane, xa, xo, xb, mx);
    // ---- heads: rows 0-1 = mu_, rows 2-3 = std_ (lane half 0, registers 0..3)
    stage_fence();
    f32x16 hd = zero16();
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) hd = mfma(as_frag(bufB[kk * kWave + lane]), xb[kk], hd);
    if (h == 0 && valid) head_outputs(env, hd[0], hd[1], hd[2], hd[3], value_out, mu, stdv, act);
#if MLP_ELEM_F16
    // an activation of 65 520 or more became inf on its way to float16 (NaN compares false): tell the host
    if (valid && !(mx < 65520.0f) && img.range_flag) *img.range_flag = 1;
#endif
}

#if !MLP_ELEM_F16
// ------------------------------------------------------------------------------------------ experiences
// collect_experiences (main/src/torch_ac/algos/base.py:131-216) on the device.  Every experience buffer is TIME-major
// [T][N][...] in memory -- the head kernel's epilogue (MlpRecord) and the GAE scan then touch whole lines, and the
// step kernel writes obs_{t+1} / zone_obs_{t+1} straight into slot t+1 (zenv_collect points it there), so recording
// the observations costs no copy.  The caller sees the transposed [N][T][...] view, whose reshape(N*T, ...) is the
// layout of exps.* in the reference ("k-th block of T consecutive frames = k-th env", :125-128).

// after the LAST env step of a call: rewards[T-1] (shaped_reward when the env provides it, :153-159) and the new
// self.mask (the earlier frames' rewards are recorded by the next frame's head kernel)
__global__ __launch_bounds__(256) void k_exp_reward(ExpBuffers x, int N, int t, const float *__restrict__ reward,
                                                    const double *__restrict__ shaped, const uint8_t *__restrict__ done)
{
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= N) return;
    x.reward[(size_t)t * N + env] = shaped ? (float)shaped[env] : reward[env];
    x.cur_mask[env] = done[env] ? 0.f : 1.f;            // self.mask = 1 - done (:150)
}

// advantages[i] = delta + discount * gae_lambda * advantages[i+1] * masks[i+1] (:190-196); returnn = value + advantage
__global__ __launch_bounds__(256) void k_exp_gae(ExpBuffers x, int N, const float *__restrict__ next_value, float discount,
                                                 float gae_lambda)
{
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= N) return;
    float nv = next_value[env], nm = x.cur_mask[env], na = 0.f;
    for (int i = x.T - 1; i >= 0; --i) {
        const size_t at = (size_t)i * N + env;              // time-major: the threads of a wave read one line
        const float v = x.value[at];
        const float delta = x.reward[at] + discount * nv * nm - v;
        const float adv = delta + discount * gae_lambda * na * nm;
        x.advantage[at] = adv;
        x.returnn[at] = v + adv;
        nv = v;
        nm = x.mask[at];
        na = adv;
    }
}

// float -> float16 bits, round to nearest even (overflow -> inf: zenv_mlp_load refuses such weights before it packs)
uint16_t to_f16(float f)
{
    uint32_t u;
    std::memcpy(&u, &f, 4);
    const uint32_t sign = (u >> 16) & 0x8000u, ex = (u >> 23) & 0xFFu, man = u & 0x007FFFFFu;
    if (ex == 0xFFu) return (uint16_t)(sign | 0x7C00u | (man ? 0x200u : 0u));
    const int e = (int)ex - 127 + 15;
    if (e >= 31) return (uint16_t)(sign | 0x7C00u);
    if (e <= 0) {                                        // subnormal half (or zero)
        if (e < -10) return (uint16_t)sign;
        const uint32_t m = (man | 0x00800000u), shift = (uint32_t)(14 - e);
        uint32_t half = m >> shift;
        const uint32_t rem = m & ((1u << shift) - 1u), mid = 1u << (shift - 1);
        if (rem > mid || (rem == mid && (half & 1u))) ++half;
        return (uint16_t)(sign | half);
    }
    uint32_t half = ((uint32_t)e << 10) | (man >> 13);
    const uint32_t rem = man & 0x1FFFu;
    if (rem > 0x1000u || (rem == 0x1000u && (half & 1u))) ++half;      // may carry into the exponent: still correct
    return (uint16_t)(sign | half);
}
uint16_t to_bf16(float f)
{
    uint32_t u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7F800000u) == 0x7F800000u && (u & 0x007FFFFFu)) return (uint16_t)((u >> 16) | 0x40);   // NaN
    u += 0x7FFFu + ((u >> 16) & 1u);      // round to nearest even
    return (uint16_t)(u >> 16);
}

// logical k of element j of lane half h in k-step kk
int k_natural(int kk, int h, int j) { return 16 * kk + 8 * h + j; }
int k_from_acc(int kk, int h, int j) { return 32 * (kk >> 1) + 16 * (kk & 1) + 8 * (j >> 2) + 4 * h + (j & 3); }
#endif   // !MLP_ELEM_F16 (the experience kernels and the packer exist once, in the bf16 build of this file)

}  // namespace

#if !MLP_ELEM_F16

// One image = [n_tiles][n_ksteps] fragments of 64 lanes x 8 bf16.  Lane (r, h) of fragment (n, kk) holds
// value(out = 32 n + r, k = order(kk, h, j)), j = 0..7.  `value` resolves a (row, logical k) pair.
static thread_local bool g_pack_f16 = false;      // element type of the images pack_images is packing on this thread
template <typename ValueFn, typename OrderFn>
static void pack_image(std::vector<uint16_t> &out, int n_tiles, int n_ksteps, ValueFn value, OrderFn order)
{
    for (int n = 0; n < n_tiles; ++n)
        for (int kk = 0; kk < n_ksteps; ++kk)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j)
                {
                    const float v = value(32 * n + (lane & 31), kk, order(kk, lane >> 5, j));
                    out.push_back(g_pack_f16 ? to_f16(v) : to_bf16(v));
                }
}

int pack_images(const zenv_mlp_weights &w, int F, std::vector<uint16_t> &out, size_t offs[8], bool f16)
{
    const int h = w.h_dim;
    if (h < 1 || h + 1 > kMlpHP || 8 + F > 15) return -1;
    out.clear();
    g_pack_f16 = f16;
    // a hidden layer [h][h] whose input carries the constant 1 in feature h: bias in column h, row h keeps the 1
    auto hidden = [h](const float *W, const float *b) {
        return [=](int row, int, int k) -> float {
            if (row == h) return k == h ? 1.f : 0.f;
            if (row > h || k > h) return 0.f;
            return k == h ? b[row] : W[(size_t)row * h + k];
        };
    };
    // zone_net_.0: input k = [obs 0..7, zone row 0..F-1, 0.., bias slot 15]
    offs[0] = out.size() * 2;
    pack_image(out, kMlpNT, 1, [&](int row, int, int k) -> float {
        if (row == h) return k == 15 ? 1.f : 0.f;
        if (row > h) return 0.f;
        if (k == 15) return w.zone_b1[row];
        return k < 8 + F ? w.zone_w1[(size_t)row * (8 + F) + k] : 0.f;
    }, k_natural);
    offs[1] = out.size() * 2;
    pack_image(out, kMlpNT, kMlpKS, hidden(w.zone_w2, w.zone_b2), k_from_acc);
    // zone_net_.4 (applied to the zone mean: no activation) folded into combine_net_ (no activation either):
    //   c = Wc [obs; W3 m + b3] + bc = Wc_obs obs + (Wc_emb W3) m + (Wc_emb b3 + bc),
    // products in float64, rounded to bf16 once -- one layer and one bf16 rounding of the activations fewer.
    // torch input order of combine_net_: [obs (8), zone_emb (h)]; here k-steps 0..KS-1 = the mean as it is read from
    // memory (natural order, + the constant), k-step KS = obs in natural order.
    std::vector<double> wf((size_t)h * (h + 1));
    for (int i = 0; i < h; ++i) {
        const float *ce = w.comb_w + (size_t)i * (8 + h) + 8;
        for (int j = 0; j <= h; ++j) {
            double a = j == h ? (double)w.comb_b[i] : 0.0;
            for (int k = 0; k < h; ++k) a += (double)ce[k] * (j == h ? (double)w.zone_b3[k] : (double)w.zone_w3[(size_t)k * h + j]);
            wf[(size_t)i * (h + 1) + j] = a;
        }
    }
    offs[2] = offs[3] = out.size() * 2;
    pack_image(out, kMlpNT, kMlpKS + 1, [&](int row, int kk, int k) -> float {
        if (kk == kMlpKS) {
            const int ko = k - 16 * kMlpKS;
            return (row < h && ko < 8) ? w.comb_w[(size_t)row * (8 + h) + ko] : 0.f;
        }
        if (row == h) return k == h ? 1.f : 0.f;
        if (row > h || k > h) return 0.f;
        return (float)wf[(size_t)row * (h + 1) + k];
    }, k_natural);
    offs[4] = out.size() * 2;
    pack_image(out, kMlpNT, kMlpKS, hidden(w.enc_w, w.enc_b), k_from_acc);
    offs[5] = out.size() * 2;
    pack_image(out, 1, kMlpKS, [&](int row, int, int k) -> float {
        if (row > 3 || k > h) return 0.f;
        const float *W = row < 2 ? w.mu_w : w.std_w;
        const float *b = row < 2 ? w.mu_b : w.std_b;
        const int rr = row & 1;
        return k == h ? b[rr] : W[(size_t)rr * h + k];
    }, k_from_acc);
    offs[6] = offs[7] = 0;
    if (w.critic_w1) {
        offs[6] = out.size() * 2;
        pack_image(out, kMlpNT, kMlpKS, hidden(w.critic_w1, w.critic_b1), k_from_acc);
        offs[7] = out.size() * 2;
        // row 0: critic.2 (or critic_mu), row 1: critic_sigma of the distributional critic (flat_model.py:35-41)
        pack_image(out, 1, kMlpKS, [&](int row, int, int k) -> float {
            if (row > 1 || k > h || (row == 1 && !w.critic_sigma_w)) return 0.f;
            if (row == 1) return k == h ? w.critic_sigma_b[0] : w.critic_sigma_w[k];
            return k == h ? w.critic_b2[0] : w.critic_w2[k];
        }, k_from_acc);
    }
    return 0;
}
#endif   // !MLP_ELEM_F16

#if MLP_ELEM_F16
hipError_t launch_mlp_forward_f16(const MlpImages &img, int N, int Z, int F, const float *obs, const float *zone_obs,
                                  void *pooled_v, float *mu, float *stdv, float *value, float *value_sigma,
                                  const MlpAction &act, hipStream_t s)
{
#else
hipError_t launch_mlp_forward(const MlpImages &img, int N, int Z, int F, const float *obs, const float *zone_obs,
                              void *pooled_v, float *mu, float *stdv, float *value, float *value_sigma,
                              const MlpAction &act, hipStream_t s)
{
    if (img.f32) return launch_mlp_forward_f32(*img.f32, N, Z, F, obs, zone_obs, mu, stdv, value, value_sigma, act, s);
    if (img.elem_f16)       // the same kernels compiled for float16 operands (mlp_policy_f16.hip)
        return launch_mlp_forward_f16(img, N, Z, F, obs, zone_obs, pooled_v, mu, stdv, value, value_sigma, act, s);
#endif
    elem_t *pooled = static_cast<elem_t *>(pooled_v);
    // Batch layout of the zone kernel.  A wave works through its tiles one after the other (~1.1 us each), so a small
    // batch laid out like a full one leaves the chip idle for the same 100 us: below kMlpSplitMaxEnvs a workgroup's four
    // waves share one group of 32 envs (zones w, w + 4, ... each; partial sums meet in LDS), up to 32 768 envs a wave
    // owns one group of 32, above that two.
    int epw = N <= 32768 ? 32 : kWave, zsplit = (N <= kMlpSplitMaxEnvs && Z >= kZone1Waves) ? kZone1Waves : 1;
    if (const char *force = std::getenv("ZENV_MLP_LAYOUT")) {      // diagnostic: "64", "32" or "split"
        zsplit = std::strcmp(force, "split") == 0 ? kZone1Waves : 1;
        epw = std::strcmp(force, "64") == 0 ? kWave : 32;
    }
#if MLP_KERNEL != 1
    epw = kWave, zsplit = 1;      // the predecessor kernel knows one layout
#endif
    const dim3 grid(zsplit > 1 ? (N + 31) / 32 : (N + kZone1Waves * epw - 1) / (kZone1Waves * epw));
    const size_t lds = (size_t)(NT * KS + NT) * kWave * sizeof(uint4) +
                       (zsplit > 1 ? (size_t)(kZone1Waves - 1) * NT * 16 * kWave * sizeof(float) : 0);
#if MLP_KERNEL == 1
#define ZENV_MLP_KERNEL k_mlp_zone1
#define ZENV_MLP_LAYOUT_ARGS , epw
    const dim3 block(kZone1Waves * kWave);
#else
#define ZENV_MLP_KERNEL k_mlp_zone
#define ZENV_MLP_LAYOUT_ARGS
    const dim3 block(kZoneWaves * kWave);
#endif
#if MLP_KERNEL == 1
#define ZENV_MLP_1(KERN)                                                                                          \
    do {                                                                                                          \
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&KERN), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                  (int)lds);                                                                      \
        hipLaunchKernelGGL(KERN, grid, block, lds, s, img, N, Z, obs, zone_obs, pooled ZENV_MLP_LAYOUT_ARGS);      \
    } while (0)
#define ZENV_MLP(ZT, FF)                                                                                          \
    do {                                                                                                          \
        if (zsplit > 1) ZENV_MLP_1((k_mlp_zone1<ZT, FF, true>));                                                  \
        else ZENV_MLP_1((k_mlp_zone1<ZT, FF, false>));                                                            \
    } while (0)
#else
#define ZENV_MLP(ZT, FF)                                                                                          \
    do {                                                                                                          \
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&ZENV_MLP_KERNEL<ZT, FF>),                       \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                          \
        hipLaunchKernelGGL((ZENV_MLP_KERNEL<ZT, FF>), grid, block, lds, s, img, N, Z, obs, zone_obs, pooled);      \
    } while (0)
#endif
    if (F == 6) {
        switch (Z) {
        case 25: ZENV_MLP(25, 6); break;
        case 15: ZENV_MLP(15, 6); break;
        default: ZENV_MLP(0, 6); break;
        }
    } else {
        switch (Z) {
        case 25: ZENV_MLP(25, 7); break;
        case 15: ZENV_MLP(15, 7); break;
        case 6: ZENV_MLP(6, 7); break;
        default: ZENV_MLP(0, 7); break;
        }
    }
#undef ZENV_MLP
#undef ZENV_MLP_1
#undef ZENV_MLP_KERNEL
#undef ZENV_MLP_LAYOUT_ARGS
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const size_t lds_head = 2 * (size_t)kImgFrags * kWave * sizeof(uint4);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_mlp_head), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds_head);
    hipLaunchKernelGGL(k_mlp_head, dim3((N + kHeadWaves * 32 - 1) / (kHeadWaves * 32)), dim3(kHeadWaves * kWave), lds_head, s,
                       img, N, obs, pooled, mu, stdv, value, img.distributional ? value_sigma : nullptr, act);
    return hipGetLastError();
}

#if !MLP_ELEM_F16
hipError_t launch_exp_reward(const ExpBuffers &x, int N, int t, const float *reward, const double *shaped,
                             const uint8_t *done, hipStream_t s)
{
    hipLaunchKernelGGL(k_exp_reward, dim3((N + 255) / 256), dim3(256), 0, s, x, N, t, reward, shaped, done);
    return hipGetLastError();
}

hipError_t launch_exp_gae(const ExpBuffers &x, int N, const float *next_value, float discount, float gae_lambda,
                          hipStream_t s)
{
    hipLaunchKernelGGL(k_exp_gae, dim3((N + 255) / 256), dim3(256), 0, s, x, N, next_value, discount, gae_lambda);
    return hipGetLastError();
}
#endif   // !MLP_ELEM_F16

}  // namespace zenvk

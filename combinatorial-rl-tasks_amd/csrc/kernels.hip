// kernels.hip -- gfx950 (CDNA4, wave64) kernels of the batched zone-env hot path.
//
// K1  k_step_lane<TASK>   one env.step() for every env (TSP_env.py:45-72, TTSP_env.py:62-71,
//                         colour_match_env.py:86-123, ZoneEnvBase.py:143-235 + the not-vendored
//                         Engine.step / mj_step of xmls/point.xml), with the ParallelEnv
//                         auto-reset of penv.py:8-11 fused in.
// K2  k_reset_lane<TASK>  masked re-init from the HBM layout bank (Engine.reset).
// K3  k_policy_lane<TASK> scripted action sources (the build's own).
//
// Execution shape (lane-per-env): a 64-thread workgroup = one wave64 owns 64 consecutive
// envs; lane i holds env i's dynamics in registers and streams the zone-major SoA zone
// arrays (512 B fully-coalesced loads).  The (N,Z,F) float32 zone_obs rows are built in an
// LDS tile [64][Z*F] and leave as one contiguous 64*Z*F*4-byte burst of dwordx4 stores.
// No MFMA: there is no contraction anywhere on this path; the bound is HBM bytes.
// Build with -ffp-contract=off: the float64 state must match the CPU oracle bit for bit.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include "det_math.hpp"
#include "dev_params.hpp"
#include "kernels.hpp"

namespace zenvk {

namespace {

constexpr int kWave = 64;

// ---------------------------------------------------------------------------- small helpers
__device__ __forceinline__ int hamming_to_goal(uint64_t colpack, int Z)
{
    // colour_match_env.py:38-55 with colours packed 2 bits/zone: 0 Blue, 1 Green, 2 Red
    const uint64_t lo_bits = 0x5555555555555555ull;
    const uint64_t used = (Z >= 32) ? ~0ull : ((1ull << (2 * Z)) - 1ull);
    const uint64_t g = colpack & lo_bits & used;          // bit0 set  -> Green
    const uint64_t r = (colpack >> 1) & lo_bits & used;   // bit1 set  -> Red
    const int ng = __popcll(g), nr = __popcll(r), nb = Z - ng - nr;
    const int to_blue = ng * 2 + nr, to_green = nr * 2 + nb, to_red = nb * 2 + ng;
    return min(to_blue, min(to_green, to_red));
}

// PCG64 (numpy default_rng) step/output on device, for the FixedSeedsWrapper schedule
__device__ __forceinline__ void pcg_step_dev(uint64_t &hi, uint64_t &lo, uint64_t ihi, uint64_t ilo)
{
    const uint64_t mh = 2549297995355413924ull, ml = 4865540595714422341ull;
    const uint64_t plo = lo * ml;
    const uint64_t phi = __umul64hi(lo, ml) + hi * ml + lo * mh;
    const uint64_t nlo = plo + ilo;
    const uint64_t nhi = phi + ihi + (nlo < plo ? 1ull : 0ull);
    hi = nhi;
    lo = nlo;
}

__device__ uint32_t pcg_next32_dev(const DevParams &p, int env)
{
    uint32_t *buf = p.pcg_buf + 2 * (size_t)env;
    if (buf[0]) {
        buf[0] = 0;
        return buf[1];
    }
    uint64_t *s = p.pcg + 4 * (size_t)env;
    uint64_t hi = s[0], lo = s[1];
    pcg_step_dev(hi, lo, s[2], s[3]);
    s[0] = hi;
    s[1] = lo;
    const uint64_t x = hi ^ lo;
    const unsigned rot = (unsigned)(hi >> 58);
    const uint64_t out = (x >> rot) | (x << ((0u - rot) & 63u));
    buf[0] = 1;
    buf[1] = (uint32_t)(out >> 32);
    return (uint32_t)out;
}

__device__ int next_bank_slot(const DevParams &p, int env)
{
    const int k = p.episode_idx[env];
    p.episode_idx[env] = k + 1;
    if (p.sched_mode == SCHED_SEQUENTIAL) {
        const long long s = (long long)p.slot_first[env] + (long long)k * (long long)p.sched_stride;
        return (int)(s % (long long)p.bank_size);
    }
    // wrappers.py:20-23: rng.integers(min_seed, max_seed + 1) -- Lemire on 32-bit draws
    const uint64_t rng = (uint64_t)(p.seed_max - p.seed_min);
    if (rng == 0) return 0;
    const uint32_t rng_excl = (uint32_t)rng + 1u;
    uint64_t m = (uint64_t)pcg_next32_dev(p, env) * rng_excl;
    uint32_t leftover = (uint32_t)m;
    if (leftover < rng_excl) {
        const uint32_t threshold = (0xFFFFFFFFu - (uint32_t)rng) % rng_excl;
        while (leftover < threshold) {
            m = (uint64_t)pcg_next32_dev(p, env) * rng_excl;
            leftover = (uint32_t)m;
        }
    }
    return (int)(m >> 32);
}


// a / b for a divisor known on the host, b = 3, 1.5 or a positive integer < 2^31, with
// inv_b = RN(1/b): q1 = RN(a*inv_b); r = a - b*q1 (exact in one fma); q = RN(q1 + r*inv_b).
// The result equals the IEEE quotient RN(a/b) bit for bit (argument in DESIGN.md: r is exact
// and the exact value of q1 + r*inv_b lies within 2^-53 ulp of a/b, which for these b is
// never that close to a rounding boundary), at 3 instructions instead of the ~14 of v_div_*.
// A zero numerator gives +0.0.
__device__ __forceinline__ double div_const(double a, double b, double inv_b)
{
    const double q1 = a * inv_b;
    const double r = __builtin_fma(-b, q1, a);
    return __builtin_fma(r, inv_b, q1);
}
// Same with the sign of a zero numerator preserved (-0.0 / b = -0.0 like a true division).
// Zone coordinates and step counts cannot be -0.0; velocities use this form.
__device__ __forceinline__ double div_const_z(double a, double b, double inv_b)
{
    const double q = div_const(a, b, inv_b);
    return a == 0.0 ? a : q;
}

// Per-lane registers of one env
struct EnvRegs {
    double q0, q1, q2, v0, v1, v2;
    double x0, y0, bq0, bq3;
    uint32_t vis;
    uint64_t colpack;
    int32_t goal_dist;
    int32_t steps;
};

// World-frame quantities of mj_forward for body "robot" (slides act in the placement frame)
struct Pose {
    double px, py, vx, vy, w, xq0, xq3;
};

__device__ __forceinline__ void world_pos(const EnvRegs &e, double &px, double &py)
{
    const double cr = e.bq0 * e.bq0 - e.bq3 * e.bq3;
    const double sr = 2.0 * (e.bq0 * e.bq3);
    px = e.x0 + (cr * e.q0 - sr * e.q1);
    py = e.y0 + (sr * e.q0 + cr * e.q1);
}

__device__ __forceinline__ Pose forward_pose(const EnvRegs &e)
{
    Pose o;
    const double cr = e.bq0 * e.bq0 - e.bq3 * e.bq3;
    const double sr = 2.0 * (e.bq0 * e.bq3);
    o.px = e.x0 + (cr * e.q0 - sr * e.q1);
    o.py = e.y0 + (sr * e.q0 + cr * e.q1);
    o.vx = cr * e.v0 - sr * e.v1;
    o.vy = sr * e.v0 + cr * e.v1;
    o.w = e.v2;
    double hs, hc;
    det_sincos_inl(0.5 * e.q2, hs, hc);
    o.xq0 = e.bq0 * hc - e.bq3 * hs;
    o.xq3 = e.bq0 * hs + e.bq3 * hc;
    return o;
}

// ZoneEnvBase.py:190-192,217-224 -> the 8-float 'obs' of wrappers.py:136-142
__device__ __forceinline__ void emit_obs8(const DevParams &p, const EnvRegs &e, float *o)
{
    const Pose f = forward_pose(e);
    o[0] = (float)(1.0 - div_const((double)e.steps, p.d_steps, p.inv_steps));
    o[1] = (float)div_const_z(f.px, 3.0, p.inv3);
    o[2] = (float)div_const_z(f.py, 3.0, p.inv3);
    const double a0 = (double)(float)f.xq0, a3 = (double)(float)f.xq3;
    o[3] = (float)(a0 * a0 - a3 * a3);
    o[4] = (float)((2.0 * a0) * a3);
    o[5] = (float)div_const_z(f.vx, 1.5, p.inv1_5);
    o[6] = (float)div_const_z(f.vy, 1.5, p.inv1_5);
    o[7] = (float)div_const_z(f.w, 3.0, p.inv3);
}

__device__ __forceinline__ void store_obs8(const DevParams &p, int env, const float *o)
{
    float4 *dst = reinterpret_cast<float4 *>(p.obs + (size_t)env * 8);
    dst[0] = make_float4(o[0], o[1], o[2], o[3]);
    dst[1] = make_float4(o[4], o[5], o[6], o[7]);
}

// One zone row of 'zone_obs' (TSP_env.py:31-35, TTSP_env.py:86-92, colour_match_env.py:75-80)
template <int TASK>
__device__ __forceinline__ void write_row(const DevParams &p, float *row, double zx, double zy,
                                          int flag_or_colour, int aux, int k)
{
    const float fx = (float)div_const(zx, 3.0, p.inv3);
    const float fy = (float)div_const(zy, 3.0, p.inv3);
    if (TASK == ZENV_TASK_COLOUR_MATCH) {
        row[0] = fx;
        row[1] = fy;
        row[2] = flag_or_colour == 2 ? 1.f : 0.f;
        row[3] = flag_or_colour == 1 ? 1.f : 0.f;
        row[4] = flag_or_colour == 0 ? 1.f : 0.f;
        row[5] = 0.25f;
        row[6] = (float)div_const((double)(float)aux, p.d_maxcd, p.inv_maxcd);
    } else if (TASK == ZENV_TASK_TIMED_TSP) {
        row[0] = fx;
        row[1] = fy;
        row[2] = flag_or_colour ? 1.f : 0.f;
        row[3] = 1.f;
        row[4] = flag_or_colour ? 0.f : 1.f;
        row[5] = 0.25f;
        row[6] = flag_or_colour ? 1.f : (float)div_const((double)(aux - k), p.d_steps, p.inv_steps);
    } else {
        // 24-byte rows: three 8-byte stores (ds_write_b64 when `row` is the LDS tile)
        float2 *r2 = reinterpret_cast<float2 *>(row);
        r2[0] = make_float2(fx, fy);
        r2[1] = make_float2(flag_or_colour ? 1.f : 0.f, 1.f);
        r2[2] = make_float2(flag_or_colour ? 0.f : 1.f, 0.25f);
    }
}

// Engine.reset for one env from bank slot `slot`: writes the SoA zone arrays, the lane's
// registers and the env's zone_obs rows (rows may point to LDS or to global memory).
template <int TASK>
__device__ void reset_env(const DevParams &p, int env, int slot, EnvRegs &e, float *rows)
{
    const int Z = p.Z, N = p.N, F = (TASK == ZENV_TASK_TSP) ? 6 : 7;
    const double *br = p.bank_robot + 4 * (size_t)slot;
    e.x0 = br[0]; e.y0 = br[1]; e.bq0 = br[2]; e.bq3 = br[3];
    e.q0 = e.q1 = e.q2 = 0.0;
    e.v0 = e.v1 = e.v2 = 0.0;
    e.vis = 0u;
    e.colpack = 0ull;
    e.goal_dist = 0;
    e.steps = 0;
    const double *bz = p.bank_zone + 2 * (size_t)slot * Z;
    const int32_t *ba = p.bank_aux + (size_t)slot * Z;
    for (int z = 0; z < Z; ++z) {
        const double zx = bz[2 * z], zy = bz[2 * z + 1];
        const size_t zi = (size_t)z * N + env;
        p.zxy[zi] = make_double2(zx, zy);
        int flag = 0, aux = 0;
        if (TASK == ZENV_TASK_TIMED_TSP) {
            aux = ba[z];
            p.tmax[zi] = aux;
        } else if (TASK == ZENV_TASK_COLOUR_MATCH) {
            flag = ba[z];
            e.colpack |= (uint64_t)flag << (2 * z);
            p.cooldown[zi] = 0;
        }
        write_row<TASK>(p, rows + z * F, zx, zy, flag, aux, 0);
    }
    if (TASK == ZENV_TASK_COLOUR_MATCH) e.goal_dist = hamming_to_goal(e.colpack, Z);
    p.seed[env] = p.bank_seed[slot];
}

__device__ __forceinline__ void store_dyn(const DevParams &p, int env, const EnvRegs &e)
{
    p.qa[env] = make_double2(e.q0, e.q1);
    p.qb[env] = make_double2(e.q2, e.v0);
    p.qc[env] = make_double2(e.v1, e.v2);
}

__device__ __forceinline__ void store_frame(const DevParams &p, int env, const EnvRegs &e)
{
    p.fa[env] = make_double2(e.x0, e.y0);
    p.fb[env] = make_double2(e.bq0, e.bq3);
}

__device__ __forceinline__ void store_regs(const DevParams &p, int env, int task, const EnvRegs &e,
                                           bool frame_too)
{
    store_dyn(p, env, e);
    if (frame_too) store_frame(p, env, e);
    p.steps[env] = e.steps;
    if (task == ZENV_TASK_COLOUR_MATCH) {
        p.colpack[env] = e.colpack;
        p.goal_dist[env] = e.goal_dist;
    } else {
        p.vis[env] = e.vis;
    }
}

// MuJoCo mj_step for point.xml: 3 dof (slide x, slide y, hinge z), offset COM, implicit
// joint damping, no active constraints.  Operation order is the
// oracle's, token for token.
__device__ __forceinline__ void mj_substep(const DevParams &p, EnvRegs &e, double c0, double c1)
{
    double s, k;
    det_sincos_inl(e.q2, s, k);
    const double mcs = p.mc * s, mck = p.mc * k;
    const double w2 = e.v2 * e.v2;
    const double f0 = det_clamp(c0, -p.fmax, p.fmax);
    const double f1 = det_clamp(p.kv * c1 - p.kv * (p.gear * e.v2), -p.fmax, p.fmax);
    const double gf0 = p.gear * f0;
    const double rhs0 = (gf0 * k + mck * w2) - p.b0 * e.v0;
    const double rhs1 = (gf0 * s + mcs * w2) - p.b1 * e.v1;
    const double rhs2 = p.gear * f1 - p.b2 * e.v2;
    const double t0 = rhs0 * p.inv00, t1 = rhs1 * p.inv11;
    const double den = (p.A22 - (mcs * mcs) * p.inv00) - (mck * mck) * p.inv11;
    const double num = (rhs2 + mcs * t0) - mck * t1;
    const double a2 = num / den;
    const double a0 = (rhs0 + mcs * a2) * p.inv00;
    const double a1 = (rhs1 - mck * a2) * p.inv11;
    e.v0 = e.v0 + p.h * a0;
    e.v1 = e.v1 + p.h * a1;
    e.v2 = e.v2 + p.h * a2;
    e.q0 = e.q0 + p.h * e.v0;
    e.q1 = e.q1 + p.h * e.v1;
    e.q2 = e.q2 + p.h * e.v2;
}

// Contiguous LDS tile -> HBM burst: [n_env_blk][ZF] floats, 16 B per lane per store
__device__ __forceinline__ void flush_tile(const float *tile, float *dst, int n_floats, int lane)
{
    const int n4 = n_floats >> 2;
    const float4 *s4 = reinterpret_cast<const float4 *>(tile);
    float4 *d4 = reinterpret_cast<float4 *>(dst);
    int i = lane;
    // four ds_read_b128 in flight per lane before the four dwordx4 stores
    for (; i + 3 * kWave < n4; i += 4 * kWave) {
        const float4 a = s4[i], b = s4[i + kWave], c = s4[i + 2 * kWave], d = s4[i + 3 * kWave];
        d4[i] = a; d4[i + kWave] = b; d4[i + 2 * kWave] = c; d4[i + 3 * kWave] = d;
    }
    for (; i < n4; i += kWave) d4[i] = s4[i];
    for (int j = (n4 << 2) + lane; j < n_floats; j += kWave) dst[j] = tile[j];
}

// =========================================================================== K3: policies
// Scripted action sources (the build's own, same definition as oracle/zenv_oracle.c:orc_policy).
// Both are functions of the float32 observation only.
__device__ __forceinline__ void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1)
{
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
        const uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
        c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}

__device__ __forceinline__ float2 uniform_action(uint64_t global_env, uint32_t step_index, uint64_t seed)
{
    uint32_t c[4] = { (uint32_t)global_env, (uint32_t)(global_env >> 32), step_index, 0u };
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    float2 a;
    a.x = 2.0f * ((float)(c[0] >> 8) * 5.9604644775390625e-08f) - 1.0f;
    a.y = 2.0f * ((float)(c[1] >> 8) * 5.9604644775390625e-08f) - 1.0f;
    return a;
}

// Steer towards the nearest eligible zone.  rows: this env's [Z][F] float32 zone_obs rows
// (LDS or global); (opx, opy, ohx, ohy) = obs[1..4].
template <int TASK, int ZT = 0>
__device__ __forceinline__ float2 greedy_action(const float *rows, int Zrt, float opx, float opy, float ohx,
                                                float ohy)
{
    constexpr int F = (TASK == ZENV_TASK_TSP) ? 6 : 7;
    const int Z = ZT > 0 ? ZT : Zrt;
    const double px = 3.0 * (double)opx, py = 3.0 * (double)opy;
    const double hx = (double)ohx, hy = (double)ohy;
    int target_colour = -1;
    if (TASK == ZENV_TASK_COLOUR_MATCH) {
        int cb = 0, cg = 0, cr = 0;
#pragma unroll
        for (int z = 0; z < Z; ++z) {
            const float *row = rows + z * F;
            const int col = row[4] != 0.f ? 0 : (row[3] != 0.f ? 1 : 2);
            cb += col == 0; cg += col == 1; cr += col == 2;
        }
        target_colour = 0;
        int best_cnt = cb;
        if (cg > best_cnt) { target_colour = 1; best_cnt = cg; }
        if (cr > best_cnt) { target_colour = 2; }
    }
    // nearest eligible zone, lowest index on ties (== the oracle's sequential `d2 < best` scan)
    int best = -1;
    double bd2 = 0.0, bdx = 0.0, bdy = 0.0;
    if (ZT > 0) {
        // all distances first (independent), then a pairwise tournament: log-depth dependency
        // chain instead of a Z-long one.  Ineligible zones carry +inf.
        constexpr int ZP = ZT > 0 ? ZT : 1;
        double d2s[ZP];
        int idx[ZP];
#pragma unroll
        for (int z = 0; z < ZP; ++z) {
            const float *row = rows + z * F;
            bool eligible;
            if (TASK == ZENV_TASK_COLOUR_MATCH) {
                const int col = row[4] != 0.f ? 0 : (row[3] != 0.f ? 1 : 2);
                eligible = (row[6] == 0.f) && (col != target_colour);
            } else {
                eligible = row[2] == 0.f;
            }
            const double dx = 3.0 * (double)row[0] - px, dy = 3.0 * (double)row[1] - py;
            const double d2 = dx * dx + dy * dy;
            d2s[z] = eligible ? d2 : __builtin_inf();
            idx[z] = z;
        }
#pragma unroll
        for (int stride = 1; stride < ZP; stride *= 2) {
#pragma unroll
            for (int z = 0; z + stride < ZP; z += 2 * stride) {
                // the left entry always has the lower zone index: it wins ties
                const bool take = d2s[z + stride] < d2s[z];
                d2s[z] = take ? d2s[z + stride] : d2s[z];
                idx[z] = take ? idx[z + stride] : idx[z];
            }
        }
        if (d2s[0] < __builtin_inf()) {
            best = idx[0];
            const float *row = rows + best * F;
            bdx = 3.0 * (double)row[0] - px;
            bdy = 3.0 * (double)row[1] - py;
            bd2 = bdx * bdx + bdy * bdy;
        }
    } else {
        for (int z = 0; z < Z; ++z) {
            const float *row = rows + z * F;
            bool eligible;
            if (TASK == ZENV_TASK_COLOUR_MATCH) {
                const int col = row[4] != 0.f ? 0 : (row[3] != 0.f ? 1 : 2);
                eligible = (row[6] == 0.f) && (col != target_colour);
            } else {
                eligible = row[2] == 0.f;
            }
            const double dx = 3.0 * (double)row[0] - px, dy = 3.0 * (double)row[1] - py;
            const double d2 = dx * dx + dy * dy;
            if (eligible && (best < 0 || d2 < bd2)) { best = z; bd2 = d2; bdx = dx; bdy = dy; }
        }
    }
    float2 a = make_float2(0.f, 0.f);
    if (best >= 0 && bd2 > 1e-18) {
        const double n = sqrt(bd2);
        const double cs = (hx * bdx + hy * bdy) / n;
        const double sn = (hx * bdy - hy * bdx) / n;
        if (cs < 0.0) a.y = sn >= 0.0 ? 1.f : -1.f;
        else a.y = (float)det_clamp(4.0 * sn, -1.0, 1.0);
        a.x = cs > 0.8 ? 1.f : 0.f;
    }
    return a;
}

template <int TASK, int ZT>
__device__ __forceinline__ float2 scripted_action(const StepPolicy &pol, int env, const float *rows, int Z,
                                                  const float *o)
{
    if (pol.policy == ZENV_POLICY_UNIFORM)
        return uniform_action(pol.env_index0 + (uint64_t)env, pol.step_index, pol.seed);
    return greedy_action<TASK, ZT>(rows, Z, o[1], o[2], o[3], o[4]);
}

template <int TASK>
__global__ __launch_bounds__(kWave) void k_policy_lane(DevParams p, StepPolicy pol)
{
    extern __shared__ __align__(16) float tile[];
    constexpr int F = (TASK == ZENV_TASK_TSP) ? 6 : 7;
    const int lane = threadIdx.x;
    const int env0 = blockIdx.x * kWave;
    const int env = env0 + lane;
    const int N = p.N, Z = p.Z, ZF = Z * F;

    if (pol.policy == ZENV_POLICY_UNIFORM) {
        if (env < N)
            reinterpret_cast<float2 *>(pol.out)[env] =
                uniform_action(pol.env_index0 + (uint64_t)env, pol.step_index, pol.seed);
        return;
    }
    // greedy: stage the wave's zone_obs rows through LDS (coalesced 16 B/lane loads)
    {
        const int n_blk = min(kWave, N - env0);
        const int n_floats = n_blk * ZF, n4 = n_floats >> 2;
        const float *src = p.zone_obs + (size_t)env0 * ZF;
        const float4 *s4 = reinterpret_cast<const float4 *>(src);
        float4 *t4 = reinterpret_cast<float4 *>(tile);
        for (int i = lane; i < n4; i += kWave) t4[i] = s4[i];
        for (int i = (n4 << 2) + lane; i < n_floats; i += kWave) tile[i] = src[i];
    }
    __syncthreads();
    if (env >= N) return;
    const float4 *ob = reinterpret_cast<const float4 *>(p.obs + (size_t)env * 8);
    const float4 oa = ob[0], obb = ob[1];
    reinterpret_cast<float2 *>(pol.out)[env] = greedy_action<TASK>(tile + lane * ZF, Z, oa.y, oa.z, oa.w, obb.x);
}

// =========================================================================== K1: step
// A 128-thread workgroup (two wave64) owns a tile of 64 consecutive envs, lane i <-> env i in
// BOTH waves, with the work split by what it is bound by:
//   wave 0, "zone wave"   (memory):  streams the zone-major zone arrays, runs set_mocaps(),
//       reward / goal / termination / auto-reset (none of which needs this step's physics:
//       set_mocaps() sees the PRE-physics pose), builds the 64 x Z x F float32 tile in LDS
//       and flushes it as one contiguous burst of dwordx4 stores;
//   wave 1, "physics wave" (latency): the 10 serial MuJoCo substeps and the 8-float obs.
// The two meet once (one s_barrier): the zone wave tells the physics wave per env whether
// its result is observable (mode 0) or superseded by a reset / masked no-op (mode 1) and
// what the new step count is.  On different SIMDs the ~5 us serial fp64 chain of the physics
// wave runs underneath the zone wave's load->LDS->store stream instead of after it.
//
// ZT > 0: zone count known at compile time -> the zone loop is fully unrolled and every
// zone load of the wave (2*Z x 512 B, + Z x 256 B tmax / Z x 64 B cooldown) is issued up
// front.  ZT == 0: generic runtime-Z fallback with the loads inside the loop.
constexpr int kStepThreads = 2 * kWave;

// Diagnostic build only (-DZENV_STAMPS, never shipped): lane 0 of each wave drops
// s_memrealtime (100 MHz) stamps into p.dbg[block][16] at phase boundaries.
#ifdef ZENV_STAMPS
#define ZSTAMP(slot)                                                                      \
    do {                                                                                  \
        if (p.dbg && lane == 0) p.dbg[(size_t)blockIdx.x * 16 + (slot)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define ZSTAMP(slot) do { } while (0)
#endif

// amdgpu_waves_per_eu(2, 2): the grid needs exactly two waves per SIMD (4 tiles x 2 waves per
// CU), so the scheduler may spend up to 256 VGPRs to keep every zone load in flight at once
// instead of sinking loads to save registers for an occupancy the launch never uses.
template <int TASK, int ZT>
__global__ __launch_bounds__(kStepThreads) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_step_lane(DevParams p, const float *__restrict__ actions,
                                                            int auto_reset, StepPolicy pol)
{
    extern __shared__ __align__(16) float tile[];
    constexpr int F = (TASK == ZENV_TASK_TSP) ? 6 : 7;
    constexpr int ZR = ZT > 0 ? ZT : 1;
    const int lane = threadIdx.x & (kWave - 1);
    const int role = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform
    const int env0 = blockIdx.x * kWave;
    const int env = env0 + lane;
    const int N = p.N;
    const int Z = ZT > 0 ? ZT : p.Z;
    const int ZF = Z * F;
    float *rows = tile + lane * ZF;
    int *xmode = reinterpret_cast<int *>(tile + kWave * ZF);   // zone wave -> physics wave
    int *xstep = xmode + kWave;

    EnvRegs e;
    e.steps = 0;
    float o[8];   // physics wave: this step's obs

    if (role == 0) {
        // =================================================================== zone wave
        ZSTAMP(0);
        if (env < N) {
            // ---- issue every load of this env first
            const uint8_t was_done = p.done_state[env];
            {
                const double2 qa = p.qa[env], fa = p.fa[env], fb = p.fb[env];
                e.q0 = qa.x; e.q1 = qa.y;
                e.x0 = fa.x; e.y0 = fa.y; e.bq0 = fb.x; e.bq3 = fb.y;
            }
            e.steps = p.steps[env];
            e.vis = 0u; e.colpack = 0ull; e.goal_dist = 0;
            if (TASK == ZENV_TASK_COLOUR_MATCH) {
                e.colpack = p.colpack[env];
                e.goal_dist = p.goal_dist[env];
            } else {
                e.vis = p.vis[env];
            }
            double ep_ret = p.ep_return[env];
            double zxr[ZR], zyr[ZR];
            int auxr[ZR];
            if (ZT > 0) {
                // issue order = consumption order: pose first, then zone 0, 1, ... so the
                // in-order vmcnt waits of the zone pass release one zone at a time
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int z = 0; z < ZR; ++z) {
                    const size_t zi = (size_t)z * N + env;
                    const double2 zz = p.zxy[zi];   // one 16 B/lane load: 1 KiB per wave
                    zxr[z] = zz.x;
                    zyr[z] = zz.y;
                    auxr[z] = 0;
                    if (TASK == ZENV_TASK_TIMED_TSP) auxr[z] = p.tmax[zi];
                    if (TASK == ZENV_TASK_COLOUR_MATCH) auxr[z] = p.cooldown[zi];
                }
                // nothing below may be scheduled above this point, nor any load below it
                __builtin_amdgcn_sched_barrier(0);
            }

            float rew_out = 0.f;
            uint8_t done_out = 1, goal_out = 0;
            int mode = 1;
            if (was_done) {
                // finished and not auto-reset: masked no-op (WaitWrapper, wrappers.py:34-45)
                for (int i = 0; i < ZF; ++i) rows[i] = 0.f;
                for (int i = 0; i < 8; ++i) o[i] = 0.f;
                store_obs8(p, env, o);
                if (pol.policy >= 0) reinterpret_cast<float2 *>(pol.out)[env] = make_float2(0.f, 0.f);
            } else {
                const int k = e.steps + 1;   // step index after this call
                double rx, ry;               // pre-physics pose: what set_mocaps() sees
                world_pos(e, rx, ry);
                ZSTAMP(1);

                // ---- zone pass: set_mocaps() of the first substep
                int first = -1;
                bool timed_out = false;
#pragma unroll
                for (int z = 0; z < (ZT > 0 ? ZT : Z); ++z) {
                    const size_t zi = (size_t)z * N + env;
                    double zx, zy;
                    int aux = 0;
                    if (ZT > 0) {
                        zx = zxr[z]; zy = zyr[z]; aux = auxr[z];
                    } else {
                        const double2 zz = p.zxy[zi];
                        zx = zz.x; zy = zz.y;
                        if (TASK == ZENV_TASK_TIMED_TSP) aux = p.tmax[zi];
                        if (TASK == ZENV_TASK_COLOUR_MATCH) aux = p.cooldown[zi];
                    }
                    const double dx = zx - rx, dy = zy - ry;
                    const double d2 = dx * dx + dy * dy;
                    const bool inside = d2 <= p.hit_d2;
                    if (TASK == ZENV_TASK_COLOUR_MATCH) {
                        int cd = aux;
                        if (cd > 0) cd -= 1;                       // colour_match_env.py:98-100
                        int col = (int)((e.colpack >> (2 * z)) & 3ull);
                        if (first < 0 && cd == 0 && inside) {       // :106-120, lowest index wins
                            first = z;
                            col = (col == 2) ? 0 : col + 1;         // Blue->Green->Red->Blue
                            e.colpack = (e.colpack & ~(3ull << (2 * z))) | ((uint64_t)col << (2 * z));
                            cd = p.max_cd;
                        }
                        p.cooldown[zi] = (uint8_t)cd;
                        write_row<TASK>(p, rows + z * F, zx, zy, col, cd, k);
                    } else {
                        bool vis = (e.vis >> z) & 1u;
                        if (first < 0 && !vis && inside) {          // TSP_env.py:54-69
                            first = z;
                            vis = true;
                            e.vis |= 1u << z;
                        }
                        if (TASK == ZENV_TASK_TIMED_TSP) {
                            if (!vis && (aux - k) <= 0) timed_out = true;   // TTSP_env.py:67
                        }
                        write_row<TASK>(p, rows + z * F, zx, zy, vis ? 1 : 0, aux, k);
                    }
                }

                // ---- reward / goal / termination (Engine.step order; none of it needs the physics)
                double r = 0.0;
                bool goal;
                if (TASK == ZENV_TASK_COLOUR_MATCH) {
                    if (first >= 0) {
                        const int nd = hamming_to_goal(e.colpack, Z);
                        r = (double)(e.goal_dist - nd);
                        e.goal_dist = nd;
                    }
                    goal = e.goal_dist == 0;
                } else {
                    r = first >= 0 ? 1.0 : 0.0;
                    const uint32_t full = (Z >= 32) ? 0xFFFFFFFFu : ((1u << Z) - 1u);
                    goal = e.vis == full;
                }
                bool done = false;
                if (goal) {
                    r += (double)(p.num_steps - e.steps) * p.tsr;   // pre-increment steps
                    done = true;
                    goal_out = 1;
                }
                e.steps = k;
                if (k >= p.num_steps) done = true;
                if (TASK == ZENV_TASK_TIMED_TSP && !done && timed_out) done = true;

                ep_ret = ep_ret + r;
                rew_out = (float)r;
                done_out = done ? 1 : 0;
                p.visit_count[env] = (TASK == ZENV_TASK_COLOUR_MATCH) ? e.goal_dist : __popc(e.vis);

                mode = 0;
                if (done) {
                    p.last_return[env] = ep_ret;
                    p.last_len[env] = k;
                    p.episodes[env] += 1;
                    if (auto_reset) {
                        // penv.py:8-11: the returned obs is the first obs of the next episode;
                        // this step's physics result is never observed (mode 1 discards it)
                        const int slot = next_bank_slot(p, env);
                        reset_env<TASK>(p, env, slot, e, rows);
                        ep_ret = 0.0;
                        mode = 1;
                        store_frame(p, env, e);
                        store_dyn(p, env, e);   // reset_env zeroed q, v
                        emit_obs8(p, e, o);
                        store_obs8(p, env, o);
                        if (pol.policy >= 0)   // next action from the new episode's first obs
                            reinterpret_cast<float2 *>(pol.out)[env] = scripted_action<TASK, ZT>(pol, env, rows, Z, o);
                    } else {
                        p.done_state[env] = 1;
                    }
                }
                p.ep_return[env] = ep_ret;
                p.steps[env] = e.steps;
                if (TASK == ZENV_TASK_COLOUR_MATCH) {
                    p.colpack[env] = e.colpack;
                    p.goal_dist[env] = e.goal_dist;
                } else {
                    p.vis[env] = e.vis;
                }
            }
            xmode[lane] = mode;
            xstep[lane] = e.steps;
            p.reward[env] = rew_out;
            p.done_out[env] = done_out;
            p.goal_met[env] = goal_out;
        }
        ZSTAMP(2);
    } else {
        // =================================================================== physics wave
        ZSTAMP(8);
        if (env < N) {
            {
                const double2 qa = p.qa[env], qb = p.qb[env], qc = p.qc[env];
                const double2 fa = p.fa[env], fb = p.fb[env];
                e.q0 = qa.x; e.q1 = qa.y; e.q2 = qb.x;
                e.v0 = qb.y; e.v1 = qc.x; e.v2 = qc.y;
                e.x0 = fa.x; e.y0 = fa.y; e.bq0 = fb.x; e.bq3 = fb.y;
            }
            const float2 act = reinterpret_cast<const float2 *>(actions)[env];
            // Engine.step: ctrl = clip(action, ctrlrange); frameskip x mj_step
            const double c0 = det_clamp((double)act.x, -1.0, 1.0);
            const double c1 = det_clamp((double)act.y, -1.0, 1.0);
            ZSTAMP(9);
            for (int i = 0; i < p.frameskip; ++i) mj_substep(p, e, c0, c1);
            emit_obs8(p, e, o);   // o[0] (remaining) is patched after the rendezvous
        }
        ZSTAMP(10);
    }

    // The one rendezvous of the two waves: the tile rows + per-env mode / step count are in
    // LDS.  After it the zone wave streams the tile out while the physics wave finishes.
    __syncthreads();
    if (role == 0) ZSTAMP(4); else ZSTAMP(11);

    if (role == 0) {
        const int n_blk = min(kWave, N - env0);
        flush_tile(tile, p.zone_obs + (size_t)env0 * ZF, n_blk * ZF, lane);
        ZSTAMP(3);
    } else if (env < N) {
        if (xmode[lane] == 0) {
            e.steps = xstep[lane];
            store_dyn(p, env, e);
            o[0] = (float)(1.0 - div_const((double)e.steps, p.d_steps, p.inv_steps));
            store_obs8(p, env, o);
            // fused K3: the action of the NEXT step, from this step's obs and the tile in LDS
            ZSTAMP(12);
            if (pol.policy >= 0)
                reinterpret_cast<float2 *>(pol.out)[env] = scripted_action<TASK, ZT>(pol, env, rows, Z, o);
        }
        ZSTAMP(13);
    }
}

// =========================================================================== K2: reset
template <int TASK>
__global__ __launch_bounds__(kWave) void k_reset_lane(DevParams p, const uint8_t *__restrict__ mask)
{
    constexpr int F = (TASK == ZENV_TASK_TSP) ? 6 : 7;
    const int env = blockIdx.x * kWave + threadIdx.x;
    if (env >= p.N) return;
    if (mask && !mask[env]) return;
    EnvRegs e;
    const int slot = next_bank_slot(p, env);
    reset_env<TASK>(p, env, slot, e, p.zone_obs + (size_t)env * p.Z * F);
    store_regs(p, env, TASK, e, true);
    p.done_state[env] = 0;
    p.ep_return[env] = 0.0;
    p.visit_count[env] = (TASK == ZENV_TASK_COLOUR_MATCH) ? e.goal_dist : 0;
    float o[8];
    emit_obs8(p, e, o);
    store_obs8(p, env, o);
    p.reward[env] = 0.f;
    p.done_out[env] = 0;
    p.goal_met[env] = 0;
}

}  // namespace

// ---------------------------------------------------------------------------- launchers
static inline int n_blocks(int n) { return (n + kWave - 1) / kWave; }
static inline size_t tile_bytes(const DevParams &p) { return (size_t)kWave * p.Z * p.F * sizeof(float); }
static inline size_t step_lds_bytes(const DevParams &p) { return tile_bytes(p) + 2 * kWave * sizeof(int); }

template <int TASK>
static void launch_step_task(const DevParams &p, const float *actions, int auto_reset, const StepPolicy &pol,
                             hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop)
{
    const dim3 grid(n_blocks(p.N)), block(kStepThreads);
    const size_t lds = step_lds_bytes(p);
    // hipExtLaunchKernelGGL stamps ev_start/ev_stop with the dispatch's own begin/end
#define ZENV_LAUNCH(ZT)                                                                                 \
    hipExtLaunchKernelGGL((k_step_lane<TASK, ZT>), grid, block, lds, s, ev_start, ev_stop, 0, p, actions, \
                          auto_reset, pol)
    switch (p.Z) {
    case 5: ZENV_LAUNCH(5); break;
    case 6: ZENV_LAUNCH(6); break;
    case 15: ZENV_LAUNCH(15); break;
    case 25: ZENV_LAUNCH(25); break;
    default: ZENV_LAUNCH(0); break;
    }
#undef ZENV_LAUNCH
}

hipError_t launch_step(const DevParams &p, const float *actions, int auto_reset, const StepPolicy &pol,
                       hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop)
{
    switch (p.task) {
    case ZENV_TASK_TSP: launch_step_task<ZENV_TASK_TSP>(p, actions, auto_reset, pol, s, ev_start, ev_stop); break;
    case ZENV_TASK_TIMED_TSP: launch_step_task<ZENV_TASK_TIMED_TSP>(p, actions, auto_reset, pol, s, ev_start, ev_stop); break;
    default: launch_step_task<ZENV_TASK_COLOUR_MATCH>(p, actions, auto_reset, pol, s, ev_start, ev_stop); break;
    }
    return hipGetLastError();
}

hipError_t launch_reset(const DevParams &p, const uint8_t *mask, hipStream_t s)
{
    const dim3 grid(n_blocks(p.N)), block(kWave);
    switch (p.task) {
    case ZENV_TASK_TSP:
        hipLaunchKernelGGL(k_reset_lane<ZENV_TASK_TSP>, grid, block, 0, s, p, mask);
        break;
    case ZENV_TASK_TIMED_TSP:
        hipLaunchKernelGGL(k_reset_lane<ZENV_TASK_TIMED_TSP>, grid, block, 0, s, p, mask);
        break;
    default:
        hipLaunchKernelGGL(k_reset_lane<ZENV_TASK_COLOUR_MATCH>, grid, block, 0, s, p, mask);
        break;
    }
    return hipGetLastError();
}

hipError_t launch_policy(const DevParams &p, const StepPolicy &pol, hipStream_t s)
{
    const dim3 grid(n_blocks(p.N)), block(kWave);
    const size_t lds = pol.policy == ZENV_POLICY_GREEDY ? tile_bytes(p) : 0;
    switch (p.task) {
    case ZENV_TASK_TSP: hipLaunchKernelGGL(k_policy_lane<ZENV_TASK_TSP>, grid, block, lds, s, p, pol); break;
    case ZENV_TASK_TIMED_TSP: hipLaunchKernelGGL(k_policy_lane<ZENV_TASK_TIMED_TSP>, grid, block, lds, s, p, pol); break;
    default: hipLaunchKernelGGL(k_policy_lane<ZENV_TASK_COLOUR_MATCH>, grid, block, lds, s, p, pol); break;
    }
    return hipGetLastError();
}

}  // namespace zenvk

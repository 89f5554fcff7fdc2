// kernels.hip -- gfx950 (CDNA4, wave64) kernels of the batched zone-env hot path.
//
// K1  k_step_lane<TASK,ZT>  one env.step() for every env (TSP_env.py:45-72, TTSP_env.py:62-71,
//                           colour_match_env.py:86-123, ZoneEnvBase.py:143-235 + the not-vendored
//                           Engine.step / mj_step of xmls/point.xml), with the ParallelEnv
//                           auto-reset of penv.py:8-11 and (optionally) the scripted action
//                           source of the next step fused in.
// K1p k_rollout_lane<TASK,ZT> the same step, up to 256 of them per launch, when the action source is on the
//                           device: env state in registers, one wave steps the envs, one streams the rows.
// K2  k_reset_lane<TASK>    masked re-init from the HBM layout bank (Engine.reset).
// K3  k_policy_lane<TASK>   stand-alone scripted action sources (the build's own).
// K6  k_goal_set/step/clear goal-conditioned variants (TSP_next_city_env.py and its zone-goals siblings).
//
// Execution shape: a tile = 64 consecutive envs, lane i <-> env i.  State is struct-of-arrays
// in 16-byte pairs (1 KiB per wave load), zone arrays are zone-major.  The (N,Z,F) float32
// zone_obs rows are staged in LDS as compact 16-byte (x/3, y/3, code, aux) entries and expanded
// to their F floats while they stream out as contiguous dwordx4 bursts.
// No MFMA: there is no contraction anywhere on this path (the actor network is mlp_policy.hip).
// Build with -ffp-contract=off: the float64 state must match the CPU oracle bit for bit; every fused
// multiply-add is an explicit __builtin_fma, mirrored in oracle/zenv_oracle.c.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include "det_math.hpp"
#include "dev_params.hpp"
#include "kernels.hpp"

namespace zenvk {

namespace {

constexpr int kWave = 64;

// Diagnostic build only (-DZENV_STAMPS, never shipped): lane 0 of each wave drops
#ifndef ZENV_SUBSTEP_UNROLL
#define ZENV_SUBSTEP_UNROLL 10      // = the reference's frameskip (Engine frameskip_binom_n, p = 1.0)
#endif

// s_memrealtime (100 MHz) stamps into p.dbg[block][16] at phase boundaries.
#ifdef ZENV_STAMPS
#define ZSTAMP(slot)                                                                                  \
    do {                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        unsigned long long t_;                                                                        \
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        if (p.dbg && lane == 0) p.dbg[(size_t)blockIdx.x * 16 + (slot)] = t_;                         \
    } while (0)
#else
#define ZSTAMP(slot) do { } while (0)
#endif

// Rows per flush chunk of the 6-float rows (RPC * 6 floats must be whole float4: 2 or 4).  4 halves the number of
// expand -> slab -> store iterations of a 25-zone tile (13 -> 7), which is what made PointTSP-25 flush slower per byte
// than TimedTSP-25's 7-float rows.
#ifndef ZENV_RPC6
#define ZENV_RPC6 4
#endif

// Every wave of a launch starts by reading the same ~700 bytes of kernel arguments (DevParams by value).  Left to the
// compiler those reads come as a dozen scalar loads spread through the prologue and the rare branches, re-issued
// wherever SGPR pressure dropped a value, each followed by its own s_waitcnt: on the zone wave's path to "pose known"
// that was eight to ten SERIALISED scalar-cache misses at the moment all 2 048 waves of the grid miss on the same
// lines (round 3 stamps: pose ready 2.3 us after the first wave of the grid started, for 0.4 us in the wave that
// needs four fields).  This touches every 64-byte line of the argument block at once -- all misses in flight
// together, one wait -- so that the compiler's own loads behind it are scalar-cache hits.  (The loaded values are
// thrown away; one asm statement, so the destination registers are dead only after its s_waitcnt.)
typedef int v16i_t __attribute__((ext_vector_type(16)));
template <int BYTES>
__device__ __forceinline__ void warm_kernarg()
{
#ifndef ZENV_NO_KERNARG_WARM
    const auto k = __builtin_amdgcn_kernarg_segment_ptr();   // constant address space: an SGPR pair
    v16i_t t;
    static_assert(BYTES <= 12 * 64, "extend the list");
    asm volatile("s_load_dwordx16 %0, %1, 0x0\n\t"
                 "s_load_dwordx16 %0, %1, 0x40\n\t"
                 "s_load_dwordx16 %0, %1, 0x80\n\t"
                 "s_load_dwordx16 %0, %1, 0xc0\n\t"
                 "s_load_dwordx16 %0, %1, 0x100\n\t"
                 "s_load_dwordx16 %0, %1, 0x140\n\t"
                 "s_load_dwordx16 %0, %1, 0x180\n\t"
                 "s_load_dwordx16 %0, %1, 0x1c0\n\t"
                 "s_load_dwordx16 %0, %1, 0x200\n\t"
                 "s_load_dwordx16 %0, %1, 0x240\n\t"
                 "s_load_dwordx16 %0, %1, 0x280\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&s"(t)
                 : "s"(k)
                 : "memory");
#endif
}

template <int TASK>
struct TaskTraits {
    static constexpr int F = (TASK == ZENV_TASK_TSP) ? 6 : 7;        // floats per zone row
    static constexpr int RPC = (F == 6) ? ZENV_RPC6 : 4;             // rows per flush chunk (RPC * F floats = G float4)
    static constexpr int G = RPC * F / 4;                            // float4 per flush chunk
};

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

// ColourMatch cooldowns: one byte per zone, EIGHT zones of one env per 64-bit word, words [ceil(Z/8)][N] -- an env's
// cooldowns are one 8-byte load and one 8-byte store per step (Z <= 8) instead of Z byte-wide read-modify-writes of
// half-empty cache lines (round 2: u8 [Z][N], PMC traffic 1.28x the algorithmic bytes on ColourMatch-6).
template <typename P>
__device__ __forceinline__ uint64_t *cd_word(const P &p, int w, int env)
{
    if (w == 0) return &p.hotc[env].cd0;
    return reinterpret_cast<uint64_t *>(p.cooldown) + (size_t)(w - 1) * p.N + env;
}
template <typename P>
__device__ __forceinline__ uint8_t *cd_byte(const P &p, int z, int env)
{
    return reinterpret_cast<uint8_t *>(cd_word(p, z >> 3, env)) + (z & 7);
}
// every non-zero byte of w minus one (colour_match_env.py:98-100 for eight zones at once; no borrow crosses a byte)
__device__ __forceinline__ uint64_t cd_decrement(uint64_t w)
{
    const uint64_t lo7 = 0x7F7F7F7F7F7F7F7Full;
    const uint64_t nz = (((w & lo7) + lo7) | w) & 0x8080808080808080ull;
    return w - (nz >> 7);
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

// (P = DevParams, or the same block seen through the constant address space: see k_rollout_lane)
template <typename P>
__device__ __forceinline__ uint32_t pcg_next32_dev(const P &p, int env)
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

// Bank slot of env's next episode (and advance the schedule).  episode index / first slot are
// passed in when the caller has already loaded them.
// slot of episode k under the two host-laid-out schedules: sequential (first + k * stride, wrapping around the bank)
// and ring (env i owns the slots first .. first + depth - 1 and walks round them; the host refills a slot with episode
// k + depth once episode k has been taken from it -- Engine.reset's unbounded _seed += 1 stream, zenv_bank_update)
template <typename P>
__device__ __forceinline__ int seq_slot(const P &p, int first, int k)
{
    if (p.sched_mode == SCHED_RING) return first + k % p.sched_stride;
    const long long s = (long long)first + (long long)k * (long long)p.sched_stride;
    return (int)(s % (long long)p.bank_size);
}

// The slot after `slot` in env's schedule, without a division: sequential walks the bank by sched_stride (< bank_size,
// one conditional subtraction), ring walks round its own `depth` slots.  Kept per env in p.next_slot (state), so that
// the step kernels know the slot of an env's NEXT episode from one 4-byte load: seq_slot()'s 64-bit modulo by a run-time
// divisor used to sit on the zone wave's path of every step in which some env of the tile was close to its end (the
// reset-prefetch hint) -- ~150 instructions.  Invariant: next_slot[env] == seq_slot(slot_first[env], episode_idx[env])
// (k_sched_sync re-establishes it when the host changes the bank or the schedule).
template <typename P>
__device__ __forceinline__ int slot_after(const P &p, int env, int slot)
{
    if (p.sched_mode == SCHED_RING) {
        const int first = p.slot_first[env];
        const int n = slot + 1;
        return n >= first + p.sched_stride ? first : n;
    }
    const int n = slot + p.sched_stride;
    return n >= p.bank_size ? n - p.bank_size : n;
}

// Bank slot of env's next episode (and advance the schedule).  k = its episode index, nslot = p.sched[env].next_slot, passed
// in when the caller has already loaded them; nslot comes back advanced.
template <typename P>
__device__ __forceinline__ int next_bank_slot(const P &p, int env, int k, int &nslot)
{
    p.sched[env].episode_idx = k + 1;
    if (p.sched_mode != SCHED_FIXED_SEEDS) {
        const int slot = nslot;
        nslot = slot_after(p, env, slot);
        p.sched[env].next_slot = nslot;
        return slot;
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
// The same on a register copy of the env's schedule record (k_step_lane stores the record whole at its end).
template <typename P>
__device__ __forceinline__ int next_bank_slot_rec(const P &p, int env, Sched &sc)
{
    const int k = sc.episode_idx;
    sc.episode_idx = k + 1;
    if (p.sched_mode != SCHED_FIXED_SEEDS) {
        const int slot = sc.next_slot;
        sc.next_slot = slot_after(p, env, slot);
        return slot;
    }
    int unused = 0;
    const int slot = next_bank_slot(p, env, k, unused);     // draws from the env's PCG64 stream (its own arrays)
    sc.episode_idx = k + 1;
    return slot;
}
template <typename P>
__device__ __forceinline__ int next_bank_slot(const P &p, int env)
{
    int nslot = p.sched[env].next_slot;
    return next_bank_slot(p, env, p.sched[env].episode_idx, nslot);
}

// bit i of x -> bit 2i of the result
__device__ __forceinline__ uint64_t spread_even_bits(uint32_t x32)
{
    uint64_t x = x32;
    x = (x | (x << 16)) & 0x0000FFFF0000FFFFull;
    x = (x | (x << 8)) & 0x00FF00FF00FF00FFull;
    x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full;
    x = (x | (x << 2)) & 0x3333333333333333ull;
    x = (x | (x << 1)) & 0x5555555555555555ull;
    return x;
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
    const double cr = __builtin_fma(e.bq0, e.bq0, -(e.bq3 * e.bq3));
    const double sr = 2.0 * (e.bq0 * e.bq3);
    px = __builtin_fma(cr, e.q0, __builtin_fma(-sr, e.q1, e.x0));
    py = __builtin_fma(sr, e.q0, __builtin_fma(cr, e.q1, e.y0));
}

// hs, hc: sin / cos of half the hinge angle -- what the next step's physics starts from (physics_step)
__device__ __forceinline__ Pose forward_pose(const EnvRegs &e, double &hs, double &hc)
{
    Pose o;
    const double cr = __builtin_fma(e.bq0, e.bq0, -(e.bq3 * e.bq3));
    const double sr = 2.0 * (e.bq0 * e.bq3);
    o.px = __builtin_fma(cr, e.q0, __builtin_fma(-sr, e.q1, e.x0));
    o.py = __builtin_fma(sr, e.q0, __builtin_fma(cr, e.q1, e.y0));
    o.vx = __builtin_fma(cr, e.v0, -(sr * e.v1));
    o.vy = __builtin_fma(sr, e.v0, cr * e.v1);
    o.w = e.v2;
    det_sincos_inl(0.5 * e.q2, hs, hc);
    o.xq0 = __builtin_fma(e.bq0, hc, -(e.bq3 * hs));
    o.xq3 = __builtin_fma(e.bq0, hs, e.bq3 * hc);
    return o;
}

// ZoneEnvBase.py:190-192,217-224 -> the 8-float 'obs' of wrappers.py:136-142
__device__ __forceinline__ void emit_obs8(const DevParams &p, const EnvRegs &e, float *o, double &hs, double &hc)
{
    const Pose f = forward_pose(e, hs, hc);
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
__device__ __forceinline__ void emit_obs8(const DevParams &p, const EnvRegs &e, float *o)
{
    double hs, hc;
    emit_obs8(p, e, o, hs, hc);
}

__device__ __forceinline__ void store_obs8(const DevParams &p, int env, const float *o)
{
    float4 *dst = reinterpret_cast<float4 *>(p.obs + (size_t)env * 8);
    dst[0] = make_float4(o[0], o[1], o[2], o[3]);
    dst[1] = make_float4(o[4], o[5], o[6], o[7]);
}

// ---------------------------------------------------------------------------- zone rows
// One zone of one env as it sits in LDS: (x/3, y/3, code, aux), 16 bytes.
//   TSP / TimedTSP: code = visited ? 1 : 0, aux = the time feature (TimedTSP);
//   ColourMatch:    code = colour 0 Blue / 1 Green / 2 Red, aux = cooldown / max_cd;
//   code < 0:       an all-zero row (finished env stepped without auto-reset).
// expand_entry() turns it into the reference's row (TSP_env.py:31-35, TTSP_env.py:86-92,
// colour_match_env.py:75-80; colours ZoneEnvBase.py:68-77).
template <int TASK>
__device__ __forceinline__ float4 make_entry(const DevParams &p, double zx, double zy, int code, int aux, int k)
{
    float4 en;
    en.x = (float)div_const(zx, 3.0, p.inv3);
    en.y = (float)div_const(zy, 3.0, p.inv3);
    en.z = (float)code;
    en.w = 0.f;
    if (TASK == ZENV_TASK_COLOUR_MATCH) en.w = (float)div_const((double)(float)aux, p.d_maxcd, p.inv_maxcd);
    if (TASK == ZENV_TASK_TIMED_TSP)
        en.w = code ? 1.f : (float)div_const((double)(aux - k), p.d_steps, p.inv_steps);
    return en;
}

// DEFER (k_step_lane, ColourMatch): the entry's aux slot holds the cooldown COUNT and the division by max_cd (den, inv =
// RN(1 / den)) happens here, in the flush, where the wave otherwise waits for its stores, instead of once per zone on
// the zone wave's way to the flush (same-box A/B, ColourMatch-6: 9.32 -> 9.15 us per launch; the same move for
// TimedTSP's (tmax - k) / num_steps changed nothing -- 25 divisions per lane either way -- and is not made).
template <int TASK, bool DEFER = false>
__device__ __forceinline__ void expand_entry(const float4 en, float *row, double den = 1.0, double inv = 1.0)
{
    const bool zero = en.z < 0.f;
    if (TASK == ZENV_TASK_COLOUR_MATCH) {
        row[0] = en.x;
        row[1] = en.y;
        row[2] = en.z == 2.f ? 1.f : 0.f;
        row[3] = en.z == 1.f ? 1.f : 0.f;
        row[4] = en.z == 0.f ? 1.f : 0.f;
        row[5] = zero ? 0.f : 0.25f;
        row[6] = DEFER ? (float)div_const((double)en.w, den, inv) : en.w;
    } else {
        row[0] = en.x;
        row[1] = en.y;
        row[2] = en.z > 0.f ? 1.f : 0.f;
        row[3] = zero ? 0.f : 1.f;
        row[4] = en.z == 0.f ? 1.f : 0.f;
        row[5] = zero ? 0.f : 0.25f;
        if (TASK == ZENV_TASK_TIMED_TSP) row[6] = en.w;
    }
}

// where reset_env / the zone pass put a zone: the LDS entry array, or expanded rows in HBM
struct EntrySink {
    float4 *ents;   // this env's [Z] entries
    template <int TASK>
    __device__ __forceinline__ void put(int z, const float4 en) const { ents[z] = en; }
};
struct GlobalRowSink {
    float *rows;    // this env's [Z][F] rows
    template <int TASK>
    __device__ __forceinline__ void put(int z, const float4 en) const
    {
        float r[7];
        expand_entry<TASK>(en, r);
        for (int f = 0; f < TaskTraits<TASK>::F; ++f) rows[z * TaskTraits<TASK>::F + f] = r[f];
    }
};

// Derived bank rows (k_bank_derive): [slot][3] float4 = first obs (8 floats), first greedy action (2 floats), pad.
template <typename P>
__device__ __forceinline__ float4 load_bank_first(const P &p, int slot, float *of)
{
    const float4 *bf = p.bank_first + 3 * (size_t)slot;     // same address in every lane
    const float4 f0 = bf[0], f1 = bf[1], f2 = bf[2];
    of[0] = f0.x; of[1] = f0.y; of[2] = f0.z; of[3] = f0.w;
    of[4] = f1.x; of[5] = f1.y; of[6] = f1.z; of[7] = f1.w;
    return f2;
}

// Engine.reset for one env from bank slot `slot`: writes the zone arrays, the lane's registers
// and the env's zone rows / entries.
template <int TASK, typename Sink>
__device__ __forceinline__ void reset_env(const DevParams &p, int env, int slot, EnvRegs &e, const Sink &sink)
{
    const int Z = p.Z, N = p.N;
    const double *br = p.bank_robot + 4 * (size_t)slot;
    e.x0 = br[0]; e.y0 = br[1]; e.bq0 = br[2]; e.bq3 = br[3];
    e.q0 = e.q1 = e.q2 = 0.0;
    e.v0 = e.v1 = e.v2 = 0.0;
    e.vis = (TASK == ZENV_TASK_COLOUR_MATCH) ? 0u : p.vis0;   // TSPHardEnv: some zones start visited
    e.colpack = 0ull;
    e.goal_dist = 0;
    e.steps = 0;
    const double *bz = p.bank_zone + 2 * (size_t)slot * Z;
    const int32_t *ba = p.bank_aux + (size_t)slot * Z;
    float4 pair = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int z = 0; z < Z; ++z) {
        const double zx = bz[2 * z], zy = bz[2 * z + 1];
        const size_t zi = (size_t)z * N + env;
        p.zxy[zi] = make_double2(zx, zy);
        int code = (TASK == ZENV_TASK_COLOUR_MATCH) ? 0 : (int)((p.vis0 >> z) & 1u), aux = 0;
        if (TASK == ZENV_TASK_TIMED_TSP) {
            aux = ba[z];
            p.tmax[zi] = aux;
        } else if (TASK == ZENV_TASK_COLOUR_MATCH) {
            code = ba[z];
            e.colpack |= (uint64_t)code << (2 * z);
            *cd_byte(p, z, env) = 0;
        }
        const float4 en = make_entry<TASK>(p, zx, zy, code, aux, 0);
        sink.template put<TASK>(z, en);
        if (z & 1) { pair.z = en.x; pair.w = en.y; } else { pair = make_float4(en.x, en.y, 0.f, 0.f); }
        if ((z & 1) || z == Z - 1) p.zpf[(size_t)(z >> 1) * N + env] = pair;
    }
    if (TASK == ZENV_TASK_COLOUR_MATCH) e.goal_dist = hamming_to_goal(e.colpack, Z);
    p.seed[env] = p.bank_seed[slot];
}

template <typename P>
__device__ __forceinline__ void store_dyn(const P &p, int env, const EnvRegs &e)
{
    p.qa[env] = make_double2(e.q0, e.q1);
    p.qb[env] = make_double2(e.q2, e.v0);
    p.qc[env] = make_double2(e.v1, e.v2);
}

template <typename P>
__device__ __forceinline__ void store_frame(const P &p, int env, const EnvRegs &e)
{
    p.fa[env] = make_double2(e.x0, e.y0);
    p.fb[env] = make_double2(e.bq0, e.bq3);
}

template <typename P>
__device__ __forceinline__ void store_counters(const P &p, int env, int task, const EnvRegs &e)
{
    p.hota[env].steps = e.steps;
    if (task == ZENV_TASK_COLOUR_MATCH) {
        p.hotc[env].colpack = e.colpack;
        p.hota[env].vis = e.goal_dist;
    } else {
        p.hota[env].vis = e.vis;
    }
}

// The frameskip mj_step calls of one Engine.step for point.xml: 3 dof (slide x, slide y, hinge z), offset COM,
// implicit joint damping, no active constraints.  Operation order is the oracle's
// (oracle/zenv_oracle.c: mj_env_step), token for token:
//   * sin/cos of the hinge angle come from the half-angle pair (hs, hc) the observation's quaternion needs anyway
//     (one det_sincos per env step instead of eleven) and are then TURNED by d = h*omega after every substep with
//     short Taylor kernels of sin d / cos d (|d| < 0.05 by config validation; 0.01 for point.xml);
//   * (M + h diag(b)) qacc = qfrc is solved by the Schur complement on the hinge row, which for equal slide damping
//     (p.iso) does not depend on the angle: a multiplication by p.inv_den replaces the division;
//   * the controls are finite here (a NaN action takes the exception path before the physics), so the force clamps
//     are v_min_f64 / v_max_f64 -- same values as the oracle's compare-and-select for every finite input.
__device__ __forceinline__ double clamp_sym(double x, double lim)
{
    return __builtin_fmax(__builtin_fmin(x, lim), -lim);
}

template <bool ISO>
__device__ __forceinline__ void physics_substep(const DevParams &p, EnvRegs &e, double gf0, double kvc1, double &s,
                                                double &k)
{
    const double mcs = p.mc * s, mck = p.mc * k;
    const double w2 = e.v2 * e.v2;
    const double f1 = clamp_sym(__builtin_fma(-p.kvg, e.v2, kvc1), p.fmax);
    const double rhs0 = __builtin_fma(-p.b0, e.v0, __builtin_fma(mck, w2, gf0 * k));
    const double rhs1 = __builtin_fma(-p.b1, e.v1, __builtin_fma(mcs, w2, gf0 * s));
    const double rhs2 = __builtin_fma(-p.b2, e.v2, p.gear * f1);
    const double t0 = rhs0 * p.inv00, t1 = rhs1 * p.inv11;
    const double num = __builtin_fma(-mck, t1, __builtin_fma(mcs, t0, rhs2));
    double a2;
    if (ISO) {
        a2 = num * p.inv_den;
    } else {
        const double den = __builtin_fma(-(mck * mck), p.inv11, __builtin_fma(-(mcs * mcs), p.inv00, p.A22));
        a2 = num / den;
    }
    const double a0 = __builtin_fma(mcs, a2, rhs0) * p.inv00;
    const double a1 = __builtin_fma(-mck, a2, rhs1) * p.inv11;
    e.v0 = __builtin_fma(p.h, a0, e.v0);
    e.v1 = __builtin_fma(p.h, a1, e.v1);
    e.v2 = __builtin_fma(p.h, a2, e.v2);
    e.q0 = __builtin_fma(p.h, e.v0, e.q0);
    e.q1 = __builtin_fma(p.h, e.v1, e.q1);
    e.q2 = __builtin_fma(p.h, e.v2, e.q2);
    // (s, k) <- sin / cos(theta + d), d = h * omega
    const double d = p.h * e.v2;
    const double z = d * d;
    double ps = __builtin_fma(z, 1.0 / 362880.0, -1.0 / 5040.0);
    ps = __builtin_fma(z, ps, 1.0 / 120.0);
    ps = __builtin_fma(z, ps, -1.0 / 6.0);
    const double sd = __builtin_fma(d, z * ps, d);
    double pc = __builtin_fma(z, -1.0 / 3628800.0, 1.0 / 40320.0);
    pc = __builtin_fma(z, pc, -1.0 / 720.0);
    pc = __builtin_fma(z, pc, 1.0 / 24.0);
    pc = __builtin_fma(z, pc, -0.5);
    const double cd = __builtin_fma(z, pc, 1.0);
    const double s2 = __builtin_fma(k, sd, s * cd);
    const double k2 = __builtin_fma(-s, sd, k * cd);
    s = s2;
    k = k2;
}

// c0, c1: the clipped controls; hs, hc: sin / cos of half the hinge angle on entry
// straight10: frameskip == 10, the reference's value -- ten substeps as straight-line code (a runtime trip count costs
// the unrolled loop its remainder loop and a set of register copies between the two)
__device__ __forceinline__ void physics_step(const DevParams &p, EnvRegs &e, double c0, double c1, double hs, double hc,
                                             bool straight10)
{
    double s = 2.0 * (hs * hc);
    double k = __builtin_fma(hc, hc, -(hs * hs));
    const double gf0 = p.gear * clamp_sym(c0, p.fmax);
    const double kvc1 = p.kv * c1;
    // one wave-uniform branch per env step, not per substep
    if (p.iso && straight10) {
#pragma unroll
        for (int i = 0; i < ZENV_SUBSTEP_UNROLL; ++i) physics_substep<true>(p, e, gf0, kvc1, s, k);
    } else if (p.iso) {
#pragma unroll 1
        for (int i = 0; i < p.frameskip; ++i) physics_substep<true>(p, e, gf0, kvc1, s, k);
    } else {
#pragma unroll 1
        for (int i = 0; i < p.frameskip; ++i) physics_substep<false>(p, e, gf0, kvc1, s, k);
    }
}

__device__ __forceinline__ void wave_lds_fence()
{
    // Cross-lane hand-off through LDS inside ONE wave.  The LDS executes a wave's instructions
    // in order, so program order is all that is needed: stop the compiler from moving LDS
    // accesses across this point, and emit no s_waitcnt (a fence builtin would also drain the
    // wave's outstanding global stores here, once per flush iteration).
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}

// Expand the wave's n_rows compact entries into their F floats and stream them to `dst`
// (the tile's contiguous [n_rows][F] block of zone_obs) as dwordx4 bursts.  Each lane
// expands RPC consecutive rows = G whole float4 into the wave-private staging slab, which
// is then read back lane-linear so that one store instruction covers 1 KiB of contiguous HBM.
// Cache policy of the tile stores (buffer_store aux bits: 1 = sc0, 2 = nt, 16 = sc1).
// zone_obs is write-once, read-by-nobody-in-this-launch streaming output (39 MB per step at N = 65 536, Z = 25: more
// than the aggregate L2).  Which policy is fastest depends on the kernel and on the size of a tile's row block
// (round 2, same box, us per step -- scripts/exp_variant.py -DZENV_STORE_AUX=<bits>):
//                          plain (0)   nt (2)   sc1 (16)
//   K1p PointTSP-25          5.84       6.38      5.26      sc1 = write-through, the line leaves the XCD's L2
//   K1p TimedTSP-25          7.14       7.77      6.43      (MI355X_MICROARCH.md, stores of each flavour): the
//   K1p PointTSP-15          3.83       3.99      3.79      row stream no longer competes with its own write-back
//   K1p ColourMatch-6        2.83       3.01      2.97      small blocks are best left to the L2
//   K1  PointTSP-25         15.27      13.73     13.23
//   K1  TimedTSP-25         20.60      18.71     17.97
//   K1  ColourMatch-6       12.72      11.98     12.13
// ZENV_STORE_AUX (diagnostic builds) overrides the choice everywhere.
template <int ZF>
struct StorePolicy {
#ifdef ZENV_STORE_AUX
    static constexpr int kRollout = ZENV_STORE_AUX, kStep = ZENV_STORE_AUX;
#else
    static constexpr int kRollout = ZF >= 80 ? 16 : 0;     // persistent kernel's stream wave
    static constexpr int kStep = ZF >= 100 ? 16 : 2;       // per-step kernel (ZF = 0: zone count known at run time only)
#endif
};
typedef float v4f_t __attribute__((ext_vector_type(4)));

template <int TASK, int AUX, bool DEFER = false>
__device__ __forceinline__ void flush_entries(const float4 *ents, float4 *stage, float *dst, int n_rows,
                                              int lane, double den = 1.0, double inv = 1.0)
{
    constexpr int F = TaskTraits<TASK>::F, RPC = TaskTraits<TASK>::RPC, G = TaskTraits<TASK>::G;
    const int n_chunks = n_rows / RPC;
    // dst is wave-uniform: one buffer descriptor for the tile's [n_rows][F] block
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(dst, 0, n_rows * F * (int)sizeof(float), 0x00020000);
    // full iterations (64 chunks = 64*G float4, no guards), software-pipelined: while the staged rows of
    // iteration i make their round trip through the slab, the entries of iteration i+1 are read and expanded
    // (the LDS executes one wave's accesses in order, so the slab write of i+1 cannot overtake the slab read
    // of i; the fences only stop the compiler)
    int c0 = 0;
    const int n_full = n_chunks / kWave;
    float v[RPC * F];
    if (n_full > 0) {
#pragma unroll
        for (int j = 0; j < RPC; ++j) expand_entry<TASK, DEFER>(ents[lane * RPC + j], v + j * F, den, inv);
    }
    for (int it = 0; it < n_full; ++it, c0 += kWave) {
#pragma unroll
        for (int g = 0; g < G; ++g)
            stage[lane * G + g] = make_float4(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
        wave_lds_fence();
        float4 t[G];
#pragma unroll
        for (int g = 0; g < G; ++g) t[g] = stage[g * kWave + lane];
        wave_lds_fence();
        if (it + 1 < n_full) {
            const int c = c0 + kWave + lane;
#pragma unroll
            for (int j = 0; j < RPC; ++j) expand_entry<TASK, DEFER>(ents[c * RPC + j], v + j * F, den, inv);
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const v4f_t val = { t[g].x, t[g].y, t[g].z, t[g].w };
            __builtin_amdgcn_raw_buffer_store_b128(val, rsrc, (c0 * G + g * kWave + lane) * 16, 0, AUX);
        }
    }
    // last, partial iteration
    if (c0 < n_chunks) {
        const int c = c0 + lane;
        if (c < n_chunks) {
            float v[RPC * F];
#pragma unroll
            for (int j = 0; j < RPC; ++j) expand_entry<TASK, DEFER>(ents[c * RPC + j], v + j * F, den, inv);
#pragma unroll
            for (int g = 0; g < G; ++g)
                stage[lane * G + g] = make_float4(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
        }
        wave_lds_fence();
        const int n4 = (n_chunks - c0) * G;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int i = g * kWave + lane;
            if (i < n4) {
                const float4 tv = stage[i];
                const v4f_t val = { tv.x, tv.y, tv.z, tv.w };
                __builtin_amdgcn_raw_buffer_store_b128(val, rsrc, (c0 * G + i) * 16, 0, AUX);
            }
        }
        wave_lds_fence();
    }
    // ragged tail (n_rows not a multiple of RPC): one row per lane, scalar stores
    const int r = n_chunks * RPC + lane;
    if (r < n_rows) {
        float v[F];
        expand_entry<TASK, DEFER>(ents[r], v, den, inv);
        for (int f = 0; f < F; ++f) dst[(size_t)r * F + f] = v[f];
    }
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

// What the greedy policy reads of a zone row: (x/3, y/3, colour or visited flag, aux column)
struct ZoneView {
    float x, y;
    int code;     // TSP/Timed: 1 = visited; ColourMatch: colour 0/1/2
    float aux;    // row[6]
};
template <int TASK>
struct FullRows {      // the reference layout [Z][F] (global zone_obs or an LDS copy of it)
    const float *rows;
    __device__ __forceinline__ ZoneView get(int z) const
    {
        constexpr int F = TaskTraits<TASK>::F;
        const float *row = rows + z * F;
        ZoneView v;
        v.x = row[0];
        v.y = row[1];
        if (TASK == ZENV_TASK_COLOUR_MATCH) v.code = row[4] != 0.f ? 0 : (row[3] != 0.f ? 1 : 2);
        else v.code = row[2] == 0.f ? 0 : 1;
        v.aux = F == 7 ? row[F - 1] : 0.f;
        return v;
    }
};
template <int TASK>
struct EntryRows {     // compact LDS entries
    const float4 *ents;
    __device__ __forceinline__ ZoneView get(int z) const
    {
        const float4 en = ents[z];
        ZoneView v;
        v.x = en.x;
        v.y = en.y;
        // an all-zero row (code < 0) reads like the reference row of zeros: TSP "unvisited",
        // ColourMatch "Red" (row[4] == row[3] == 0)
        if (TASK == ZENV_TASK_COLOUR_MATCH) v.code = en.z < 0.f ? 2 : (int)en.z;
        else v.code = en.z > 0.f ? 1 : 0;
        v.aux = en.w;
        return v;
    }
};

// Steer towards the nearest eligible zone; (opx, opy, ohx, ohy) = obs[1..4].
template <int TASK, int ZT, typename Rows>
__device__ __forceinline__ float2 greedy_action(const Rows rows, int Zrt, float opx, float opy, float ohx,
                                                float ohy)
{
    const int Z = ZT > 0 ? ZT : Zrt;
    const double px = 3.0 * (double)opx, py = 3.0 * (double)opy;
    const double hx = (double)ohx, hy = (double)ohy;
    int target_colour = -1;
    if (TASK == ZENV_TASK_COLOUR_MATCH) {
        int cb = 0, cg = 0, cr = 0;
#pragma unroll
        for (int z = 0; z < Z; ++z) {
            const int col = rows.get(z).code;
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
            const ZoneView v = rows.get(z);
            const bool eligible = (TASK == ZENV_TASK_COLOUR_MATCH) ? (v.aux == 0.f && v.code != target_colour)
                                                                    : (v.code == 0);
            const double dx = __builtin_fma(3.0, (double)v.x, -px), dy = __builtin_fma(3.0, (double)v.y, -py);
            const double d2 = __builtin_fma(dx, dx, dy * dy);
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
            const ZoneView v = rows.get(best);
            bdx = __builtin_fma(3.0, (double)v.x, -px);
            bdy = __builtin_fma(3.0, (double)v.y, -py);
            bd2 = __builtin_fma(bdx, bdx, bdy * bdy);
        }
    } else {
        for (int z = 0; z < Z; ++z) {
            const ZoneView v = rows.get(z);
            const bool eligible = (TASK == ZENV_TASK_COLOUR_MATCH) ? (v.aux == 0.f && v.code != target_colour)
                                                                    : (v.code == 0);
            const double dx = __builtin_fma(3.0, (double)v.x, -px), dy = __builtin_fma(3.0, (double)v.y, -py);
            const double d2 = __builtin_fma(dx, dx, dy * dy);
            if (eligible && (best < 0 || d2 < bd2)) { best = z; bd2 = d2; bdx = dx; bdy = dy; }
        }
    }
    float2 a = make_float2(0.f, 0.f);
    if (best >= 0 && bd2 > 1e-18) {
        const double n = sqrt(bd2);
        const double cs = __builtin_fma(hx, bdx, hy * bdy) / n;
        const double sn = __builtin_fma(hx, bdy, -(hy * bdx)) / n;
        if (cs < 0.0) a.y = sn >= 0.0 ? 1.f : -1.f;
        else a.y = (float)det_clamp(4.0 * sn, -1.0, 1.0);
        a.x = cs > 0.8 ? 1.f : 0.f;
    }
    return a;
}

// The same policy evaluated by a whole wave for ONE env: lane z holds zone z (lanes >= Z
// idle), a butterfly arg-min replaces the Z-long scan.  Used on the auto-reset path, where a
// single lane would otherwise run the full scan on the zone wave's critical path.  Every lane
// returns the action.  Ties go to the lowest zone index, like the sequential scan.
template <int TASK>
__device__ __forceinline__ float2 greedy_action_coop(int lane, int Z, float zx3, float zy3, int code, float aux,
                                                     float opx, float opy, float ohx, float ohy)
{
    const bool live = lane < Z;
    const double px = 3.0 * (double)opx, py = 3.0 * (double)opy;
    const double hx = (double)ohx, hy = (double)ohy;
    bool eligible = live && code == 0;
    if (TASK == ZENV_TASK_COLOUR_MATCH) {
        const int cb = __popcll(__ballot(live && code == 0));
        const int cg = __popcll(__ballot(live && code == 1));
        const int cr = __popcll(__ballot(live && code == 2));
        int target = 0, best_cnt = cb;
        if (cg > best_cnt) { target = 1; best_cnt = cg; }
        if (cr > best_cnt) { target = 2; }
        eligible = live && aux == 0.f && code != target;
    }
    const double dx = __builtin_fma(3.0, (double)zx3, -px), dy = __builtin_fma(3.0, (double)zy3, -py);
    double d2 = eligible ? __builtin_fma(dx, dx, dy * dy) : __builtin_inf();
    int idx = lane;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const double d2o = __shfl_xor(d2, off);
        const int io = __shfl_xor(idx, off);
        const bool take = d2o < d2 || (d2o == d2 && io < idx);
        d2 = take ? d2o : d2;
        idx = take ? io : idx;
    }
    float2 a = make_float2(0.f, 0.f);
    const double bdx = __shfl(dx, idx), bdy = __shfl(dy, idx);
    const double bd2 = __builtin_fma(bdx, bdx, bdy * bdy);
    if (d2 < __builtin_inf() && bd2 > 1e-18) {
        const double n = sqrt(bd2);
        const double cs = __builtin_fma(hx, bdx, hy * bdy) / n;
        const double sn = __builtin_fma(hx, bdy, -(hy * bdx)) / n;
        if (cs < 0.0) a.y = sn >= 0.0 ? 1.f : -1.f;
        else a.y = (float)det_clamp(4.0 * sn, -1.0, 1.0);
        a.x = cs > 0.8 ? 1.f : 0.f;
    }
    return a;
}

template <int TASK, int ZT>
__device__ __forceinline__ float2 scripted_action(const StepPolicy &pol, int env, const float4 *ents, int Z,
                                                  const float *o)
{
    if (pol.policy == ZENV_POLICY_UNIFORM)
        return uniform_action(pol.env_index0 + (uint64_t)env, pol.step_index, pol.seed);
    return greedy_action<TASK, ZT>(EntryRows<TASK>{ ents }, Z, o[1], o[2], o[3], o[4]);
}

template <int TASK>
__global__ __launch_bounds__(kWave) void k_policy_lane(DevParams p, StepPolicy pol)
{
    extern __shared__ __align__(16) float tile[];
    constexpr int F = TaskTraits<TASK>::F;
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
    reinterpret_cast<float2 *>(pol.out)[env] =
        greedy_action<TASK, 0>(FullRows<TASK>{ tile + lane * ZF }, Z, oa.y, oa.z, oa.w, obb.x);
}

// =========================================================================== K1: step
// A 128-thread workgroup (two wave64) owns a tile of 64 consecutive envs, lane i <-> env i in
// BOTH waves, with the work split by what it is bound by:
//   wave 0, "zone wave"   (memory):  streams the zone-major zone array, runs set_mocaps(),
//       reward / goal / termination / auto-reset (none of which needs this step's physics:
//       set_mocaps() sees the PRE-physics pose), leaves one 16-byte entry per zone in LDS and,
//       after the rendezvous, expands + flushes the 64 x Z x F float32 tile as dwordx4 bursts;
//   wave 1, "physics wave" (latency): the 10 serial MuJoCo substeps, the 8-float obs and, when
//       asked, the scripted action of the NEXT step from that obs and the LDS entries.
// The two meet once (one s_barrier): the zone wave tells the physics wave per env whether its
// result is observable (mode 0) or superseded by a reset / masked no-op (mode 1) and what the
// new step count is.  On different SIMDs the serial fp64 chain of the physics wave runs
// underneath the zone wave's load -> LDS -> store stream instead of after it.
//
// ZT > 0: zone count known at compile time -> the zone loop is fully unrolled and every zone
// load of the wave (Z x 1 KiB, + Z x 256 B tmax / Z x 64 B cooldown) is issued up front, in
// consumption order.  ZT == 0: generic runtime-Z fallback with the loads inside the loop.
constexpr int kStepThreads = 2 * kWave;

// amdgpu_waves_per_eu(2, 2): the grid needs exactly two waves per SIMD (4 tiles x 2 waves per
// CU), so the scheduler may spend up to 256 VGPRs to keep every zone load in flight at once
// instead of sinking loads to save registers for an occupancy the launch never uses.
template <int TASK, int ZT>
__global__ __launch_bounds__(kStepThreads) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_step_lane(DevParams p, const float *__restrict__ actions, int auto_reset, StepPolicy pol)
{
    extern __shared__ __align__(16) float4 lds4[];
    warm_kernarg<sizeof(DevParams) + 8 + 8 + sizeof(StepPolicy)>();
    constexpr int F = TaskTraits<TASK>::F;
    constexpr int G = TaskTraits<TASK>::G;
    constexpr int ZR = ZT > 0 ? ZT : 1;
    constexpr int ZH = ZT > 0 ? (ZT + 1) / 2 : 1;
    const int lane = threadIdx.x & (kWave - 1);
    const int role = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform
    const int env0 = blockIdx.x * kWave;
    const int env = env0 + lane;
    const int N = p.N;
    const int Z = ZT > 0 ? ZT : p.Z;
    float4 *ents = lds4;                                  // [64][Z] compact entries
    float4 *stage = lds4 + kWave * Z;                     // [64*G]  flush staging slab
    int *xmode = reinterpret_cast<int *>(stage + kWave * G);   // zone wave -> physics wave
    int *xstep = xmode + kWave;
    double2 *xpose = reinterpret_cast<double2 *>(xstep + kWave);   // physics wave -> zone wave: pre-physics world position
    int *xexc = reinterpret_cast<int *>(xpose + kWave);            //                           and "the action holds a NaN"
    float4 *my_ents = ents + lane * Z;
    constexpr bool kDefer = TASK == ZENV_TASK_COLOUR_MATCH;   // cooldown / max_cd is left to the flush (expand_entry)

    EnvRegs e;
    e.steps = 0;
    // zone wave: what its epilogue (behind the flush) stores
    bool z_valid = false, z_was_done = false, z_need_reset = false;
    int z_hint = -1, z_first = -1;
    double z_ep_ret = 0.0;
    float z_rew = 0.f;
    uint8_t z_done = 0, z_goal = 0;
    float o[8];   // physics wave: this step's obs
    int pf[11] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };   // physics wave: reset prefetch, one dword per cache line

    if (role == 0) {
        // =================================================================== zone wave
        ZSTAMP(0);
        const bool valid = env < N;
        uint8_t was_done = 0;
        double ep_ret = 0.0;
        float4 zp[ZH];      // float32 zone positions (x/3, y/3), two zones per 16-byte load
        int auxr[ZR];
        constexpr int ZW = ZT > 0 ? (ZT + 7) / 8 : 1;
        uint64_t cdw[ZW];   // ColourMatch: the env's cooldown bytes, eight zones per word
        Sched sc;           // the env's schedule record: loaded whole, stored whole at the end
        bool act_nan = false;
        {
            // ---- issue every load of this env first.  No branch around them: a lane beyond the batch (the last tile
            // of a ragged batch) loads the last env's state and ignores it -- inside an `if (valid)` the loaded values
            // reach the code below through copies at the end of the block, and those copies made the wave wait for
            // EVERY zone load before it could look at the pose (round 3: s_waitcnt vmcnt right behind the issue).
            // The per-env scalars are three 16-byte records (round 4; ten narrow arrays before), and the pose and the
            // action are not loaded here at all: the physics wave has them and hands the pre-physics world position
            // and the action's NaN-ness over through LDS (first barrier below).
            const int envl = valid ? env : N - 1;
            sc = p.sched[envl];
            const HotA ha = p.hota[envl];
            was_done = (uint8_t)sc.done_state;
            e.steps = ha.steps;
            ep_ret = ha.ep_return;
            e.vis = 0u; e.colpack = 0ull; e.goal_dist = 0;
            if (TASK == ZENV_TASK_COLOUR_MATCH) e.goal_dist = (int32_t)ha.vis;
            else e.vis = ha.vis;
            if (TASK == ZENV_TASK_COLOUR_MATCH) {
                const HotC hc = p.hotc[envl];
                e.colpack = hc.colpack;
                cdw[0] = hc.cd0;
            }
            if (ZT > 0) {
                // issue order = consumption order: zone pairs 0, 1, ... so the in-order vmcnt waits of the zone pass
                // release one pair at a time
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int h = 0; h < ZH; ++h) zp[h] = p.zpf[(size_t)h * N + envl];   // 1 KiB per wave
#pragma unroll
                for (int z = 0; z < ZR; ++z) {
                    const size_t zi = (size_t)z * N + envl;
                    auxr[z] = 0;
                    if (TASK == ZENV_TASK_TIMED_TSP) auxr[z] = p.tmax[zi];
                }
                if (TASK == ZENV_TASK_COLOUR_MATCH) {
#pragma unroll
                    for (int w = 1; w < ZW; ++w) cdw[w] = *cd_word(p, w, envl);
                }
                // nothing below may be scheduled above this point, nor any load below it
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        const int nslot = sc.next_slot;   // bank slot of the env's NEXT episode (sequential / ring schedules)
        __syncthreads();          // the physics wave has published the pre-physics pose
        double rx, ry;            // what set_mocaps() sees
        {
            const double2 pz = xpose[lane];
            rx = pz.x; ry = pz.y;
            act_nan = xexc[lane] != 0;
        }

        float rew_out = 0.f;
        uint8_t done_out = 1, goal_out = 0;
        int mode = 1;
        bool need_reset = false;
        bool ends_soon = false;   // may terminate at the NEXT step: its bank rows get prefetched below
        uint32_t near_mask = 0u;  // zones the robot could be inside at the next step (within radius + one step of travel)
        uint32_t soon_elig = 0u;  // ColourMatch: zones whose cooldown will have run out by then
        bool timed_out = false;
        int first = -1;
        const int k = e.steps + 1;   // step index after this call
        const uint32_t full = (Z >= 32) ? 0xFFFFFFFFu : ((1u << Z) - 1u);
        if (valid && was_done) {
            // finished earlier and left alone (step_no_reset): a masked no-op -- WaitWrapper.step's zero obs, reward 0,
            // done (wrappers.py:34-45).  With auto_reset the worker then does what it does after any done step,
            // `obs = env.reset()` (penv.py:8-11): the env comes back with its next episode's first obs.
            if (auto_reset) {
                need_reset = true;
            } else {
                for (int z = 0; z < Z; ++z) my_ents[z] = make_float4(0.f, 0.f, -1.f, 0.f);
                for (int i = 0; i < 8; ++i) o[i] = 0.f;
                store_obs8(p, env, o);
                if (pol.policy >= 0) reinterpret_cast<float2 *>(pol.out)[env] = make_float2(0.f, 0.f);
            }
        } else if (valid) {
            ZSTAMP(1);

            // ---- zone pass: set_mocaps() of the first substep (TSP_env.py:54-69,
            // colour_match_env.py:98-120).  "Inside" is decided per zone from the float32 zone
            // positions the obs rows need anyway: d2f differs from the exact float64 d2 by
            // < 2.1e-7 near the rim (float rounding of x/3, y/3 and of the pose), so outside the
            // shell [d2_lo, d2_hi] (+-8e-7 around r^2) the float verdict IS the exact verdict;
            // inside the shell (about once per 10^7 env-steps) the exact float64 test below
            // decides.  Everything is collected in bit masks; the lowest set bit wins.
            const float rxf = (float)rx, ryf = (float)ry;
            uint32_t in_mask = 0u, out_mask = 0u, amb_mask = 0u, elig_mask = 0u, expired = 0u, expiring = 0u;
            if (TASK == ZENV_TASK_COLOUR_MATCH && ZT > 0) {
#pragma unroll
                for (int w = 0; w < ZW; ++w) cdw[w] = cd_decrement(cdw[w]);   // colour_match_env.py:98-100, all zones
            }
            // The masks are built as in the persistent kernel: d2f >= +0, so its bit pattern orders like its value and
            // "d2f < bound" is the borrow of an integer subtraction; small integers compare the same way; the borrow is
            // shifted into the mask with one v_alignbit.  The zones are visited from the top so that zone z ends up in bit z.
            const uint32_t lo_bits = __float_as_uint(p.d2_lo), hi_bits = __float_as_uint(p.d2_hi);
            const uint32_t near_bits = __float_as_uint(p.d2_near);
            const int zn = ZT > 0 ? ZT : Z;
#pragma unroll
            for (int zr = 0; zr < zn; ++zr) {
                const int z = zn - 1 - zr;
                const size_t zi = (size_t)z * N + env;
                int aux = 0;
                float4 pr;
                if (ZT > 0) {
                    pr = zp[z >> 1];
                    aux = auxr[z];
                    // (the words were decremented above the loop: this is the cooldown AFTER this step's decrement)
                    if (TASK == ZENV_TASK_COLOUR_MATCH) aux = (int)((cdw[z >> 3] >> (8 * (z & 7))) & 0xFFull);
                } else {
                    pr = p.zpf[(size_t)(z >> 1) * N + env];
                    if (TASK == ZENV_TASK_TIMED_TSP) aux = p.tmax[zi];
                    if (TASK == ZENV_TASK_COLOUR_MATCH) aux = *cd_byte(p, z, env);
                }
                const float x3 = (z & 1) ? pr.z : pr.x, y3 = (z & 1) ? pr.w : pr.y;
                const float dxf = __builtin_fmaf(3.f, x3, -rxf), dyf = __builtin_fmaf(3.f, y3, -ryf);
                const uint32_t d2b = __float_as_uint(__builtin_fmaf(dxf, dxf, dyf * dyf));
                in_mask = __builtin_amdgcn_alignbit(in_mask, d2b - lo_bits, 31);         // d2f <  d2_lo: inside for sure
                out_mask = __builtin_amdgcn_alignbit(out_mask, hi_bits - d2b, 31);       // d2f >  d2_hi: outside for sure
                near_mask = __builtin_amdgcn_alignbit(near_mask, d2b - near_bits, 31);   // d2f <  d2_near
                float4 en = make_float4(x3, y3, 0.f, 0.f);
                if (TASK == ZENV_TASK_COLOUR_MATCH) {
                    int cd = aux;
                    if (ZT == 0) {
                        if (cd > 0) cd -= 1;                    // colour_match_env.py:98-100
                        *cd_byte(p, z, env) = (uint8_t)cd;
                    }
                    elig_mask = __builtin_amdgcn_alignbit(elig_mask, (uint32_t)(cd - 1), 31);   // cd == 0 (cd in 0..255)
                    soon_elig = __builtin_amdgcn_alignbit(soon_elig, (uint32_t)(cd - 2), 31);   // cd <= 1: triggerable at the NEXT step
                    en.z = (float)(int)((e.colpack >> (2 * z)) & 3ull);
                    en.w = kDefer ? (float)cd : (float)div_const((double)(float)cd, p.d_maxcd, p.inv_maxcd);
                } else {
                    const bool vis = (e.vis >> z) & 1u;
                    en.z = vis ? 1.f : 0.f;
                    if (TASK == ZENV_TASK_TIMED_TSP) {
                        expired = __builtin_amdgcn_alignbit(expired, (uint32_t)(aux - k - 1), 31);     // tmax - k <= 0, TTSP_env.py:67
                        expiring = __builtin_amdgcn_alignbit(expiring, (uint32_t)(aux - k - 2), 31);   // tmax - k <= 1
                        en.w = vis ? 1.f : (float)div_const((double)(aux - k), p.d_steps, p.inv_steps);
                    }
                }
                my_ents[z] = en;
            }
            amb_mask = ~(in_mask | out_mask);
            if (amb_mask & full) {
                // the rim: exact float64 test on the float64 zone centres (rare, divergent)
                uint32_t m = amb_mask & full;
                while (m) {
                    const int z = __ffs((int)m) - 1;
                    m &= m - 1;
                    const double2 zz = p.zxy[(size_t)z * N + env];
                    const double dx = zz.x - rx, dy = zz.y - ry;
                    if (dx * dx + dy * dy <= p.hit_d2) in_mask |= 1u << z;
                }
            }
            if (TASK != ZENV_TASK_COLOUR_MATCH) elig_mask = ~e.vis;
            const uint32_t hits = in_mask & elig_mask & full;
            first = hits ? __ffs((int)hits) - 1 : -1;                // lowest index wins, one per step
            if (first >= 0) {
                if (TASK == ZENV_TASK_COLOUR_MATCH) {
                    int col = (int)((e.colpack >> (2 * first)) & 3ull);
                    col = (col == 2) ? 0 : col + 1;                  // Blue->Green->Red->Blue
                    e.colpack = (e.colpack & ~(3ull << (2 * first))) | ((uint64_t)col << (2 * first));
                    soon_elig &= ~(1u << first);                     // its cooldown starts over
                    if (ZT > 0) {
#pragma unroll
                        for (int w = 0; w < ZW; ++w)
                            if ((first >> 3) == w)
                                cdw[w] = (cdw[w] & ~(0xFFull << (8 * (first & 7)))) | ((uint64_t)p.max_cd << (8 * (first & 7)));
                    } else {
                        *cd_byte(p, first, env) = (uint8_t)p.max_cd;
                    }
                    float *slot = reinterpret_cast<float *>(my_ents + first);
                    slot[2] = (float)col;
                    slot[3] = kDefer ? (float)p.max_cd : (float)div_const((double)(float)p.max_cd, p.d_maxcd, p.inv_maxcd);
                } else {
                    e.vis |= 1u << first;
                    float *slot = reinterpret_cast<float *>(my_ents + first);
                    slot[2] = 1.f;
                    if (TASK == ZENV_TASK_TIMED_TSP) slot[3] = 1.f;
                }
            }
            if (TASK == ZENV_TASK_TIMED_TSP) {
                timed_out = (expired & ~e.vis & full) != 0u;
                ends_soon = (expiring & ~e.vis & full) != 0u;
            }
            // (ZT > 0: the cooldown words go back with the env's records, below)
        }

        ZSTAMP(14);
        if (valid && !was_done) {
            // ---- reward / goal / termination (Engine.step order; none of it needs the physics)
            // Engine.step's MujocoException path: MuJoCo cannot simulate a NaN control (np.clip keeps it; mj_checkAcc
            // -> BADQACC -> mujoco-py raises): done, reward_exception, no reward() / goal test; the physics wave
            // leaves the joint state mj_resetData would (zeros).  set_mocaps() above already ran, as in the reference.
            const bool exc = act_nan;
            double r = 0.0;
            bool goal = false;
            bool done = false;
            if (exc) {
                r = p.reward_exc;
                done = true;
            } else if (TASK == ZENV_TASK_COLOUR_MATCH) {
                if (first >= 0) {
                    const int nd = hamming_to_goal(e.colpack, Z);
                    r = (double)(e.goal_dist - nd);
                    e.goal_dist = nd;
                }
                goal = e.goal_dist == 0;
            } else {
                r = first >= 0 ? 1.0 : 0.0;
                goal = (e.vis & full) == full;
            }
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
                p.exception[env] = exc ? 1 : 0;
                if (auto_reset) need_reset = true;
                else sc.done_state = 1;
            }
            // next step ends the episode for sure (time limit / a deadline) or possibly (one zone
            // left; ColourMatch: one cycle from the goal)
            // ... AND the robot is near a zone whose visit could do it: "one zone left" / "within one cycle of the goal"
            // hold for tens to hundreds of steps, and every hinted env costs the next launch eleven cache lines of its
            // bank slot (round 4 PMC: ColourMatch-6 fetched 24 MB per step against 12 MB of state, the rest was this)
            const int open_zones = (TASK == ZENV_TASK_COLOUR_MATCH) ? e.goal_dist : Z - (int)__popc(e.vis);
            // one visit per step moves ColourMatch's distance (cycle steps to the nearest uniform colouring) by one at most
            const uint32_t could_hit = (TASK == ZENV_TASK_COLOUR_MATCH) ? (near_mask & soon_elig & full) : (near_mask & ~e.vis & full);
            ends_soon = !done && (ends_soon || k + 1 >= p.num_steps || (open_zones <= 1 && could_hit != 0u));
        }

        ZSTAMP(6);
        // ---- reset hint, one step ahead.  An env that can terminate at the NEXT step leaves the bank slot of its next
        // episode behind; the physics wave of the next launch -- whose own loads land first and which then has time to
        // spare -- touches that slot's cache lines right at its start, so that they are in this XCD's L2 when the
        // cooperative reset asks for them ~2.5 us later instead of costing it a 1.6 us round trip to HBM at the end
        // of the launch's critical path (round 3 stamps; a prefetch at the END of the previous launch, as rounds 1-2
        // had it, only reaches the Infinity Cache: the L2s are invalidated between launches).
        int hint = -1;
        if (ends_soon && auto_reset && p.sched_mode != SCHED_FIXED_SEEDS) hint = nslot;

        ZSTAMP(15);
        // ---- auto-reset (penv.py:8-11), wave-cooperative: for each finished env of the tile,
        // lane z fetches zone z of the new layout (one 400-byte coalesced burst from the bank
        // instead of Z dependent round trips in one lane) and writes it to the zone array and to
        // the env's LDS entries.  This step's physics result is never observed (mode 1).
        unsigned long long pending = __ballot(need_reset);
        if (pending) {
            ZSTAMP(5);
            int my_slot = 0;
            if (need_reset) {
                // (the schedule record goes back whole at the end of the wave: advance the local copy)
                my_slot = next_bank_slot_rec(p, env, sc);
            }
            while (pending) {
                const int j = __ffsll((long long)pending) - 1;   // wave-uniform
                pending &= pending - 1;
                const int slot = __shfl(my_slot, j);
                const int env_j = env0 + j;
                const double *br = p.bank_robot + 4 * (size_t)slot;
                const double b0 = br[0], b1 = br[1], b2 = br[2], b3 = br[3];   // same address in every lane
                int code = (TASK == ZENV_TASK_COLOUR_MATCH) ? 0 : (int)((p.vis0 >> (lane & 31)) & 1u);   // pre-visited zones
                float4 en = make_float4(0.f, 0.f, 0.f, 0.f);
                if (lane < Z) {
                    const size_t bi = (size_t)slot * Z + lane;
                    const double2 zz = reinterpret_cast<const double2 *>(p.bank_zone)[bi];
                    const size_t zi = (size_t)lane * N + env_j;
                    p.zxy[zi] = zz;
                    int aux = 0;
                    if (TASK == ZENV_TASK_TIMED_TSP) {
                        aux = p.bank_aux[bi];
                        p.tmax[zi] = aux;
                    } else if (TASK == ZENV_TASK_COLOUR_MATCH) {
                        code = p.bank_aux[bi];
                        if (ZT == 0) *cd_byte(p, lane, env_j) = 0;      // (ZT > 0: zeroed in lane j's registers below)
                    }
                    en = make_entry<TASK>(p, zz.x, zz.y, code, aux, 0);
                    if (kDefer) en.w = 0.f;                   // cooldown count 0
                    ents[j * Z + lane] = en;
                }
                {
                    // float32 positions, two zones per float4: even lanes pick up their odd neighbour
                    const float nx = __shfl_down(en.x, 1), ny = __shfl_down(en.y, 1);
                    if (lane < Z && !(lane & 1))
                        p.zpf[(size_t)(lane >> 1) * N + env_j] =
                            make_float4(en.x, en.y, lane + 1 < Z ? nx : 0.f, lane + 1 < Z ? ny : 0.f);
                }
                ZSTAMP(6);
                uint64_t colpack = 0ull;
                if (TASK == ZENV_TASK_COLOUR_MATCH) {
                    const unsigned long long m0 = __ballot(lane < Z && (code & 1));
                    const unsigned long long m1 = __ballot(lane < Z && (code & 2));
                    colpack = spread_even_bits((uint32_t)m0) | (spread_even_bits((uint32_t)m1) << 1);
                }
                EnvRegs fresh;
                fresh.x0 = b0; fresh.y0 = b1; fresh.bq0 = b2; fresh.bq3 = b3;
                fresh.q0 = fresh.q1 = fresh.q2 = 0.0;
                fresh.v0 = fresh.v1 = fresh.v2 = 0.0;
                fresh.vis = (TASK == ZENV_TASK_COLOUR_MATCH) ? 0u : p.vis0;
                fresh.colpack = colpack;
                fresh.goal_dist = (TASK == ZENV_TASK_COLOUR_MATCH) ? hamming_to_goal(colpack, Z) : 0;
                fresh.steps = 0;
                // the first obs of the next episode and the greedy policy's first action: derived once per bank row
                // (k_bank_derive, same arithmetic), so a reset is one round of loads and a handful of stores
                float of[8];
                const float4 f2 = load_bank_first(p, slot, of);
                float2 next_act = make_float2(f2.x, f2.y);
                if (pol.policy == ZENV_POLICY_UNIFORM)
                    next_act = uniform_action(pol.env_index0 + (uint64_t)env_j, pol.step_index, pol.seed);
                if (lane == j) {
                    e = fresh;
                    ep_ret = 0.0;
#pragma unroll
                    for (int w = 0; w < ZW; ++w) cdw[w] = 0ull;
                    mode = 2;   // superseded by the reset (the physics wave keeps the terminal position)
                    p.seed[env] = p.bank_seed[slot];
                    store_frame(p, env, e);
                    store_dyn(p, env, e);
                    store_obs8(p, env, of);
                    if (pol.policy >= 0) reinterpret_cast<float2 *>(pol.out)[env] = next_act;
                }
                ZSTAMP(7);
            }
        }

        if (valid) {
            xmode[lane] = mode;
            xstep[lane] = e.steps;
        }
        // the env's small per-step results and counters go out BEHIND the row stream (after the rendezvous, below): a
        // dozen store instructions that nothing waits for no longer sit between the zone pass and the start of the flush
        z_valid = valid; z_was_done = was_done != 0; z_need_reset = need_reset; z_hint = hint; z_first = first;
        z_ep_ret = ep_ret; z_rew = rew_out; z_done = done_out; z_goal = goal_out;
        // ---- the env's records go back whole: one 16-byte store each
        if (z_valid) {
            if (!z_was_done || auto_reset) {
                p.hota[env] = HotA{ z_ep_ret, e.steps, TASK == ZENV_TASK_COLOUR_MATCH ? (uint32_t)e.goal_dist : e.vis };
                if (TASK == ZENV_TASK_COLOUR_MATCH) {
                    if (ZT > 0) {
                        p.hotc[env] = HotC{ e.colpack, cdw[0] };
#pragma unroll
                        for (int w = 1; w < ZW; ++w) *cd_word(p, w, env) = cdw[w];
                    } else {
                        p.hotc[env].colpack = e.colpack;       // (runtime zone count: the cooldown bytes went back one by one)
                    }
                }
                if (z_was_done) sc.done_state = 0;      // revived by the reset
            }
            sc.reset_hint = z_need_reset ? -1 : z_hint;
            p.sched[env] = sc;
            if (p.visit_zone) p.visit_zone[env] = z_first;
            p.reward[env] = z_rew;
            p.done_out[env] = z_done;
            p.goal_met[env] = z_goal;
        }
        ZSTAMP(2);
    } else {
        // =================================================================== physics wave
        ZSTAMP(8);
        // ---- loads (a lane beyond the batch reads the last env's state and ignores it), then the hand-over the zone
        // wave waits for: the pre-physics world position -- what set_mocaps() sees -- and whether the action holds a NaN
        const int envl = env < N ? env : N - 1;
        float2 act;
        int hint;
        {
            const double2 qa = p.qa[envl], fa = p.fa[envl], fb = p.fb[envl];      // the pose first
            act = reinterpret_cast<const float2 *>(actions)[envl];
            const double2 qb = p.qb[envl], qc = p.qc[envl];
            hint = auto_reset ? p.sched[envl].reset_hint : -1;                     // the previous launch's reset hint
            e.q0 = qa.x; e.q1 = qa.y; e.q2 = qb.x;
            e.v0 = qb.y; e.v1 = qc.x; e.v2 = qc.y;
            e.x0 = fa.x; e.y0 = fa.y; e.bq0 = fb.x; e.bq3 = fb.y;
            double rx, ry;
            world_pos(e, rx, ry);
            xpose[lane] = make_double2(rx, ry);
            xexc[lane] = !(act.x == act.x && act.y == act.y);
        }
        __syncthreads();
        if (env < N) {
            // (see the zone wave: touch every cache line of the bank slot this env may reset into)
            if (hint >= 0 && hint < p.bank_size) {
                const int *bz = reinterpret_cast<const int *>(p.bank_zone + 2 * (size_t)hint * Z);
                pf[0] = bz[0];                                   // 128-B lines of the 16*Z-byte zone block
                if (4 * Z > 32) pf[1] = bz[32];
                if (4 * Z > 64) pf[2] = bz[64];
                if (4 * Z > 96) pf[3] = bz[96];
                pf[4] = bz[4 * Z - 1];
                pf[5] = reinterpret_cast<const int *>(p.bank_robot + 4 * (size_t)hint)[0];
                if (TASK != ZENV_TASK_TSP) {
                    const int *ba = p.bank_aux + (size_t)hint * Z;
                    pf[6] = ba[0];
                    pf[7] = ba[Z - 1];
                }
                pf[8] = (int)p.bank_seed[hint];
                pf[9] = reinterpret_cast<const int *>(p.bank_first + 3 * (size_t)hint)[0];
                pf[10] = reinterpret_cast<const int *>(p.bank_first + 3 * (size_t)hint)[11];
            }
            // Engine.step: ctrl = clip(action, ctrlrange); frameskip x mj_step
            const double c0 = det_clamp((double)act.x, -1.0, 1.0);
            const double c1 = det_clamp((double)act.y, -1.0, 1.0);
            ZSTAMP(9);
            {
                double hs, hc;
                det_sincos_inl(0.5 * e.q2, hs, hc);
                physics_step(p, e, c0, c1, hs, hc, p.frameskip == ZENV_SUBSTEP_UNROLL);
            }
            if (!(c0 == c0 && c1 == c1)) {
                // exception path (see the zone wave): what mj_resetData leaves -- qpos = qpos0, qvel = 0
                e.q0 = e.q1 = e.q2 = 0.0;
                e.v0 = e.v1 = e.v2 = 0.0;
            }
            emit_obs8(p, e, o);   // o[0] (remaining) is patched after the rendezvous
        }
        ZSTAMP(10);
    }

    // The one rendezvous of the two waves: the zone entries + per-env mode / step count are in
    // LDS.  After it the zone wave streams the tile out while the physics wave finishes.
    __syncthreads();
    if (role == 0) ZSTAMP(4); else ZSTAMP(11);

    if (role == 0) {
        const int n_blk = min(kWave, N - env0);
#if !defined(ZENV_EXP) || !(ZENV_EXP & 1)   // diagnostic builds only: ZENV_EXP bit 0 drops the flush
        flush_entries<TASK, StorePolicy<ZT * F>::kStep, kDefer>(ents, stage, p.zone_obs + (size_t)env0 * Z * F, n_blk * Z, lane,
                                                               p.d_maxcd, p.inv_maxcd);
#endif
        ZSTAMP(3);
    } else if (env < N) {
        if (xmode[lane] == 0) {
            e.steps = xstep[lane];
            store_dyn(p, env, e);
            o[0] = (float)(1.0 - div_const((double)e.steps, p.d_steps, p.inv_steps));
            store_obs8(p, env, o);
            ZSTAMP(12);
            // fused K3: the action of the NEXT step, from this step's obs and the entries in LDS
            if (pol.policy >= 0)
                reinterpret_cast<float2 *>(pol.out)[env] = scripted_action<TASK, ZT>(pol, env, my_ents, Z, o);
        } else if (xmode[lane] == 2 && p.term_xy) {
            double tx, ty;
            world_pos(e, tx, ty);           // where the finished episode ended (TSP_next_city_env.py:63-66)
            p.term_xy[env] = make_double2(tx, ty);
        }
        // (nothing of the prefetch is consumed: the registers are only kept alive until the end of the wave)
        asm volatile("" ::"v"(pf[0]), "v"(pf[1]), "v"(pf[2]), "v"(pf[3]), "v"(pf[4]), "v"(pf[5]), "v"(pf[6]),
                     "v"(pf[7]), "v"(pf[8]), "v"(pf[9]), "v"(pf[10]));
        ZSTAMP(13);
    }
}

// =========================================================================== K1p: persistent rollout
// K closed-loop steps a_t = pi(obs_t, t); step(a_t) in ONE launch, for the case the rollout API
// exists for: the action source is on the device (zenv_rollout).  Same arithmetic per step as
// k_step_lane, different split of the tile's two waves:
//   wave 0, "env wave":     lane i owns env i with ALL of its state in registers for the whole
//       launch -- zone positions (ZT/2 float4), deadlines / cooldowns, joint state, counters,
//       the pending action.  Per step: set_mocaps() zone pass on the pre-physics pose,
//       reward / goal / termination, 10 physics substeps, obs, next action -- no LDS or global
//       reads at all (bank rows on a reset, zone centres on a rim hit).  It publishes per env
//       one 8-byte word of per-step zone state (visited mask [+ step count] / packed colours,
//       + cooldown bytes for ColourMatch) into a double-buffered LDS slot.
//   wave 1, "stream wave":  expands the tile's 64 x Z rows from the static LDS entries
//       (x/3, y/3 [, deadline]; rewritten only on a reset) and that per-step word, and streams
//       them to zone_obs as 1 KiB non-temporal bursts.  It runs up to two steps behind the env
//       wave, so the env wave never waits for the store path unless that is the bottleneck.
// Hand-over through two monotonic LDS counters (published(t): env -> stream; flushed(t):
// stream -> env).  Per step the kernel reads nothing and writes what a step must publish: obs,
// zone_obs, reward, done, goal_met (+ visit_count); state goes back to memory when it ends.
// Needs a compile-time zone count (register arrays); zenv_rollout falls back to per-step
// launches otherwise.
// The counters are touched through explicit LDS (address space 3) pointers: through a generic
// pointer the compiler emits FLAT loads, and a FLAT load's s_waitcnt also drains the wave's
// outstanding global stores -- once per spin.
typedef __attribute__((address_space(3))) volatile int lds_vint;
__device__ __forceinline__ void lds_ctr_set(int *c, int v)
{
    asm volatile("" ::: "memory");
    *(lds_vint *)c = v;
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void lds_ctr_wait(const int *c, int v)
{
    // bounded: the producer always gets there; the bound turns a logic error into wrong output
    // instead of a hung GPU
#pragma unroll 1
    for (int spins = 0; spins < (1 << 24); ++spins) {
        if (*(lds_vint *)c >= v) break;
        __builtin_amdgcn_s_sleep(1);
    }
    asm volatile("" ::: "memory");
}

// static part of a zone row in LDS: (x/3, y/3) and, for TimedTSP, the deadline (int bits)
template <int TASK> struct StaticEnt { using type = float2; };
template <> struct StaticEnt<ZENV_TASK_TIMED_TSP> { using type = float4; };
template <int TASK>
__device__ __forceinline__ typename StaticEnt<TASK>::type make_static(float x3, float y3, int aux)
{
    if constexpr (TASK == ZENV_TASK_TIMED_TSP) return make_float4(x3, y3, __int_as_float(aux), 0.f);
    else return make_float2(x3, y3);
}
constexpr uint64_t kDynZero = 1ull << 63;   // the env's rows are all zero (finished, no auto-reset)

// r / ZT for r < 64 * ZT without an integer division (checked exhaustively at compile time)
template <int ZT> struct RowDiv {
    static constexpr uint32_t M = (65536u + ZT - 1) / ZT;
    static constexpr bool ok()
    {
        for (uint32_t r = 0; r < 64u * ZT; ++r)
            if (((r * M) >> 16) != r / ZT) return false;
        return true;
    }
    static_assert(ok(), "multiplier does not reproduce r / ZT");
    static __device__ __forceinline__ int div(int r) { return (int)(__umul24((uint32_t)r, M) >> 16); }
};

// One reference row (TSP_env.py:31-35, TTSP_env.py:86-92, colour_match_env.py:75-80) from
// its static entry + the env's per-step word; same values as make_entry() + expand_entry().
// MAY_ZERO = false: the env wave has told the stream wave that no env of the tile is frozen this
// step (always the case with auto-reset on), so the all-zero-row selects drop out.
// ColourMatch rows in the persistent kernel: the colour's (R, G, B, A) and the cooldown feature cd / max_cd take a
// handful of values, so the stream wave looks them up in two small LDS tables filled once per launch with the same
// expressions (rt.col[colour 0..2, 3 = all-zero row], rt.cd[0..255]) instead of re-deriving them for every row of every
// step (three compare-select pairs and a three-instruction float64 division per row: ~100 of the stream wave's ~200
// vector instructions per step at Z = 6).
struct RowTables {
    const float4 *col;
    const float *cd;
};
template <int TASK, int ZT, bool MAY_ZERO>
__device__ __forceinline__ void expand_row(const DevParams &p, const typename StaticEnt<TASK>::type s, uint64_t d,
                                           const uint8_t *cd_env, int z, float *row, const RowTables &rt)
{
    const bool zero = MAY_ZERO && (d & kDynZero) != 0ull;
    row[0] = zero ? 0.f : s.x;
    row[1] = zero ? 0.f : s.y;
    if constexpr (TASK == ZENV_TASK_COLOUR_MATCH) {
        const int col = zero ? 3 : (int)((d >> (2 * z)) & 3ull);
        const float4 c = rt.col[col];
        row[2] = c.x; row[3] = c.y; row[4] = c.z; row[5] = c.w;
        const int cd = zero ? 0 : (int)cd_env[z];
        row[6] = rt.cd[cd];
    } else {
        row[5] = zero ? 0.f : 0.25f;
        const bool vis = (((uint32_t)d >> z) & 1u) != 0u;
        row[2] = vis ? 1.f : 0.f;
        row[3] = zero ? 0.f : 1.f;
        row[4] = (vis || zero) ? 0.f : 1.f;
        if constexpr (TASK == ZENV_TASK_TIMED_TSP) {
            const int k = (int)(uint32_t)(d >> 32) & 0x7FFFFFFF;
            const int tmax = __float_as_int(s.z);
            const float left = (float)div_const((double)(tmax - k), p.d_steps, p.inv_steps);
            row[6] = zero ? 0.f : (vis ? 1.f : left);
        }
    }
}

// flush_entries() for the persistent kernel: rows come from static entries + per-env words
template <int TASK, int ZT, bool MAY_ZERO>
__device__ __forceinline__ void flush_static(const DevParams &p, const typename StaticEnt<TASK>::type *sent,
                                             const uint64_t *dynw, const uint8_t *cdb, float4 *stage, float *dst,
                                             int n_rows, int lane, const RowTables &rt)
{
    constexpr int F = TaskTraits<TASK>::F, RPC = TaskTraits<TASK>::RPC, G = TaskTraits<TASK>::G;
    constexpr int ZB = (ZT + 3) & ~3;
    constexpr int AUX = StorePolicy<ZT * F>::kRollout;
    const int n_chunks = n_rows / RPC;
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(dst, 0, n_rows * F * (int)sizeof(float), 0x00020000);
    auto expand_chunk = [&](int c, float *v) {
#pragma unroll
        for (int j = 0; j < RPC; ++j) {
            const int r = c * RPC + j;
            const int el = RowDiv<ZT>::div(r);
            expand_row<TASK, ZT, MAY_ZERO>(p, sent[r], dynw[el], cdb + el * ZB, r - el * ZT, v + j * F, rt);
        }
    };
    // Full iterations (64 chunks = 64*G float4 each), software-pipelined: while the staged rows of
    // iteration i make their round trip through the slab, the static entries / step words of
    // iteration i+1 are fetched and expanded.  The LDS executes one wave's accesses in order, so the
    // slab write of i+1 cannot overtake the slab read of i; the fences only stop the compiler.
    int c0 = 0;
    const int n_full = n_chunks / kWave;
    float v[RPC * F];
    if (n_full > 0) expand_chunk(lane, v);
    for (int it = 0; it < n_full; ++it, c0 += kWave) {
#pragma unroll
        for (int g = 0; g < G; ++g)
            stage[lane * G + g] = make_float4(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
        wave_lds_fence();
        float4 t[G];
#pragma unroll
        for (int g = 0; g < G; ++g) t[g] = stage[g * kWave + lane];
        wave_lds_fence();
        if (it + 1 < n_full) expand_chunk(c0 + kWave + lane, v);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const v4f_t val = { t[g].x, t[g].y, t[g].z, t[g].w };
            __builtin_amdgcn_raw_buffer_store_b128(val, rsrc, (c0 * G + g * kWave + lane) * 16, 0, AUX);
        }
    }
    if (c0 < n_chunks) {
        const int c = c0 + lane;
        if (c < n_chunks) {
            float v[RPC * F];
            expand_chunk(c, v);
#pragma unroll
            for (int g = 0; g < G; ++g)
                stage[lane * G + g] = make_float4(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
        }
        wave_lds_fence();
        const int n4 = (n_chunks - c0) * G;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int i = g * kWave + lane;
            if (i < n4) {
                const float4 tv = stage[i];
                const v4f_t val = { tv.x, tv.y, tv.z, tv.w };
                __builtin_amdgcn_raw_buffer_store_b128(val, rsrc, (c0 * G + i) * 16, 0, AUX);
            }
        }
        wave_lds_fence();
    }
    // ragged tail (n_rows not a multiple of RPC): one row per lane, scalar stores
    const int r = n_chunks * RPC + lane;
    if (r < n_rows) {
        float v[F];
        const int el = RowDiv<ZT>::div(r);
        expand_row<TASK, ZT, MAY_ZERO>(p, sent[r], dynw[el], cdb + el * ZB, r - el * ZT, v, rt);
        for (int f = 0; f < F; ++f) dst[(size_t)r * F + f] = v[f];
    }
}

// greedy_action() on the env wave's registers (no memory behind it, so no dynamic zone index):
// same distances, same winner -- the nearest eligible zone, lowest index on ties -- found as
// "minimum value, then the lowest zone that attains it" instead of a (value, index) tournament.
// Ineligible zones carry a finite sentinel >= 2^1023 in place of +inf (only the high word is
// replaced); a NaN pose leaves no zone below the sentinel, like the scan it replaces.
__device__ __forceinline__ double min_f64(double a, double b)
{
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
template <int TASK, int ZT>
__device__ __forceinline__ float2 greedy_action_regs(const float4 *zp, const int *auxr, uint32_t vis, uint64_t colpack,
                                                     float opx, float opy, float ohx, float ohy)
{
    const double px = 3.0 * (double)opx, py = 3.0 * (double)opy;
    const double hx = (double)ohx, hy = (double)ohy;
    int target_colour = -1;
    if (TASK == ZENV_TASK_COLOUR_MATCH) {
        const uint64_t used = (1ull << (2 * ZT)) - 1ull, lo_bits = 0x5555555555555555ull;
        const int cg = __popcll(colpack & lo_bits & used), cr = __popcll((colpack >> 1) & lo_bits & used);
        const int cb = ZT - cg - cr;
        target_colour = 0;
        int best_cnt = cb;
        if (cg > best_cnt) { target_colour = 1; best_cnt = cg; }
        if (cr > best_cnt) { target_colour = 2; }
    }
    // 3.0 from a register: as a literal it only fits v_fmac_f64, whose accumulator has to be loaded with -px by a
    // v_mov_b64 first -- two instructions per coordinate instead of one v_fma_f64
    double three = 3.0;
    asm("" : "+v"(three));
    // two halves of the zone list, one after the other: halves the live distance array
    double mbest = 0.0;
    float bx = 0.f, by = 0.f;
#pragma unroll
    for (int part = 0; part < 2; ++part) {
        constexpr int ZA = (ZT + 1) / 2;
        const int z0 = part == 0 ? 0 : ZA;
        const int zn = part == 0 ? ZA : ZT - ZA;          // zones in this half (compile-time after unrolling)
        double d2s[ZA];
#pragma unroll
        for (int i = 0; i < ZA; ++i) {
            if (i >= zn) { d2s[i] = 0.0; continue; }
            const int z = z0 + i;
            const float4 pr = zp[z >> 1];
            const float x3 = (z & 1) ? pr.z : pr.x, y3 = (z & 1) ? pr.w : pr.y;
            bool eligible;
            if (TASK == ZENV_TASK_COLOUR_MATCH)
                eligible = auxr[z] == 0 && (int)((colpack >> (2 * z)) & 3ull) != target_colour;
            else eligible = ((vis >> z) & 1u) == 0u;
            const double dx = __builtin_fma(three, (double)x3, -px), dy = __builtin_fma(three, (double)y3, -py);
            const double d2 = __builtin_fma(dx, dx, dy * dy);
            d2s[i] = __hiloint2double(eligible ? __double2hiint(d2) : 0x7FE00000, __double2loint(d2));
        }
        constexpr int ZM = (ZA + 1) / 2;
        double m[ZM];
#pragma unroll
        for (int h = 0; h < ZM; ++h) m[h] = (2 * h + 1 < zn) ? min_f64(d2s[2 * h], d2s[2 * h + 1]) : d2s[2 * h < zn ? 2 * h : 0];
#pragma unroll
        for (int stride = 1; stride < ZM; stride *= 2) {
#pragma unroll
            for (int h = 0; h + stride < ZM; h += 2 * stride)
                if (2 * (h + stride) < zn) m[h] = min_f64(m[h], m[h + stride]);
        }
        float hx3 = 0.f, hy3 = 0.f;
#pragma unroll
        for (int i = ZA - 1; i >= 0; --i) {      // descending: the lowest zone at the minimum is written last
            if (i >= zn) continue;
            const int z = z0 + i;
            const float4 pr = zp[z >> 1];
            const bool hit = d2s[i] == m[0];
            hx3 = hit ? ((z & 1) ? pr.z : pr.x) : hx3;
            hy3 = hit ? ((z & 1) ? pr.w : pr.y) : hy3;
        }
        // the second half only wins with a strictly smaller distance (ties go to the lower index)
        const bool take = part == 0 || m[0] < mbest;
        mbest = take ? m[0] : mbest;
        bx = take ? hx3 : bx;
        by = take ? hy3 : by;
    }
    float2 a = make_float2(0.f, 0.f);
    if (mbest < 1e300) {
        const double bdx = __builtin_fma(3.0, (double)bx, -px), bdy = __builtin_fma(3.0, (double)by, -py);
        const double bd2 = __builtin_fma(bdx, bdx, bdy * bdy);
        if (bd2 > 1e-18) {
            const double n = sqrt(bd2);
            const double cs = __builtin_fma(hx, bdx, hy * bdy) / n;
            const double sn = __builtin_fma(hx, bdy, -(hy * bdx)) / n;
            if (cs < 0.0) a.y = sn >= 0.0 ? 1.f : -1.f;
            else a.y = (float)det_clamp(4.0 * sn, -1.0, 1.0);
            a.x = cs > 0.8 ? 1.f : 0.f;
        }
    }
    return a;
}

// EXT = true: the action-chunk form (zenv_step_many) -- the actions of the launch's steps come from a caller-supplied
// buffer io.actions [n_steps][N] instead of a scripted policy (a_{t+1} is requested at the top of step t and waited
// for right behind the physics, before the step's own stores are issued, so the wait never drains them); a NaN action
// takes Engine.step's exception branch as in k_step_lane; every step's reward / done is also recorded time-major in
// io.reward / io.done.  auto_reset: 0 never (step_no_reset: a finished env idles as WaitWrapper's no-op), 1 every step
// (step), 2 only in the launch's LAST step -- skill_len - 1 step_no_reset calls, then one step
// (main/src/torch_ac/algos/_hier_policy_opt.py:68-71, hier_base.py:179-183).
template <int TASK, int ZT, bool EXT>
__global__ __launch_bounds__(2 * kWave) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_rollout_lane(DevParams p, int n_steps, int auto_reset, StepPolicy pol, int tile0, ChunkIO io)
{
    static_assert(ZT > 0 && ZT <= 30, "zone arrays live in registers; bit 31 / 63 of the step word are flags");
    using SE = typename StaticEnt<TASK>::type;
    extern __shared__ __align__(16) float4 lds4[];
    constexpr int F = TaskTraits<TASK>::F;
    constexpr int G = TaskTraits<TASK>::G;
    constexpr int Z = ZT;
    constexpr int ZH = (ZT + 1) / 2;
    constexpr int ZB = (ZT + 3) & ~3;                            // cooldown bytes per env, padded
    constexpr bool kColour = TASK == ZENV_TASK_COLOUR_MATCH;
    // Which wave wins the SIMD's issue slot when both are ready (s_setprio 3 for one of them).  Measured per workload in
    // steady state, same box, us per step env-first / stream-first (round 2, after the store-policy change):
    // PointTSP-25 5.23 / 5.43, TimedTSP-25 6.54 / 6.39, PointTSP-15 3.75 / 3.68, ColourMatch-6 2.82 / 3.04 -- the
    // stream wave goes first where its expansion work per byte is highest (7-float rows with a deadline division per
    // row; mid-size 6-float blocks), the env wave's dependent instruction stream otherwise.
#ifdef ZENV_STREAM_FIRST
    constexpr bool kStreamFirst = ZENV_STREAM_FIRST != 0;
#else
    constexpr bool kStreamFirst = TASK == ZENV_TASK_TIMED_TSP ? ZT * F >= 100 : (ZT * F >= 80 && ZT * F < 120);
#endif
    // Which wave issues the per-env results of a step (obs, reward, done, goal_met, visit count: six store instructions).
    // The waves of a CU share one vector-memory pipeline, and under the row stream's saturation an instruction of the
    // env wave waits at issue behind the stream waves' 1 KiB stores: handed over through LDS and issued by the stream
    // wave, they cost the env wave nothing.  Same-box A/B of the two forms, us per step: ColourMatch-6 2.69 / 2.65,
    // PointTSP-15 3.74 / 3.60, PointTSP-25 5.26 / 5.19, TimedTSP-25 6.47 / 6.40 (env wave / stream wave); in the
    // action-chunk form (EXT), whose env wave would issue ten vector-memory instructions per step, TimedTSP-25 8.6 / 7.5.
#ifdef ZENV_STREAM_STORES
    constexpr bool kStreamStores = ZENV_STREAM_STORES != 0;
#else
    constexpr bool kStreamStores = true;
#endif
    const int lane = threadIdx.x & (kWave - 1);
    const int role = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform
    const int env0 = ((int)blockIdx.x + tile0) * kWave;   // tile0: first tile of this launch's slice of the batch
    const int env = env0 + lane;
    const int N = p.N;
    const bool valid = env < N;
    SE *sent = reinterpret_cast<SE *>(lds4);                     // [64][Z] static entries
    float4 *stage = reinterpret_cast<float4 *>(sent + kWave * Z);        // [64*G] flush staging slab
    uint64_t *dynw = reinterpret_cast<uint64_t *>(stage + kWave * G);    // [2][64] per-step words
    uint32_t *cdw = reinterpret_cast<uint32_t *>(dynw + 2 * kWave);      // [2][64][ZB/4] cooldown bytes
    // [0] published, [1] flushed, [2 + b] "step word buffer b holds an all-zero env"
    int *ctr = reinterpret_cast<int *>(cdw + (kColour ? 2 * kWave * (ZB / 4) : 0));
    float4 *coltab = reinterpret_cast<float4 *>(ctr + 4);               // ColourMatch: [4] (R, G, B, A) by colour code
    float *cdtab = reinterpret_cast<float *>(coltab + 4);                // ColourMatch: [256] cooldown / max_cd
    // The env wave's per-env results of a step, handed to the stream wave like the step words (double-buffered): the
    // 8-float obs as the tile's [128] float4 (its 2 KiB of p.obs, in order) and one float4 (reward, flags, visit count, -)
    float4 *obs_pub = reinterpret_cast<float4 *>(kColour ? reinterpret_cast<char *>(cdtab + 256) : reinterpret_cast<char *>(ctr + 4));   // [2][128]
    float4 *misc_pub = obs_pub + 2 * 2 * kWave;                                                                                         // [2][64]
    float *act_touch = reinterpret_cast<float *>(misc_pub + 2 * kWave);      // EXT: [64] landing pad of the action prefetch (never read)
    const RowTables rt{ coltab, cdtab };
    if (kColour) {                                                       // both waves fill, before the launch's barrier
        for (int i = threadIdx.x; i < 256; i += 2 * kWave)
            cdtab[i] = (float)div_const((double)(float)i, p.d_maxcd, p.inv_maxcd);
        if (threadIdx.x < 4) {
            const int c = threadIdx.x;                                   // 0 Blue, 1 Green, 2 Red, 3 = an all-zero row
            coltab[c] = make_float4(c == 2 ? 1.f : 0.f, c == 1 ? 1.f : 0.f, c == 0 ? 1.f : 0.f, c == 3 ? 0.f : 0.25f);
        }
    }
    const uint32_t full = (1u << Z) - 1u;
    const int n_blk = min(kWave, N - env0);
    const int n_rows = n_blk * Z;
    float *tile_dst = p.zone_obs + (size_t)env0 * Z * F;

    if (role == 1) {
        // =================================================================== stream wave
        if (kStreamFirst) __builtin_amdgcn_s_setprio(3);
        __syncthreads();   // static entries of the first step + cleared counters
        // EXT: the caller's action rows are read ONCE, from memory, beside a store stream that saturates the memory system.
        // Left to the env wave alone (one 512 B row per tile and step, issued a step ahead) that read costs 0.6-1.3 us per
        // step although it is 1.6 % of the bytes: the same kernel fetching a cache-resident row runs at the scripted
        // kernel's speed, and a prefetch of ONE row per step, at any distance from 3 to 16 steps, recovers a quarter of
        // it -- so it is not latency.  What recovers all of it is reading rarely and much: this wave, which never waits
        // for a load, pulls kActBatch rows of the tile into the XCD's L2 every kActBatch steps (PointTSP-25, 2048 fresh
        // rows: 6.37 us per step without, 6.16 one row per step, 5.37 eight rows every eighth step; scripted greedy
        // 5.55 -- profiles/r04/action_prefetch_sweep.log), i.e. the memory controllers turn from writing to reading an
        // eighth as often.  The touch is an LDS-DMA into a pad nobody reads: no destination register, no wait; the env
        // wave's own load of the row then hits L2.  It is written as assembly on purpose: the compiler orders every later
        // LDS read behind a DMA it knows about with s_waitcnt vmcnt(0), which here would drain this wave's whole store
        // queue once per step.
#if defined(ZENV_ACT_AHEAD)               // diagnostic builds: the prefetch distance (0 = no prefetch) and batch
        constexpr int kActAhead = ZENV_ACT_AHEAD;
#else
        constexpr int kActAhead = 8;
#endif
#if defined(ZENV_ACT_BATCH)
        constexpr int kActBatch = ZENV_ACT_BATCH;
#else
        constexpr int kActBatch = 8;
#endif
        const uint32_t touch_lds = (uint32_t)(size_t)(__attribute__((address_space(3))) void *)act_touch;
        auto touch_row = [&](int row) {
            const float *src = reinterpret_cast<const float *>(io.actions + (size_t)row * (size_t)N + min(env, N - 1));
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(src), "s"(touch_lds) : "m0");
        };
        if (EXT && kActAhead > 0) {
#pragma unroll 1
            for (int r = 1; r < kActAhead && r < n_steps; ++r) touch_row(r);
        }
        for (int t = 0; t < n_steps; ++t) {
            if (t == (n_steps >> 1)) ZSTAMP(8);
            if (EXT && kActAhead > 0 && t % kActBatch == 0) {
#pragma unroll 1
                for (int r = t + kActAhead; r < t + kActAhead + kActBatch && r < n_steps; ++r) touch_row(r);
            }
            lds_ctr_wait(ctr + 0, t + 1);                 // published(t)
            if (t == (n_steps >> 1)) ZSTAMP(9);
            const int b = t & 1;
#if !defined(ZENV_EXP) || !(ZENV_EXP & 8)   // diagnostic: bit 3 drops the per-env results of a step
            if (kStreamStores) {
                // the env wave's results of step t: obs as two 1 KiB bursts, then reward / done / goal_met / visit count
                // (and, EXT, the chunk's time-major records)
                const float4 o0 = obs_pub[b * 2 * kWave + lane], o1 = obs_pub[b * 2 * kWave + kWave + lane];
                const float4 m = misc_pub[b * kWave + lane];
                float4 *od = reinterpret_cast<float4 *>(p.obs + (size_t)env0 * 8);
                if (lane < 2 * n_blk) od[lane] = o0;
                if (kWave + lane < 2 * n_blk) od[kWave + lane] = o1;
                if (lane < n_blk) {
                    const int fl = __float_as_int(m.y);
                    p.reward[env] = m.x;
                    p.done_out[env] = (uint8_t)(fl & 1);
                    p.goal_met[env] = (uint8_t)((fl >> 1) & 1);
                    p.visit_count[env] = __float_as_int(m.z);
#if !defined(ZENV_EXP) || !(ZENV_EXP & 16)  // diagnostic: bit 4 drops the time-major reward / done records
                    if (EXT) {
                        __builtin_nontemporal_store(m.x, io.reward + (size_t)t * (size_t)N + env);
                        __builtin_nontemporal_store((uint8_t)(fl & 1), io.done + (size_t)t * (size_t)N + env);
                    }
#endif
                }
            }
#endif
#if !defined(ZENV_EXP) || !(ZENV_EXP & 1)   // diagnostic builds only: ZENV_EXP bit 0 drops the flush
            const uint8_t *cdb = reinterpret_cast<const uint8_t *>(cdw + b * kWave * (ZB / 4));
            if (__builtin_amdgcn_readfirstlane(*(lds_vint *)(ctr + 2 + b)))   // some env of the tile is frozen
                flush_static<TASK, ZT, true>(p, sent, dynw + b * kWave, cdb, stage, tile_dst, n_rows, lane, rt);
            else
                flush_static<TASK, ZT, false>(p, sent, dynw + b * kWave, cdb, stage, tile_dst, n_rows, lane, rt);
#endif
            lds_ctr_set(ctr + 1, t + 1);                  // flushed(t)
            if (t == 0) ZSTAMP(6);
            if (t == n_steps - 1) ZSTAMP(7);
            if (t == (n_steps >> 1) - 1) ZSTAMP(11);
            if (t == (n_steps >> 1)) ZSTAMP(10);
        }
        if (EXT) asm volatile("s_waitcnt vmcnt(0)");   // no prefetch may land in LDS after the workgroup has left
        return;
    }

    // ======================================================================= env wave
    ZSTAMP(12);
    if (!kStreamFirst) __builtin_amdgcn_s_setprio(3);
    // Everything the steady state of the loop does not touch -- the bank, the schedule, the state arrays, the episode
    // counters -- is read through the device copy of the parameter block, at the point of use inside the rare
    // branches (episode end, reset, rim test) and in the epilogue.  As by-value kernel arguments those ~25 pointers
    // are loaded at kernel entry and live across the whole loop: beyond the 102 SGPRs, i.e. v_readlane / v_writelane
    // spill traffic on the env wave's critical path in every step.  The copy is read through the CONSTANT address
    // space: uniform loads from it are scalar loads (s_load, lgkmcnt) -- as plain global loads they would be vector
    // loads whose s_waitcnt vmcnt(0) also drains the wave's outstanding observation stores.
    typedef const __attribute__((address_space(4))) DevParams ConstDevParams;
    const ConstDevParams &pc = *reinterpret_cast<ConstDevParams *>(reinterpret_cast<uintptr_t>(p.self));
    EnvRegs e;
    bool frozen = false;
    double ep_ret = 0.0;
    double hs = 0.0, hc = 1.0;     // sin / cos of half the hinge angle, carried from obs to the next step's physics
    int epi_idx = 0, nslot = 0, vcount = 0;
    float2 act = make_float2(0.f, 0.f);
    float4 zp[ZH];
    int auxr[ZT];
    e.steps = 0; e.vis = 0u; e.colpack = 0ull; e.goal_dist = 0;
    e.q0 = e.q1 = e.q2 = e.v0 = e.v1 = e.v2 = 0.0;
    e.x0 = e.y0 = e.bq0 = e.bq3 = 0.0;
#pragma unroll
    for (int h = 0; h < ZH; ++h) zp[h] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int z = 0; z < ZT; ++z) auxr[z] = 0;
    if (valid) {
        const double2 qa = p.qa[env], qb = p.qb[env], qc = p.qc[env];
        const double2 fa = p.fa[env], fb = p.fb[env];
        e.q0 = qa.x; e.q1 = qa.y; e.q2 = qb.x;
        e.v0 = qb.y; e.v1 = qc.x; e.v2 = qc.y;
        e.x0 = fa.x; e.y0 = fa.y; e.bq0 = fb.x; e.bq3 = fb.y;
        act = EXT ? io.actions[env] : reinterpret_cast<const float2 *>(p.actions)[env];
        det_sincos_inl(0.5 * e.q2, hs, hc);
        frozen = p.sched[env].done_state != 0;
        e.steps = p.hota[env].steps;
        if (kColour) {
            e.colpack = p.hotc[env].colpack;
            e.goal_dist = p.hota[env].vis;
        } else {
            e.vis = p.hota[env].vis;
        }
        ep_ret = p.hota[env].ep_return;
        vcount = p.visit_count[env];
        epi_idx = p.sched[env].episode_idx;
        nslot = p.sched[env].next_slot;
#pragma unroll
        for (int h = 0; h < ZH; ++h) zp[h] = p.zpf[(size_t)h * N + env];
#pragma unroll
        for (int z = 0; z < ZT; ++z) {
            const size_t zi = (size_t)z * N + env;
            if (TASK == ZENV_TASK_TIMED_TSP) auxr[z] = p.tmax[zi];
        }
        if (kColour) {
#pragma unroll
            for (int w = 0; w < (ZT + 7) / 8; ++w) {
                const uint64_t cw = *cd_word(p, w, env);
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (8 * w + i < ZT) auxr[8 * w + i] = (int)((cw >> (8 * i)) & 0xFFull);
            }
        }
    }
#pragma unroll
    for (int z = 0; z < ZT; ++z) {
        const float4 pr = zp[z >> 1];
        sent[lane * Z + z] = make_static<TASK>((z & 1) ? pr.z : pr.x, (z & 1) ? pr.w : pr.y, auxr[z]);
    }
    if (lane < 4) ctr[lane] = 0;
    __syncthreads();   // the only barrier of the launch

    auto write_back = [&](const EnvRegs &er, double ret) {
        store_dyn(pc, env, er);
        store_frame(pc, env, er);
        store_counters(pc, env, TASK, er);
        pc.hota[env].ep_return = ret;
        if (kColour) {
#pragma unroll
            for (int w = 0; w < (ZT + 7) / 8; ++w) {
                uint64_t cw = 0ull;
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (8 * w + i < ZT) cw |= (uint64_t)((uint32_t)auxr[8 * w + i] & 0xFFu) << (8 * i);
                *cd_word(pc, w, env) = cw;
            }
        }
    };
    // A step is written so that its common case has no divergent branch: every lane of the wave runs the zone pass,
    // the reward logic, the physics and the policy on its registers -- lanes without a live env (frozen, or beyond N in
    // the last tile) on harmless zero state -- and only the global stores and the rare events (rim test, episode end,
    // reset, frozen env) sit behind exec masks.
    const bool fs10 = p.frameskip == ZENV_SUBSTEP_UNROLL;
    // (EXT: the chunk's rows are addressed as wave-uniform row pointer + env -- scalar registers and the global
    // instructions' SADDR form; per-lane 64-bit pointers walking down the buffers cost the 25-zone kernels, which sit at
    // the 256-register limit, eight more registers and with them scratch spills inside the step loop)
    const int envl = min(env, N - 1);
    ZSTAMP(13);
    for (int t = 0; t < n_steps; ++t) {
        StepPolicy polt = pol;
        polt.step_index = pol.step_index + (uint32_t)t;
        const bool live = valid && !frozen;
        const bool ar_t = auto_reset == 1 || (auto_reset == 2 && t == n_steps - 1);
        // EXT: a_{t+1} is requested now and taken at the END of the step -- the env wave's ONLY vector-memory instruction of
        // a step (everything it produces goes to the stream wave through LDS), so the wait is for this load alone, and it
        // has the whole step to arrive.  The row is in L2 by then: the stream wave pulled it in, eight rows at a time (see
        // there -- fetched from memory one row per step beside the saturated row stream, this load cost 0.6-1.9 us per step).
        float2 act_next = act;
#if !defined(ZENV_EXP) || !(ZENV_EXP & 32)      // diagnostic: bit 5 drops the action fetch (a_0 is held for the whole launch)
        if (EXT && t + 1 < n_steps) {
#if defined(ZENV_EXP) && (ZENV_EXP & 64)      // diagnostic: bit 6 fetches row 0 every step (always cache-resident)
            act_next = io.actions[envl];
#else
            act_next = (io.actions + (size_t)(t + 1) * (size_t)N)[envl];     // in L2 since the stream wave's touch
#endif
            asm volatile("" ::: "memory");      // issued HERE: the scheduler may not sink the load towards its use
        }
#endif
        const int k = e.steps + 1;
        double rx, ry;
        world_pos(e, rx, ry);          // set_mocaps() sees the PRE-physics pose
        // ---- zone pass (see k_step_lane), entirely on registers
        const float rxf = (float)rx, ryf = (float)ry;
        // d2f >= +0, so its bit pattern orders like its value: "d2f < d2_lo" is the borrow of an
        // integer subtraction, shifted into the mask with one v_alignbit (zones visited from the
        // top so that zone z ends up in bit z).
        uint32_t in_mask = 0u, out_mask = 0u, elig_mask = 0u, expired = 0u;
        const uint32_t lo_bits = __float_as_uint(p.d2_lo), hi_bits = __float_as_uint(p.d2_hi);
#pragma unroll
        for (int z = ZT - 1; z >= 0; --z) {
            const float4 pr = zp[z >> 1];
            const float x3 = (z & 1) ? pr.z : pr.x, y3 = (z & 1) ? pr.w : pr.y;
            const float dxf = __builtin_fmaf(3.f, x3, -rxf), dyf = __builtin_fmaf(3.f, y3, -ryf);
            const uint32_t d2b = __float_as_uint(__builtin_fmaf(dxf, dxf, dyf * dyf));
            in_mask = __builtin_amdgcn_alignbit(in_mask, d2b - lo_bits, 31);       // d2f <  d2_lo
            out_mask = __builtin_amdgcn_alignbit(out_mask, hi_bits - d2b, 31);     // d2f >  d2_hi
            if (kColour) {
                const int cd = max(auxr[z] - 1, 0);         // colour_match_env.py:98-100
                auxr[z] = cd;
                elig_mask = __builtin_amdgcn_alignbit(elig_mask, (uint32_t)(cd - 1), 31);          // cd == 0
            } else if (TASK == ZENV_TASK_TIMED_TSP) {
                expired = __builtin_amdgcn_alignbit(expired, (uint32_t)(auxr[z] - (k + 1)), 31);   // tmax - k <= 0, TTSP_env.py:67
            }
        }
        // the rim: exact float64 test on the float64 zone centres (rare, divergent; live envs only -- the state of
        // this kernel's envs is finite, so the pose is never a NaN)
        uint32_t amb_mask = live ? ~(in_mask | out_mask) & full : 0u;
        while (amb_mask) {
            const int z = __ffs((int)amb_mask) - 1;
            amb_mask &= ~(1u << z);
            const double2 zz = pc.zxy[(size_t)z * N + env];
            const double dx = zz.x - rx, dy = zz.y - ry;
            if (dx * dx + dy * dy <= pc.hit_d2) in_mask |= 1u << z;
        }
        if (!kColour) elig_mask = ~e.vis;
        const uint32_t hits = in_mask & elig_mask & full;
        const int first = hits ? __ffs((int)hits) - 1 : -1;                // lowest index wins, one per step
        if (first >= 0) {
            if (kColour) {
                int col = (int)((e.colpack >> (2 * first)) & 3ull);
                col = (col == 2) ? 0 : col + 1;                  // Blue->Green->Red->Blue
                e.colpack = (e.colpack & ~(3ull << (2 * first))) | ((uint64_t)col << (2 * first));
#pragma unroll
                for (int z = 0; z < ZT; ++z)
                    if (z == first) auxr[z] = p.max_cd;
            } else {
                e.vis |= 1u << first;
            }
        }
        const bool timed_out = TASK == ZENV_TASK_TIMED_TSP && (expired & ~e.vis & full) != 0u;
        // ---- reward / goal / termination (Engine.step order).  The scripted on-device policies are finite by
        // construction; a caller's action (EXT) may hold a NaN: Engine.step's MujocoException branch, as in k_step_lane
        // -- reward_exception, done, no reward() / goal test, the joint state mj_resetData leaves (below).
        const bool exc = EXT && !(act.x == act.x && act.y == act.y);
        double r = 0.0;
        bool goal = false;
        if (kColour) {
            if (first >= 0 && !exc) {
                const int nd = hamming_to_goal(e.colpack, Z);
                r = (double)(e.goal_dist - nd);
                e.goal_dist = nd;
            }
            goal = e.goal_dist == 0;
        } else {
            r = first >= 0 ? 1.0 : 0.0;
            goal = (e.vis & full) == full;
        }
        if (exc) {
            r = p.reward_exc;
            goal = false;
        }
        if (goal) r += (double)(p.num_steps - e.steps) * p.tsr;   // pre-increment steps
        e.steps = k;
        const bool done = goal || exc || k >= p.num_steps || (TASK == ZENV_TASK_TIMED_TSP && timed_out);
        ep_ret = ep_ret + r;
        const float rew_out = live ? (float)r : 0.f;                   // a frozen env reports reward 0, done 1
        const uint8_t done_out = (!live || done) ? 1 : 0, goal_out = (live && goal) ? 1 : 0;
        // (a frozen env -- finished earlier under step_no_reset -- comes back at the first auto-reset step: the worker's
        // `if done: obs = env.reset()` after WaitWrapper's no-op, penv.py:8-11, wrappers.py:34-45)
        const bool need_reset = valid && ar_t && (frozen || done);
        // (an env auto-reset in this step reports the finished episode's count; a frozen one keeps its last)
        vcount = live ? (kColour ? e.goal_dist : (int)__popc(e.vis)) : vcount;
        if (live && done) {
            pc.last_return[env] = ep_ret;
            pc.last_len[env] = k;
            pc.episodes[env] += 1;
            pc.exception[env] = exc ? 1 : 0;
        }
        uint64_t dword = !live ? kDynZero
                         : kColour ? e.colpack
                                   : ((uint64_t)e.vis | (TASK == ZENV_TASK_TIMED_TSP ? (uint64_t)(uint32_t)k << 32 : 0ull));
        if (t == (n_steps >> 1)) ZSTAMP(1);

        // ---- Engine.step: ctrl = clip(action, ctrlrange); frameskip x mj_step; obs; next action.  (An env about to be
        // reset runs it too: its result is replaced below.)
        float o[8];
        {
            const double c0 = clamp_sym((double)act.x, 1.0), c1 = clamp_sym((double)act.y, 1.0);
#if !defined(ZENV_EXP) || !(ZENV_EXP & 2)   // diagnostic: bit 1 drops the physics
            physics_step(p, e, c0, c1, hs, hc, fs10);
#endif
            if (EXT) {
                if (exc) {                    // what mj_resetData leaves: qpos = qpos0, qvel = 0
                    e.q0 = e.q1 = e.q2 = 0.0;
                    e.v0 = e.v1 = e.v2 = 0.0;
                }
            }
            if (t == (n_steps >> 1)) ZSTAMP(2);
            emit_obs8(p, e, o, hs, hc);
#if !defined(ZENV_EXP) || !(ZENV_EXP & 4)   // diagnostic: bit 2 drops the action source
            if (!EXT && pol.policy == ZENV_POLICY_UNIFORM)
                act = uniform_action(polt.env_index0 + (uint64_t)env, polt.step_index, polt.seed);
            else if (!EXT && pol.policy == ZENV_POLICY_GREEDY)
                act = greedy_action_regs<TASK, ZT>(zp, auxr, e.vis, e.colpack, o[1], o[2], o[3], o[4]);
#endif
        }
        if (!live) {
            // finished and not auto-reset: masked no-op (WaitWrapper, wrappers.py:34-45) -- zero obs, zero action
            // (what either policy kernel makes of a zero obs)
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = 0.f;
            if (pol.policy >= 0) act = make_float2(0.f, 0.f);
        }
        if (live && done && !ar_t) {
            // finished, no auto-reset: frozen from the next step on.  Its registers go back to the state arrays NOW --
            // from here on the lane only idles through the steps on them (nothing of a frozen lane is written back).
            pc.sched[env].done_state = 1;
            write_back(e, ep_ret);
            frozen = true;
        }

        // ---- auto-reset (penv.py:8-11), wave-cooperative: lane z <-> zone z of the finished env
        unsigned long long pending = __ballot(need_reset);
        if (pending) {
            lds_ctr_wait(ctr + 1, t);     // flush(t-1) has read the static entries about to change
            int my_slot = 0;
            if (need_reset) {
                my_slot = next_bank_slot(pc, env, epi_idx, nslot);
                epi_idx += 1;
            }
            while (pending) {
                const int j = __ffsll((long long)pending) - 1;   // wave-uniform
                pending &= pending - 1;
                const int slot = __shfl(my_slot, j);
                const int env_j = env0 + j;
                const double *br = pc.bank_robot + 4 * (size_t)slot;
                const double b0 = br[0], b1 = br[1], b2 = br[2], b3 = br[3];
                const uint32_t vis0 = pc.vis0;
                int code = kColour ? 0 : (int)((vis0 >> (lane & 31)) & 1u), aux = 0;   // pre-visited zones
                float4 en = make_float4(0.f, 0.f, 0.f, 0.f);
                if (lane < Z) {
                    const size_t bi = (size_t)slot * Z + lane;
                    const double2 zz = reinterpret_cast<const double2 *>(pc.bank_zone)[bi];
                    const size_t zi = (size_t)lane * N + env_j;
                    pc.zxy[zi] = zz;
                    if (TASK == ZENV_TASK_TIMED_TSP) {
                        aux = pc.bank_aux[bi];
                        pc.tmax[zi] = aux;
                    } else if (kColour) {
                        code = pc.bank_aux[bi];
                    }
                    en = make_entry<TASK>(p, zz.x, zz.y, code, aux, 0);
                    sent[j * Z + lane] = make_static<TASK>(en.x, en.y, aux);
                }
                {
                    const float nx = __shfl_down(en.x, 1), ny = __shfl_down(en.y, 1);
                    if (lane < Z && !(lane & 1))
                        pc.zpf[(size_t)(lane >> 1) * N + env_j] =
                            make_float4(en.x, en.y, lane + 1 < Z ? nx : 0.f, lane + 1 < Z ? ny : 0.f);
                }
                uint64_t colpack = 0ull;
                if (kColour) {
                    const unsigned long long m0 = __ballot(lane < Z && (code & 1));
                    const unsigned long long m1 = __ballot(lane < Z && (code & 2));
                    colpack = spread_even_bits((uint32_t)m0) | (spread_even_bits((uint32_t)m1) << 1);
                }
                EnvRegs fresh;
                fresh.x0 = b0; fresh.y0 = b1; fresh.bq0 = b2; fresh.bq3 = b3;
                fresh.q0 = fresh.q1 = fresh.q2 = 0.0;
                fresh.v0 = fresh.v1 = fresh.v2 = 0.0;
                fresh.vis = kColour ? 0u : vis0;
                fresh.colpack = colpack;
                fresh.goal_dist = kColour ? hamming_to_goal(colpack, Z) : 0;
                fresh.steps = 0;
                float of[8];                    // first obs + first greedy action of the episode: derived bank rows
                const float4 f2 = load_bank_first(pc, slot, of);
                float2 next_act = make_float2(f2.x, f2.y);
                if (!EXT && pol.policy == ZENV_POLICY_UNIFORM)
                    next_act = uniform_action(polt.env_index0 + (uint64_t)env_j, polt.step_index, polt.seed);
                wave_lds_fence();   // lane j reads back what its neighbours wrote
#pragma unroll
                for (int h = 0; h < ZH; ++h) {
                    const SE a = sent[j * Z + 2 * h];
                    const SE b = sent[j * Z + ((2 * h + 1 < Z) ? 2 * h + 1 : 2 * h)];
                    if (lane == j) zp[h] = make_float4(a.x, a.y, (2 * h + 1 < Z) ? b.x : 0.f, (2 * h + 1 < Z) ? b.y : 0.f);
                }
                if constexpr (TASK == ZENV_TASK_TIMED_TSP) {
#pragma unroll
                    for (int z = 0; z < ZT; ++z) {
                        const int v = __float_as_int(sent[j * Z + z].z);
                        if (lane == j) auxr[z] = v;
                    }
                }
                if (lane == j) {
                    if (kColour) {
#pragma unroll
                        for (int z = 0; z < ZT; ++z) auxr[z] = 0;
                    }
                    e = fresh;
                    ep_ret = 0.0;
                    hs = 0.0;
                    hc = 1.0;
                    if (!EXT) act = next_act;
                    dword = kColour ? colpack : (uint64_t)vis0;   // only the pre-visited zones, step count 0
                    pc.seed[env] = pc.bank_seed[slot];
#pragma unroll
                    for (int i = 0; i < 8; ++i) o[i] = of[i];      // the new episode's first obs is what this step returns
                    if (frozen) {
                        pc.sched[env].done_state = 0;
                        frozen = false;
                    }
                }
            }
        }

        // ---- publish the step word of this step (its slot was last read by flush(t-2))
        if (t == (n_steps >> 1)) ZSTAMP(3);
        if (t >= 2) lds_ctr_wait(ctr + 1, t - 1);
        if (t == (n_steps >> 1)) ZSTAMP(4);
        {
            const int b = t & 1;
            dynw[b * kWave + lane] = dword;
            if (kStreamStores) {
                obs_pub[b * 2 * kWave + 2 * lane] = make_float4(o[0], o[1], o[2], o[3]);
                obs_pub[b * 2 * kWave + 2 * lane + 1] = make_float4(o[4], o[5], o[6], o[7]);
                misc_pub[b * kWave + lane] = make_float4(rew_out, __int_as_float((int)done_out | ((int)goal_out << 1)),
                                                         __int_as_float(vcount), 0.f);
            }
            if (kColour) {
#pragma unroll
                for (int w = 0; w < ZB / 4; ++w) {
                    uint32_t pk = 0u;
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (4 * w + i < ZT) pk |= ((uint32_t)auxr[4 * w + i] & 0xFFu) << (8 * i);
                    cdw[(b * kWave + lane) * (ZB / 4) + w] = pk;
                }
            }
        }
        {
            const int any_zero = __ballot((dword & kDynZero) != 0ull && valid) != 0ull;
            if (lane == 0) *(lds_vint *)(ctr + 2 + (t & 1)) = any_zero;
        }
        lds_ctr_set(ctr + 0, t + 1);                      // published(t)
#if !defined(ZENV_EXP) || !(ZENV_EXP & 8)
        if (!kStreamStores && valid) {
            store_obs8(p, env, o);
            p.visit_count[env] = vcount;
            p.reward[env] = rew_out;
            p.done_out[env] = done_out;
            p.goal_met[env] = goal_out;
        }
#endif
        if (EXT) {
            asm volatile("" : "+v"(act_next.x), "+v"(act_next.y)::"memory");
            act = act_next;
        }
        if (t == (n_steps >> 1)) ZSTAMP(5);
    }

    // ---- the registers go back to the state arrays (a frozen env's did when it froze, or never left them)
    ZSTAMP(14);
    if (valid) {
        if (!frozen) write_back(e, ep_ret);
        if (!EXT && pol.policy >= 0) reinterpret_cast<float2 *>(pol.out)[env] = act;
    }
    ZSTAMP(15);
}

// =========================================================================== K6: goal shaping
// TSPNextCityEnv / TimedTSPNextCityEnv / ColourMatchNextCityEnv (main/envs/zone_envs/TSP_next_city_env.py:41-109,
// zone-goals/envs/TTSP_next_city_env.py:40-51, zone-goals/envs/colour_match_next_city_env.py): a goal zone per env, a dense reward towards it and the
// request for the next goal.  Runs right after the step kernel, which leaves the zone visited in this
// step and -- for an env that was auto-reset -- the position where the episode ended.
__global__ __launch_bounds__(256) void k_goal_set(DevParams p, const int32_t *__restrict__ new_goal, int32_t *bad)
{
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= p.N) return;
    const int g = new_goal[env];
    if (g < 0) return;                                         // leave this env's goal alone
    // set_goal asserts the zone is unvisited (:86); ColourMatchNextCityEnv only checks the range
    // (zone-goals/envs/colour_match_next_city_env.py set_goal)
    if (g >= p.Z || (p.task != ZENV_TASK_COLOUR_MATCH && ((p.hota[env].vis >> g) & 1u))) {
        atomicAdd(bad, 1);
        return;
    }
    EnvRegs e;
    const double2 qa = p.qa[env], fa = p.fa[env], fb = p.fb[env];
    e.q0 = qa.x; e.q1 = qa.y; e.x0 = fa.x; e.y0 = fa.y; e.bq0 = fb.x; e.bq3 = fb.y;
    double rx, ry;
    world_pos(e, rx, ry);
    const double2 zz = p.zxy[(size_t)g * p.N + env];
    const double dx = zz.x - rx, dy = zz.y - ry;
    p.goal[env] = g;
    p.goal_xy[env] = zz;
    p.goal_last[env] = sqrt(dx * dx + dy * dy);                // dist_to_goal, :41-45
    p.need_goal[env] = 0;
}

// reset(): the env has no goal yet (goal_zone = None), every zone is available
__global__ __launch_bounds__(256) void k_goal_clear(DevParams p, const uint8_t *__restrict__ mask)
{
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= p.N || (mask && !mask[env])) return;
    p.goal[env] = -1;
    p.need_goal[env] = 1;
    p.shaped[env] = 0.0;
    p.visit_zone[env] = -1;
    const uint32_t full = (p.Z >= 32) ? 0xFFFFFFFFu : ((1u << p.Z) - 1u);
    p.available[env] = p.task == ZENV_TASK_COLOUR_MATCH ? full : (~p.vis0 & full);
}

__global__ __launch_bounds__(256) void k_goal_step(DevParams p)
{
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= p.N) return;
    const uint32_t full = (p.Z >= 32) ? 0xFFFFFFFFu : ((1u << p.Z) - 1u);
    const int g = p.goal[env];
    const bool done = p.done_out[env] != 0;
    double sh = 0.0;
    uint8_t need = g < 0;
    if (g >= 0) {
        const bool reached = p.visit_zone[env] == g;           // new_city_reached and zones[goal] == visited
        if (!reached) {
            double rx, ry;
            if (done && !p.sched[env].done_state) {                  // auto-reset happened: the terminal position
                const double2 t = p.term_xy[env];
                rx = t.x; ry = t.y;
            } else {
                EnvRegs e;
                const double2 qa = p.qa[env], fa = p.fa[env], fb = p.fb[env];
                e.q0 = qa.x; e.q1 = qa.y; e.x0 = fa.x; e.y0 = fa.y; e.bq0 = fb.x; e.bq3 = fb.y;
                world_pos(e, rx, ry);
            }
            // the goal zone's centre is kept beside the goal: after an auto-reset the zone arrays already
            // hold the next map
            const double2 zz = p.goal_xy[env];
            const double dx = zz.x - rx, dy = zz.y - ry;
            const double d = sqrt(dx * dx + dy * dy);
            sh = p.goal_last[env] - d;                         // :63-66
            p.goal_last[env] = d;
            // ColourMatchNextCityEnv.step: cycling a zone other than the goal costs 1
            if (p.task == ZENV_TASK_COLOUR_MATCH && p.visit_zone[env] >= 0) sh -= 1.0;
        }
        if (reached || done) {                                 // :69-72, TTSP_next_city_env.py:46-49
            need = 1;
            p.goal[env] = -1;
        }
    }
    p.shaped[env] = sh;
    p.need_goal[env] = need;
    // get_available_goals, :92-100; every zone for ColourMatch
    p.available[env] = p.task == ZENV_TASK_COLOUR_MATCH ? full : (~p.hota[env].vis & full);
}

// ColourMatchSolverEnv.solver_get_next_goal (zone-goals/envs/colour_match_solver_env.py:57-97): among the zones whose
// colour is NOT the target colour of some cheapest recolouring plan, the one nearest to the robot (ties: lowest index --
// the reference sorts (distance, index) tuples).
__global__ __launch_bounds__(256) void k_solver_goal(DevParams p, int32_t *__restrict__ out)
{
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= p.N) return;
    const uint64_t cp = p.hotc[env].colpack;
    int n[3] = { 0, 0, 0 };                                    // Blue, Green, Red
    for (int z = 0; z < p.Z; ++z) n[(int)((cp >> (2 * z)) & 3ull) % 3] += 1;
    const int to[3] = { n[1] * 2 + n[2], n[2] * 2 + n[0], n[0] * 2 + n[1] };   // dist_to_blue / _green / _red
    const int mn = min(to[0], min(to[1], to[2]));
    EnvRegs e;
    const double2 qa = p.qa[env], fa = p.fa[env], fb = p.fb[env];
    e.q0 = qa.x; e.q1 = qa.y; e.x0 = fa.x; e.y0 = fa.y; e.bq0 = fb.x; e.bq3 = fb.y;
    double rx, ry;
    world_pos(e, rx, ry);
    int best = -1;
    double bd = 0.0;
    for (int z = 0; z < p.Z; ++z) {
        const int c = (int)((cp >> (2 * z)) & 3ull) % 3;
        if (!(to[(c + 1) % 3] == mn || to[(c + 2) % 3] == mn)) continue;   // a plan that recolours this zone is cheapest
        const double2 zz = p.zxy[(size_t)z * p.N + env];
        const double dx = zz.x - rx, dy = zz.y - ry;
        const double d = sqrt(dx * dx + dy * dy);
        if (best < 0 || d < bd) { best = z; bd = d; }
    }
    out[env] = best;
}

// =========================================================================== K7: solver-ordered TSP
// TSPOrderEnv (main/envs/TSP_order_env.py:13-113): the zones carry a visiting order (a route from a TSP solver,
// here the bank's aux column: rank of every zone), the observation gets the feature 0.5^(position in the remaining
// route) per zone (:37-47) and info['shaped_reward'] is the progress towards the first zone of the remaining route
// (:52-75).  Like K6 it runs after the step kernel, on the zone visited in the step and the terminal position.
__device__ __forceinline__ int current_bank_slot(const DevParams &p, int env)
{
    if (p.sched_mode != SCHED_FIXED_SEEDS) return seq_slot(p, p.slot_first[env], p.sched[env].episode_idx - 1);
    return (int)(p.seed[env] - p.seed_min);
}
__device__ __forceinline__ void env_world_pos(const DevParams &p, int env, double &rx, double &ry)
{
    EnvRegs e;
    const double2 qa = p.qa[env], fa = p.fa[env], fb = p.fb[env];
    e.q0 = qa.x; e.q1 = qa.y; e.x0 = fa.x; e.y0 = fa.y; e.bq0 = fb.x; e.bq3 = fb.y;
    world_pos(e, rx, ry);
}
// route[0] = the unvisited zone at position 0; dist_to_goal (:52-57), 0 for an empty route
__device__ __forceinline__ double order_goal_dist(const DevParams &p, int env, const int8_t *pos, double rx, double ry,
                                                  bool from_goal_xy)
{
    int g = -1;
    for (int z = 0; z < p.Z; ++z)
        if (pos[z] == 0) g = z;
    p.goal[env] = g;
    if (g < 0) return 0.0;
    const double2 zz = from_goal_xy ? p.goal_xy[env] : p.zxy[(size_t)g * p.N + env];
    if (!from_goal_xy) p.goal_xy[env] = zz;
    const double dx = zz.x - rx, dy = zz.y - ry;
    return sqrt(dx * dx + dy * dy);
}
// What obs_zones() makes of the route as it stands (:41-45): np.power(0.5, route.index(i)), 0 for a zone not in it
__device__ __forceinline__ void order_emit_feature(const DevParams &p, int env)
{
    const int8_t *pos = p.order_pos + (size_t)env * p.Z;
    float *val = p.order_val + (size_t)env * p.Z;
    for (int z = 0; z < p.Z; ++z) val[z] = pos[z] >= 0 ? __builtin_ldexpf(1.0f, -pos[z]) : 0.f;
}
// reset() (:108-113), in the reference's statement order: `init_obs = super().reset()` has ALREADY built the first
// observation when `generate_route()` runs -- from self.route as the previous episode left it (empty after a finished
// episode, the unvisited rest after a time-limit end; self.route = [] before the first reset, :27).  So: feature of
// the old route first, then the new route and last_dist_to_goal = dist_to_goal().  p.order_fresh (the build's opt-out,
// zenv_order_configure): the first observation shows the new route instead.
__device__ __forceinline__ void order_reset_env(const DevParams &p, int env)
{
    int8_t *pos = p.order_pos + (size_t)env * p.Z;
    if (!p.order_fresh) order_emit_feature(p, env);
    const int32_t *rank = p.bank_aux + (size_t)current_bank_slot(p, env) * p.Z;
    for (int z = 0; z < p.Z; ++z) pos[z] = (int8_t)rank[z];
    if (p.order_fresh) order_emit_feature(p, env);
    double rx, ry;
    env_world_pos(p, env, rx, ry);
    p.goal_last[env] = order_goal_dist(p, env, pos, rx, ry, false);
}

__global__ __launch_bounds__(256) void k_order_reset(DevParams p, const uint8_t *__restrict__ mask)
{
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= p.N || (mask && !mask[env])) return;
    order_reset_env(p, env);
    p.shaped[env] = 0.0;
    p.visit_zone[env] = -1;
}

__global__ __launch_bounds__(256) void k_order_step(DevParams p)
{
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= p.N) return;
    int8_t *pos = p.order_pos + (size_t)env * p.Z;
    const bool done = p.done_out[env] != 0;
    const bool was_reset = done && !p.sched[env].done_state;
    const int v = p.visit_zone[env];
    if (v >= 0) {
        // set_mocaps: self.route.remove(h_index) (:90)
        const int pv = pos[v];
        for (int z = 0; z < p.Z; ++z)
            if (pos[z] > pv) pos[z] -= 1;
        pos[v] = -1;
    }
    double rx, ry;
    if (was_reset) {
        const double2 t = p.term_xy[env];                      // where the finished episode ended
        rx = t.x; ry = t.y;
    } else {
        env_world_pos(p, env, rx, ry);
    }
    // shaped_reward() (:63-72): after a visit the reference re-bases on the NEW first zone and returns 0
    double sh = 0.0;
    if (v >= 0) {
        // the new route[0] belongs to the finished episode's map: its centre must come from zxy only when that
        // map is still loaded
        if (!was_reset) p.goal_last[env] = order_goal_dist(p, env, pos, rx, ry, false);
    } else {
        const double d = order_goal_dist(p, env, pos, rx, ry, true);
        sh = p.goal_last[env] - d;
        p.goal_last[env] = d;
    }
    p.shaped[env] = sh;
    // the worker's `if done: obs = env.reset()` (penv.py:8-11) goes through the same reset(): the returned observation
    // carries the feature of the route this step left behind
    if (was_reset) order_reset_env(p, env);
    else order_emit_feature(p, env);
}

// =========================================================================== K1w: wave-per-env step
// The layout north_star describes literally (cfg.kernel = ZENV_KERNEL_WAVE_PER_ENV): one wave64 per env,
// lane z owns zone z; the first-eligible-zone choice, the deadline check and the reset's colour packing are
// wave ballots; the env's scalar state (joint state, counters, the 8-float obs) is wave-uniform, so every
// lane carries the same physics and lane 0 stores it.  Same arithmetic, same results as K1 (the zone test
// is the exact float64 one in every lane).  It is here to be measured against K1: for the
// reference's Z <= 25 the 10 float64 substeps dominate and they run once per WAVE instead of once per
// LANE, so the chip does 64x the physics issue for the same batch.  Scripted policies run as the
// stand-alone K3 launch (zenv_rollout falls back to the unfused loop for this layout).
template <int TASK>
__global__ __launch_bounds__(4 * kWave) void k_step_wave(DevParams p, const float *__restrict__ actions,
                                                         int auto_reset)
{
    constexpr int F = TaskTraits<TASK>::F;
    const int lane = threadIdx.x & (kWave - 1);
    const int env = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    const int N = p.N, Z = p.Z;
    if (env >= N) return;
    const bool zl = lane < Z;
    const size_t zi = (size_t)lane * N + env;
    const GlobalRowSink sink{ p.zone_obs + (size_t)env * Z * F };

    const bool revive = p.sched[env].done_state != 0 && auto_reset;   // frozen env at an auto-reset step: see k_step_lane
    if (p.sched[env].done_state && !auto_reset) {
        // finished and not auto-reset: masked no-op (WaitWrapper, wrappers.py:34-45)
        if (zl) sink.put<TASK>(lane, make_float4(0.f, 0.f, -1.f, 0.f));
        if (lane == 0) {
            const float o[8] = { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f };
            store_obs8(p, env, o);
            if (p.visit_zone) p.visit_zone[env] = -1;
            p.reward[env] = 0.f;
            p.done_out[env] = 1;
            p.goal_met[env] = 0;
        }
        return;
    }

    EnvRegs e;
    {
        const double2 qa = p.qa[env], qb = p.qb[env], qc = p.qc[env], fa = p.fa[env], fb = p.fb[env];
        e.q0 = qa.x; e.q1 = qa.y; e.q2 = qb.x;
        e.v0 = qb.y; e.v1 = qc.x; e.v2 = qc.y;
        e.x0 = fa.x; e.y0 = fa.y; e.bq0 = fb.x; e.bq3 = fb.y;
    }
    e.steps = p.hota[env].steps;
    e.vis = 0u; e.colpack = 0ull; e.goal_dist = 0;
    if (TASK == ZENV_TASK_COLOUR_MATCH) {
        e.colpack = p.hotc[env].colpack;
        e.goal_dist = p.hota[env].vis;
    } else {
        e.vis = p.hota[env].vis;
    }
    double ep_ret = p.hota[env].ep_return;
    const float2 act = reinterpret_cast<const float2 *>(actions)[env];
    const int k = e.steps + 1;
    const uint32_t full = (Z >= 32) ? 0xFFFFFFFFu : ((1u << Z) - 1u);

    // ---- set_mocaps() of the first substep on the pre-physics pose: lane z tests zone z
    double rx, ry;
    world_pos(e, rx, ry);
    double2 zz = make_double2(0.0, 0.0);
    int aux = 0;
    if (zl) {
        zz = p.zxy[zi];
        if (TASK == ZENV_TASK_TIMED_TSP) aux = p.tmax[zi];
        if (TASK == ZENV_TASK_COLOUR_MATCH) aux = *cd_byte(p, lane, env);
    }
    const double dx = zz.x - rx, dy = zz.y - ry;
    const bool inside = zl && (dx * dx + dy * dy <= p.hit_d2);
    bool elig;
    if (TASK == ZENV_TASK_COLOUR_MATCH) {
        if (aux > 0) aux -= 1;                                   // colour_match_env.py:98-100
        elig = aux == 0;
    } else {
        elig = !((e.vis >> lane) & 1u);
    }
    const unsigned long long hits = __ballot(inside && elig);
    const int first = hits ? __ffsll((long long)hits) - 1 : -1;  // lowest index wins, one per step
    if (first >= 0) {
        if (TASK == ZENV_TASK_COLOUR_MATCH) {
            int col = (int)((e.colpack >> (2 * first)) & 3ull);
            col = (col == 2) ? 0 : col + 1;                      // Blue->Green->Red->Blue
            e.colpack = (e.colpack & ~(3ull << (2 * first))) | ((uint64_t)col << (2 * first));
            if (lane == first) aux = p.max_cd;
        } else {
            e.vis |= 1u << first;
        }
    }
    if (TASK == ZENV_TASK_COLOUR_MATCH && zl) *cd_byte(p, lane, env) = (uint8_t)aux;
    const bool visited = (e.vis >> lane) & 1u;
    bool timed_out = false;
    if (TASK == ZENV_TASK_TIMED_TSP) timed_out = __ballot(zl && !visited && (aux - k) <= 0) != 0ull;   // TTSP_env.py:67

    // ---- reward / goal / termination (Engine.step order); a NaN action takes the exception path (see k_step_lane)
    const bool exc = !(act.x == act.x && act.y == act.y);
    double r = 0.0;
    bool goal = false;
    bool done = false;
    if (exc) {
        r = p.reward_exc;
        done = true;
    } else if (TASK == ZENV_TASK_COLOUR_MATCH) {
        if (first >= 0) {
            const int nd = hamming_to_goal(e.colpack, Z);
            r = (double)(e.goal_dist - nd);
            e.goal_dist = nd;
        }
        goal = e.goal_dist == 0;
    } else {
        r = first >= 0 ? 1.0 : 0.0;
        goal = (e.vis & full) == full;
    }
    if (goal) {
        r += (double)(p.num_steps - e.steps) * p.tsr;
        done = true;
    }
    e.steps = k;
    if (k >= p.num_steps) done = true;
    if (TASK == ZENV_TASK_TIMED_TSP && !done && timed_out) done = true;
    ep_ret = ep_ret + r;

    // ---- Engine.step: ctrl = clip(action, ctrlrange); frameskip x mj_step (wave-uniform)
    const double c0 = det_clamp((double)act.x, -1.0, 1.0);
    const double c1 = det_clamp((double)act.y, -1.0, 1.0);
    {
        double hs, hc;
        det_sincos_inl(0.5 * e.q2, hs, hc);
        physics_step(p, e, c0, c1, hs, hc, p.frameskip == ZENV_SUBSTEP_UNROLL);
    }
    if (exc) {
        e.q0 = e.q1 = e.q2 = 0.0;
        e.v0 = e.v1 = e.v2 = 0.0;
    }

    if (lane == 0) {
        // (revive: the step above ran on the frozen state and counts for nothing -- reward 0, done, then the reset)
        if (!revive) p.visit_count[env] = (TASK == ZENV_TASK_COLOUR_MATCH) ? e.goal_dist : __popc(e.vis);
        if (p.visit_zone) p.visit_zone[env] = revive ? -1 : first;
        p.reward[env] = revive ? 0.f : (float)r;
        p.done_out[env] = (done || revive) ? 1 : 0;
        p.goal_met[env] = (goal && !revive) ? 1 : 0;
        if (done && !revive) {
            p.last_return[env] = ep_ret;
            p.last_len[env] = k;
            p.episodes[env] += 1;
            p.exception[env] = exc ? 1 : 0;
            if (!auto_reset) p.sched[env].done_state = 1;
        }
        if (revive) p.sched[env].done_state = 0;
    }

    if ((done || revive) && auto_reset) {
        // ---- auto-reset (penv.py:8-11): lane z fetches zone z of the next layout
        if (lane == 0 && p.term_xy) {
            double tx, ty;
            world_pos(e, tx, ty);
            p.term_xy[env] = make_double2(tx, ty);
        }
        int slot = 0;
        if (lane == 0) slot = next_bank_slot(p, env);
        slot = __builtin_amdgcn_readfirstlane(slot);
        const double *br = p.bank_robot + 4 * (size_t)slot;
        int code = (TASK == ZENV_TASK_COLOUR_MATCH) ? 0 : (int)((p.vis0 >> (lane & 31)) & 1u);
        float4 en = make_float4(0.f, 0.f, 0.f, 0.f);
        if (zl) {
            const size_t bi = (size_t)slot * Z + lane;
            zz = reinterpret_cast<const double2 *>(p.bank_zone)[bi];
            p.zxy[zi] = zz;
            int a = 0;
            if (TASK == ZENV_TASK_TIMED_TSP) {
                a = p.bank_aux[bi];
                p.tmax[zi] = a;
            } else if (TASK == ZENV_TASK_COLOUR_MATCH) {
                code = p.bank_aux[bi];
                *cd_byte(p, lane, env) = 0;
            }
            en = make_entry<TASK>(p, zz.x, zz.y, code, a, 0);
            sink.put<TASK>(lane, en);
        }
        const float nx = __shfl_down(en.x, 1), ny = __shfl_down(en.y, 1);
        if (zl && !(lane & 1))
            p.zpf[(size_t)(lane >> 1) * N + env] =
                make_float4(en.x, en.y, lane + 1 < Z ? nx : 0.f, lane + 1 < Z ? ny : 0.f);
        EnvRegs fresh;
        fresh.x0 = br[0]; fresh.y0 = br[1]; fresh.bq0 = br[2]; fresh.bq3 = br[3];
        fresh.q0 = fresh.q1 = fresh.q2 = 0.0;
        fresh.v0 = fresh.v1 = fresh.v2 = 0.0;
        fresh.vis = (TASK == ZENV_TASK_COLOUR_MATCH) ? 0u : p.vis0;
        fresh.colpack = 0ull;
        fresh.goal_dist = 0;
        if (TASK == ZENV_TASK_COLOUR_MATCH) {
            const unsigned long long m0 = __ballot(zl && (code & 1)), m1 = __ballot(zl && (code & 2));
            fresh.colpack = spread_even_bits((uint32_t)m0) | (spread_even_bits((uint32_t)m1) << 1);
            fresh.goal_dist = hamming_to_goal(fresh.colpack, Z);
        }
        fresh.steps = 0;
        e = fresh;
        ep_ret = 0.0;
        if (lane == 0) {
            p.seed[env] = p.bank_seed[slot];
            store_frame(p, env, e);
        }
    } else if (zl) {
        // this step's row of zone z
        int code = visited ? 1 : 0, a = aux;
        if (TASK == ZENV_TASK_COLOUR_MATCH) code = (int)((e.colpack >> (2 * lane)) & 3ull);
        sink.put<TASK>(lane, make_entry<TASK>(p, zz.x, zz.y, code, a, k));
    }

    if (lane == 0) {
        float o[8];
        emit_obs8(p, e, o);
        store_obs8(p, env, o);
        store_dyn(p, env, e);
        store_counters(p, env, TASK, e);
        p.hota[env].ep_return = ep_ret;
    }
}

// =========================================================================== K2: reset
template <int TASK>
__global__ __launch_bounds__(kWave) void k_reset_lane(DevParams p, const uint8_t *__restrict__ mask)
{
    constexpr int F = TaskTraits<TASK>::F;
    const int env = blockIdx.x * kWave + threadIdx.x;
    if (env >= p.N) return;
    if (mask && !mask[env]) return;
    EnvRegs e;
    const int slot = next_bank_slot(p, env);
    reset_env<TASK>(p, env, slot, e, GlobalRowSink{ p.zone_obs + (size_t)env * p.Z * F });
    store_dyn(p, env, e);
    store_frame(p, env, e);
    store_counters(p, env, TASK, e);
    p.sched[env].done_state = 0;
    p.hota[env].ep_return = 0.0;
    p.visit_count[env] = (TASK == ZENV_TASK_COLOUR_MATCH) ? e.goal_dist : __popc(e.vis);
    p.exception[env] = 0;
    float o[8];
    emit_obs8(p, e, o);
    store_obs8(p, env, o);
    p.reward[env] = 0.f;
    p.done_out[env] = 0;
    p.goal_met[env] = 0;
}

// =========================================================================== bank derive
// What an episode's first step needs that is a pure function of its bank row: the first observation (ZoneEnvBase.obs
// of the freshly reset env: qpos = qvel = 0, steps = 0) and the scripted greedy policy's first action.  One wave per
// bank slot, lane z <-> zone z; computed once when the bank is uploaded or a slot is refilled, with the same device
// functions the step kernels use, so that the auto-reset inside a step (penv.py:8-11) only copies.
template <int TASK>
__global__ __launch_bounds__(kWave) void k_bank_derive(DevParams p, const int32_t *__restrict__ slots)
{
    const int lane = threadIdx.x, Z = p.Z;
    const int slot = slots ? slots[blockIdx.x] : (int)blockIdx.x;
    const double *br = p.bank_robot + 4 * (size_t)slot;
    int code = (TASK == ZENV_TASK_COLOUR_MATCH) ? 0 : (int)((p.vis0 >> (lane & 31)) & 1u);
    float4 en = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lane < Z) {
        const size_t bi = (size_t)slot * Z + lane;
        const double2 zz = reinterpret_cast<const double2 *>(p.bank_zone)[bi];
        int aux = 0;
        if (TASK == ZENV_TASK_TIMED_TSP) aux = p.bank_aux[bi];
        if (TASK == ZENV_TASK_COLOUR_MATCH) code = p.bank_aux[bi];
        en = make_entry<TASK>(p, zz.x, zz.y, code, aux, 0);
    }
    EnvRegs fresh;
    fresh.x0 = br[0]; fresh.y0 = br[1]; fresh.bq0 = br[2]; fresh.bq3 = br[3];
    fresh.q0 = fresh.q1 = fresh.q2 = 0.0;
    fresh.v0 = fresh.v1 = fresh.v2 = 0.0;
    fresh.vis = 0u; fresh.colpack = 0ull; fresh.goal_dist = 0;
    fresh.steps = 0;
    float of[8];
    emit_obs8(p, fresh, of);
    const float2 a = greedy_action_coop<TASK>(lane, Z, en.x, en.y, code, en.w, of[1], of[2], of[3], of[4]);
    if (lane == 0) {
        float4 *bf = const_cast<float4 *>(p.bank_first) + 3 * (size_t)slot;
        bf[0] = make_float4(of[0], of[1], of[2], of[3]);
        bf[1] = make_float4(of[4], of[5], of[6], of[7]);
        bf[2] = make_float4(a.x, a.y, 0.f, 0.f);
    }
}

// =========================================================================== bank update
// zenv_bank_update: `count` freshly sampled layouts, packed one record per layout in a staging buffer
// ([robot 4 f64 | zone 2Z f64 | seed i64 | aux Z i32, padded to 8 B]), scattered into the bank slots they belong to.
__global__ __launch_bounds__(64) void k_bank_scatter(DevParams p, const int32_t *__restrict__ slots,
                                                     const unsigned char *__restrict__ staging, int rec_bytes)
{
    const int slot = slots[blockIdx.x], Z = p.Z, lane = threadIdx.x;
    const unsigned char *rec = staging + (size_t)blockIdx.x * rec_bytes;
    const double *rd = reinterpret_cast<const double *>(rec);
    double *bank_robot = const_cast<double *>(p.bank_robot), *bank_zone = const_cast<double *>(p.bank_zone);
    if (lane < 4) bank_robot[4 * (size_t)slot + lane] = rd[lane];
    for (int i = lane; i < 2 * Z; i += 64) bank_zone[2 * (size_t)slot * Z + i] = rd[4 + i];
    if (lane == 0) const_cast<int64_t *>(p.bank_seed)[slot] = reinterpret_cast<const int64_t *>(rd + 4 + 2 * Z)[0];
    const int32_t *ra = reinterpret_cast<const int32_t *>(rd + 4 + 2 * Z + 1);
    for (int i = lane; i < Z; i += 64) const_cast<int32_t *>(p.bank_aux)[(size_t)slot * Z + i] = ra[i];
}

// =========================================================================== schedule sync
// next_slot[env] = seq_slot(slot_first[env], episode_idx[env]): the invariant of slot_after(), re-established after the
// host changed the schedule or swapped the bank for one of another size (the modulo lives here, off every step path).
__global__ __launch_bounds__(256) void k_sched_sync(DevParams p, int restart)
{
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= p.N) return;
    if (restart) p.sched[env].episode_idx = 0;         // a new schedule starts at episode 0
    p.sched[env].next_slot = p.sched_mode == SCHED_FIXED_SEEDS ? 0 : seq_slot(p, p.slot_first[env], p.sched[env].episode_idx);
}

// ZENV_F_EP_RETURN / ZENV_F_EP_LEN live inside the HotA records: unpacked into plain arrays when a caller asks for them
__global__ __launch_bounds__(256) void k_unpack_hot(DevParams p, double *__restrict__ ep_return, int32_t *__restrict__ steps)
{
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= p.N) return;
    const HotA a = p.hota[env];
    ep_return[env] = a.ep_return;
    steps[env] = a.steps;
}

// =========================================================================== gather prep
// The send buffer of the job's one collective (zenv_allgather): field [N] -> 4-byte elements, float64 narrowed to
// float32 ("ncclAllGather of float ep_return[N/G]").
__global__ __launch_bounds__(256) void k_gather_prep(const void *__restrict__ src, int elem_bytes, uint32_t *__restrict__ dst, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (elem_bytes == 8) dst[i] = __float_as_uint((float)static_cast<const double *>(src)[i]);
    else dst[i] = static_cast<const uint32_t *>(src)[i];
}

// =========================================================================== probe: the bare row stream
// Measurement utility (zenv_probe_store_stream, bench.py's `store_stream_ceiling`): the tile flush of K1 / K1p with
// everything else taken away -- one wave per tile rewrites its tile's contiguous `tile_bytes` with 1 KiB
// buffer_store_dwordx4 bursts, `steps` times over, under the cache policy AUX (0 plain, 2 nt, 16 sc1 = write-through).
// What this reaches at a given footprint is the ceiling of a write-only stream of that shape on this box: the number
// the step kernels' row stores are held against (the float4-COPY figure of MI355X_MICROARCH.md reads AND writes, so
// it is not a ceiling for a stream that only writes).
template <int AUX>
__global__ __launch_bounds__(kWave) void k_probe_store(float *out, int tile_bytes, int steps)
{
    const int lane = threadIdx.x;
    float *dst = out + (size_t)blockIdx.x * (size_t)(tile_bytes / 4);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(dst, 0, tile_bytes, 0x00020000);
    v4f_t val = { (float)lane, 1.f, 2.f, 3.f };
    for (int t = 0; t < steps; ++t) {
        val.y = (float)t;
        for (int off = lane * 16; off + 16 <= tile_bytes; off += 1024)
            __builtin_amdgcn_raw_buffer_store_b128(val, rsrc, off, 0, AUX);
    }
}

}  // namespace

hipError_t launch_bank_derive(const DevParams &p, const int32_t *slots, int count, hipStream_t s)
{
    const dim3 grid(count), block(kWave);
    switch (p.task) {
    case ZENV_TASK_TSP: hipLaunchKernelGGL(k_bank_derive<ZENV_TASK_TSP>, grid, block, 0, s, p, slots); break;
    case ZENV_TASK_TIMED_TSP: hipLaunchKernelGGL(k_bank_derive<ZENV_TASK_TIMED_TSP>, grid, block, 0, s, p, slots); break;
    default: hipLaunchKernelGGL(k_bank_derive<ZENV_TASK_COLOUR_MATCH>, grid, block, 0, s, p, slots); break;
    }
    return hipGetLastError();
}

hipError_t launch_bank_scatter(const DevParams &p, const int32_t *slots, const void *staging, int rec_bytes, int count,
                               hipStream_t s)
{
    hipLaunchKernelGGL(k_bank_scatter, dim3(count), dim3(64), 0, s, p, slots, static_cast<const unsigned char *>(staging),
                       rec_bytes);
    return hipGetLastError();
}

hipError_t launch_sched_sync(const DevParams &p, int restart, hipStream_t s)
{
    hipLaunchKernelGGL(k_sched_sync, dim3((p.N + 255) / 256), dim3(256), 0, s, p, restart);
    return hipGetLastError();
}

hipError_t launch_unpack_hot(const DevParams &p, double *ep_return, int32_t *steps, hipStream_t s)
{
    hipLaunchKernelGGL(k_unpack_hot, dim3((p.N + 255) / 256), dim3(256), 0, s, p, ep_return, steps);
    return hipGetLastError();
}

hipError_t launch_gather_prep(const void *src, int elem_bytes, void *dst, int n, hipStream_t s)
{
    hipLaunchKernelGGL(k_gather_prep, dim3((n + 255) / 256), dim3(256), 0, s, src, elem_bytes, static_cast<uint32_t *>(dst), n);
    return hipGetLastError();
}

hipError_t launch_probe_store(float *out, long long n_tiles, int tile_bytes, int steps, int aux, hipStream_t s,
                              hipEvent_t ev_start, hipEvent_t ev_stop)
{
    const dim3 grid((unsigned)n_tiles), block(kWave);
    switch (aux) {
    case 0: hipExtLaunchKernelGGL(k_probe_store<0>, grid, block, 0, s, ev_start, ev_stop, 0, out, tile_bytes, steps); break;
    case 2: hipExtLaunchKernelGGL(k_probe_store<2>, grid, block, 0, s, ev_start, ev_stop, 0, out, tile_bytes, steps); break;
    case 16: hipExtLaunchKernelGGL(k_probe_store<16>, grid, block, 0, s, ev_start, ev_stop, 0, out, tile_bytes, steps); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------- launchers
static inline int n_blocks(int n) { return (n + kWave - 1) / kWave; }
static inline size_t full_tile_bytes(const DevParams &p) { return (size_t)kWave * p.Z * p.F * sizeof(float); }
static inline size_t step_lds_bytes(const DevParams &p)
{
    const int G = p.F == 6 ? (ZENV_RPC6 * 6) / 4 : 7;
    return (size_t)kWave * p.Z * sizeof(float4) + (size_t)kWave * G * sizeof(float4) + 2 * kWave * sizeof(int) +
           kWave * sizeof(double2) + kWave * sizeof(int);       // + the pose / exception hand-over
}

template <int TASK>
static void launch_step_task(const DevParams &p, const float *actions, int auto_reset, const StepPolicy &pol,
                             hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop)
{
    const dim3 grid(n_blocks(p.N)), block(kStepThreads);
    const size_t lds = step_lds_bytes(p);
    // hipExtLaunchKernelGGL stamps ev_start/ev_stop with the dispatch's own begin/end
#define ZENV_LAUNCH(ZT)                                                                                   \
    hipExtLaunchKernelGGL((k_step_lane<TASK, ZT>), grid, block, lds, s, ev_start, ev_stop, 0, p, actions, \
                          auto_reset, pol)
    switch (p.Z) {
    case 5: ZENV_LAUNCH(5); break;
    case 6: ZENV_LAUNCH(6); break;
    case 10: ZENV_LAUNCH(10); break;
    case 15: ZENV_LAUNCH(15); break;
    case 20: ZENV_LAUNCH(20); break;
    case 25: ZENV_LAUNCH(25); break;
    default: ZENV_LAUNCH(0); break;
    }
#undef ZENV_LAUNCH
}

hipError_t launch_step(const DevParams &p, const float *actions, int auto_reset, const StepPolicy &pol,
                       hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop)
{
    if (p.kernel == ZENV_KERNEL_WAVE_PER_ENV) {
        // no fused policy in this layout: the caller runs K3 (zenv_api.cpp keeps pol.policy < 0 here)
        const dim3 grid((p.N + 3) / 4), block(4 * kWave);
        switch (p.task) {
        case ZENV_TASK_TSP:
            hipExtLaunchKernelGGL((k_step_wave<ZENV_TASK_TSP>), grid, block, 0, s, ev_start, ev_stop, 0, p, actions, auto_reset);
            break;
        case ZENV_TASK_TIMED_TSP:
            hipExtLaunchKernelGGL((k_step_wave<ZENV_TASK_TIMED_TSP>), grid, block, 0, s, ev_start, ev_stop, 0, p, actions, auto_reset);
            break;
        default:
            hipExtLaunchKernelGGL((k_step_wave<ZENV_TASK_COLOUR_MATCH>), grid, block, 0, s, ev_start, ev_stop, 0, p, actions, auto_reset);
            break;
        }
        return hipGetLastError();
    }
    switch (p.task) {
    case ZENV_TASK_TSP: launch_step_task<ZENV_TASK_TSP>(p, actions, auto_reset, pol, s, ev_start, ev_stop); break;
    case ZENV_TASK_TIMED_TSP: launch_step_task<ZENV_TASK_TIMED_TSP>(p, actions, auto_reset, pol, s, ev_start, ev_stop); break;
    default: launch_step_task<ZENV_TASK_COLOUR_MATCH>(p, actions, auto_reset, pol, s, ev_start, ev_stop); break;
    }
    return hipGetLastError();
}

static inline size_t rollout_lds_bytes(const DevParams &p)
{
    const int G = p.F == 6 ? (ZENV_RPC6 * 6) / 4 : 7;
    const int ZB = (p.Z + 3) & ~3;
    const size_t ent = p.task == ZENV_TASK_TIMED_TSP ? sizeof(float4) : sizeof(float2);
    return (size_t)kWave * p.Z * ent + (size_t)kWave * G * sizeof(float4)         // static entries, flush slab
           + 2 * kWave * sizeof(uint64_t)                                          // per-step words
           + (p.task == ZENV_TASK_COLOUR_MATCH ? 2 * (size_t)kWave * ZB : 0)       // cooldown bytes
           + 4 * sizeof(int)                                                       // counters
           + (p.task == ZENV_TASK_COLOUR_MATCH ? 4 * sizeof(float4) + 256 * sizeof(float) : 0)    // row tables
           + 2 * (2 * kWave + kWave) * sizeof(float4)                              // published obs + (reward, flags, count)
           + kWave * sizeof(float);                                                // action prefetch pad (EXT)
}   // (allocated for every kernel; the ones whose env wave issues its own stores leave it unused)

bool rollout_kernel_available(const DevParams &p)
{
    return p.kernel == ZENV_KERNEL_LANE_PER_ENV && (p.Z == 5 || p.Z == 6 || p.Z == 10 || p.Z == 15 || p.Z == 20 || p.Z == 25);
}

template <int TASK, bool EXT>
static void launch_rollout_task(const DevParams &p, int n_steps, int auto_reset, const StepPolicy &pol, hipStream_t s,
                                hipEvent_t ev_start, hipEvent_t ev_stop, int tile0, int n_tiles, const ChunkIO &io)
{
    const dim3 grid(n_tiles), block(2 * kWave);
    const size_t lds = rollout_lds_bytes(p);
#define ZENV_LAUNCH(ZT)                                                                                            \
    hipExtLaunchKernelGGL((k_rollout_lane<TASK, ZT, EXT>), grid, block, lds, s, ev_start, ev_stop, 0, p, n_steps, \
                          auto_reset, pol, tile0, io)
    switch (p.Z) {
    case 5: ZENV_LAUNCH(5); break;
    case 6: ZENV_LAUNCH(6); break;
    case 10: ZENV_LAUNCH(10); break;
    case 15: ZENV_LAUNCH(15); break;
    case 20: ZENV_LAUNCH(20); break;
    default: ZENV_LAUNCH(25); break;
    }
#undef ZENV_LAUNCH
}

int rollout_tiles(const DevParams &p) { return n_blocks(p.N); }

hipError_t launch_rollout(const DevParams &p, int n_steps, int auto_reset, const StepPolicy &pol, hipStream_t s,
                          hipEvent_t ev_start, hipEvent_t ev_stop, int tile0, int n_tiles)
{
    if (!rollout_kernel_available(p)) return hipErrorInvalidValue;
    if (n_tiles < 0) n_tiles = n_blocks(p.N) - tile0;
    if (tile0 < 0 || n_tiles < 1 || tile0 + n_tiles > n_blocks(p.N)) return hipErrorInvalidValue;
    const ChunkIO none{ nullptr, nullptr, nullptr };
    switch (p.task) {
    case ZENV_TASK_TSP: launch_rollout_task<ZENV_TASK_TSP, false>(p, n_steps, auto_reset, pol, s, ev_start, ev_stop, tile0, n_tiles, none); break;
    case ZENV_TASK_TIMED_TSP: launch_rollout_task<ZENV_TASK_TIMED_TSP, false>(p, n_steps, auto_reset, pol, s, ev_start, ev_stop, tile0, n_tiles, none); break;
    default: launch_rollout_task<ZENV_TASK_COLOUR_MATCH, false>(p, n_steps, auto_reset, pol, s, ev_start, ev_stop, tile0, n_tiles, none); break;
    }
    return hipGetLastError();
}

hipError_t launch_rollout_actions(const DevParams &p, int n_steps, int reset_mode, const ChunkIO &io, hipStream_t s,
                                  hipEvent_t ev_start, hipEvent_t ev_stop, int tile0, int n_tiles)
{
    if (!rollout_kernel_available(p) || !io.actions || !io.reward || !io.done) return hipErrorInvalidValue;
    if (n_tiles < 0) n_tiles = n_blocks(p.N) - tile0;
    if (tile0 < 0 || n_tiles < 1 || tile0 + n_tiles > n_blocks(p.N) || n_steps < 1) return hipErrorInvalidValue;
    if (reset_mode < 0 || reset_mode > 2) return hipErrorInvalidValue;
    const StepPolicy pol = no_policy();
    switch (p.task) {
    case ZENV_TASK_TSP: launch_rollout_task<ZENV_TASK_TSP, true>(p, n_steps, reset_mode, pol, s, ev_start, ev_stop, tile0, n_tiles, io); break;
    case ZENV_TASK_TIMED_TSP: launch_rollout_task<ZENV_TASK_TIMED_TSP, true>(p, n_steps, reset_mode, pol, s, ev_start, ev_stop, tile0, n_tiles, io); break;
    default: launch_rollout_task<ZENV_TASK_COLOUR_MATCH, true>(p, n_steps, reset_mode, pol, s, ev_start, ev_stop, tile0, n_tiles, io); break;
    }
    return hipGetLastError();
}

hipError_t launch_order_reset(const DevParams &p, const uint8_t *mask, hipStream_t s)
{
    hipLaunchKernelGGL(k_order_reset, dim3((p.N + 255) / 256), dim3(256), 0, s, p, mask);
    return hipGetLastError();
}

hipError_t launch_order_step(const DevParams &p, hipStream_t s)
{
    hipLaunchKernelGGL(k_order_step, dim3((p.N + 255) / 256), dim3(256), 0, s, p);
    return hipGetLastError();
}

hipError_t launch_goal_set(const DevParams &p, const int32_t *new_goal, int32_t *bad, hipStream_t s)
{
    hipLaunchKernelGGL(k_goal_set, dim3((p.N + 255) / 256), dim3(256), 0, s, p, new_goal, bad);
    return hipGetLastError();
}

hipError_t launch_solver_goal(const DevParams &p, int32_t *out, hipStream_t s)
{
    hipLaunchKernelGGL(k_solver_goal, dim3((p.N + 255) / 256), dim3(256), 0, s, p, out);
    return hipGetLastError();
}

hipError_t launch_goal_clear(const DevParams &p, const uint8_t *mask, hipStream_t s)
{
    hipLaunchKernelGGL(k_goal_clear, dim3((p.N + 255) / 256), dim3(256), 0, s, p, mask);
    return hipGetLastError();
}

hipError_t launch_goal_step(const DevParams &p, hipStream_t s)
{
    hipLaunchKernelGGL(k_goal_step, dim3((p.N + 255) / 256), dim3(256), 0, s, p);
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
    const size_t lds = pol.policy == ZENV_POLICY_GREEDY ? full_tile_bytes(p) : 0;
    switch (p.task) {
    case ZENV_TASK_TSP: hipLaunchKernelGGL(k_policy_lane<ZENV_TASK_TSP>, grid, block, lds, s, p, pol); break;
    case ZENV_TASK_TIMED_TSP: hipLaunchKernelGGL(k_policy_lane<ZENV_TASK_TIMED_TSP>, grid, block, lds, s, p, pol); break;
    default: hipLaunchKernelGGL(k_policy_lane<ZENV_TASK_COLOUR_MATCH>, grid, block, lds, s, p, pol); break;
    }
    return hipGetLastError();
}

}  // namespace zenvk

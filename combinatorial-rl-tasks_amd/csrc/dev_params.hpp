// dev_params.hpp -- POD handed by value to every kernel: model constants + HBM pointers.
//
// HBM layout (lane-per-env kernels): everything is struct-of-arrays over the env index so
// that lane i of a wave touches element i of each array (64 x 8 B = 512 B per wave-load):
//   dyn      qa,qb,qc [N] double2       float64   (q0,q1) (q2,v0) (v1,v2) joint pos / vel
//   frame    fa,fb [N] double2          float64   (x0,y0) (bq0,bq3) episode-static placement
//   zones    zxy[Z][N] double2          float64   zone centres (x,y), zone-major
// (pairs so that every access is 16 B per lane = 1 KiB per wave instruction)
//   hot      hota[N] (ep_return, steps, visited mask | goal_dist), hotc[N] (colours, cooldown bytes of zones 0..7;
//            ColourMatch), sched[N] (next slot, episode index, reset hint, done state): 16-byte records, see below
//   TTSP     tmax[Z][N] i32
//   counters last_return[N] f64, episodes / last_len / visit_count[N] i32 ...
// Outputs are row-major exactly as the reference consumer reads them
// (main/src/utils/format.py:25-29): obs [N][8] f32, zone_obs [N][Z][F] f32.
#pragma once
#include <cstdint>

#include <hip/hip_runtime_api.h>   // double2

namespace zenvk {

// The per-env scalars every step reads and writes, as 16-byte records (struct-of-arrays over the env index, one
// dwordx4 per lane): a step kernel loads and stores three records where it used ten 1- to 8-byte arrays -- fewer
// instructions on the two lone waves of a tile, fewer live pointers, and every access a full-width one.
struct __attribute__((aligned(16))) HotA {
    double ep_return;      // undiscounted return of the running episode
    int32_t steps;         // steps of the running episode
    uint32_t vis;          // TSP / TimedTSP: visited bitmask (starts as vis0); ColourMatch: goal_dist
};
struct __attribute__((aligned(16))) HotC {     // ColourMatch only
    uint64_t colpack;      // 2-bit colours
    uint64_t cd0;          // cooldown bytes of zones 0..7 (zones 8.. in DevParams::cooldown)
};
struct __attribute__((aligned(16))) Sched {
    int32_t next_slot;     // sequential / ring schedules: bank slot of the env's NEXT episode (slot_after, kernels.hip)
    int32_t episode_idx;   // episodes started so far
    int32_t reset_hint;    // bank slot the env may reset into at the next step (-1: none): prefetch hint, not state
    uint32_t done_state;   // 1 = finished and left alone by step_no_reset (WaitWrapper's no-op until an auto-reset step)
};

struct DevParams {
    // sizes / task
    int32_t task, Z, F, N;
    int32_t num_steps, max_cd, frameskip, bank_size;
    int32_t sched_mode, sched_stride;
    int32_t kernel;       // ZENV_KERNEL_*
    uint32_t vis0;        // zones that start an episode visited (TSPHardEnv 'zones_colours'), bit z = zone z
    int64_t seed_min, seed_max;
    // model constants (derived on the host once; see zenv_api.cpp:derive_constants)
    double h, gear, fmax, kv, mc;
    double b0, b1, b2;
    double A22, inv00, inv11;
    double inv_den;       // 1 / (A22 - mc^2 inv00): the hinge row's Schur complement when b0 == b1 (iso)
    double kvg;           // vel_kv * gear
    int32_t iso;
    int32_t order_fresh;  // solver-ordered variant: 1 = an episode's first observation shows its own route (opt-out, see K7)
    double hit_d2;        // largest d2 with sqrt(d2) <= zones_size
    float d2_lo, d2_hi;   // float32 prefilter shell around zones_size^2 (see kernels.hip)
    float d2_near, pad_near;   // (zones_size + what the robot can travel in one env step)^2: "a zone could be hit at the NEXT step"
    double tsr;           // time_saved_reward
    double reward_exc;    // Engine 'reward_exception'
    double inv3, inv1_5;  // RN(1/3), RN(1/1.5)
    double d_steps, inv_steps;   // (double)num_steps and RN(1/num_steps)
    double d_maxcd, inv_maxcd;   // (double)max_cd and RN(1/max_cd)
    // state
    double2 *qa, *qb, *qc;   // (q0,q1) (q2,v0) (v1,v2): 16 B/lane accesses
    double2 *fa, *fb;        // (x0,y0) (bq0,bq3)
    double2 *zxy;            // [Z][N] (zone x, zone y): exact centres (rim test, resets)
    float4 *zpf;             // [ceil(Z/2)][N] float32 (x/3, y/3) of zones 2h and 2h+1
    HotA *hota;              // [N]
    HotC *hotc;              // [N] (ColourMatch)
    Sched *sched;            // [N]
    int32_t *tmax;
    uint8_t *cooldown;       // ColourMatch, zones 8 and up: one byte per zone, eight zones of an env per 64-bit word:
                             // u64 [ceil(Z/8) - 1][N] (word w of an env = zones 8w .. 8w+7; word 0 is HotC::cd0)
    double *last_return;
    int32_t *last_len, *episodes, *visit_count;
    // goal-conditioned variant (TSP_next_city_env.py), null unless zenv_goal_enable(): the zone visited this
    // step (-1: none) and the post-physics world position of an env whose episode ended in this step
    int32_t *visit_zone;
    double2 *term_xy;
    int32_t *goal;          // goal zone, -1 = none (needs one)
    double *goal_last;      // last_dist_to_goal
    double2 *goal_xy;       // the goal zone's centre (survives the auto-reset of the zone arrays)
    double *shaped;         // info['shaped_reward']
    uint8_t *need_goal;     // info['need_next_goal']
    uint32_t *available;    // get_available_goals() as a bit mask
    // solver-ordered variant (TSP_order_env.py), null unless zenv_order_enable(): position of every zone in
    // the remaining route (-1: visited) and the observation's order feature 0.5^position
    int8_t *order_pos;      // [N][Z]
    float *order_val;       // [N][Z]
    int64_t *seed;
    // schedule
    int32_t *slot_first;
    uint64_t *pcg;        // [N][4] state_hi,state_lo,inc_hi,inc_lo
    uint32_t *pcg_buf;    // [N][2] has_u32,u32
    // bank
    const double *bank_robot;   // [S][4] x0,y0,bq0,bq3
    const double *bank_zone;    // [S][Z][2]
    const int32_t *bank_aux;    // [S][Z]
    const int64_t *bank_seed;   // [S]
    const float4 *bank_first;   // [S][3] derived: first obs (8 floats), the greedy policy's first action (2), pad
    // outputs
    float *obs, *zone_obs, *reward, *actions;
    uint8_t *done_out, *goal_met;
    uint8_t *exception;   // info['exception'] of the last finished episode (written when an episode ends)
    unsigned long long *dbg;   // diagnostic stamps (ZENV_STAMPS builds), else null
    const DevParams *self;     // device copy of this block (kept current by the host): cold fields are read through it
};

enum { SCHED_SEQUENTIAL = 0, SCHED_FIXED_SEEDS = 1, SCHED_RING = 2 };   // RING: sched_stride = ring depth

}  // namespace zenvk

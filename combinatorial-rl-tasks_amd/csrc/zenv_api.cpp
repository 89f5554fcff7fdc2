// zenv_api.cpp -- the C ABI of include/zenv.h over the gfx950 kernels.
// Host-side plumbing only: device memory, the layout bank, launches, copies.  There is no
// CPU compute path here: without a usable HIP device every compute entry point fails.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>   // types and prototypes only: librccl is dlopen()ed by the first zenv_comm_* call

#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/zenv.h"
#include "dev_params.hpp"
#include "host_sampler.hpp"
#include "kernels.hpp"
#include "mlp_policy.hpp"

using namespace zenvk;

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess)                                                                 \
            return fail(ZENV_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),    \
                        __FILE__, __LINE__);                                                  \
    } while (0)

struct Alloc {
    void **slot;
    size_t bytes;
    bool is_state;   // part of zenv_get_state/zenv_set_state
    int64_t slab_off = -1;   // >= 0: lives at this offset of the handle's results slab (one hipMalloc, one download)
};

}  // namespace

struct zenv {
    zenv_config cfg{};
    int n_env = 0;
    int device = 0;
    hipStream_t stream = nullptr;       // the stream work is enqueued on
    hipStream_t own_stream = nullptr;   // created with the handle; `stream` unless zenv_set_stream
    DevParams p{};
    DevParams *d_self = nullptr;     // device copy of p (DevParams::self): the persistent kernel reads cold fields from it
    DevParams self_shadow{};         // what d_self holds
    bool self_valid = false;
    std::vector<Alloc> allocs;
    // The per-step results (obs, reward, done, goal_met, exception, zone_obs) share ONE allocation, 256-byte aligned
    // pieces in that order, so that a host policy fetches them with one copy (zenv_step_results).
    void *results_slab = nullptr;
    // zenv_host_io(): results slab + action buffer in page-locked host memory that the kernels write / read directly
    void *host_io_slab = nullptr;
    float *host_io_actions = nullptr, *dev_actions = nullptr;
    int64_t results_off[ZENV_N_RESULTS] = {};
    int64_t results_bytes = 0;
    void *bank_mem[5] = { nullptr, nullptr, nullptr, nullptr, nullptr };   // robot, zone, aux, seed, derived first rows
    uint8_t *d_mask = nullptr;
    bool bank_ready = false;
    bool sched_ready = false;
    bool was_reset = false;
    int64_t step_count = 0;
    // What the internal action buffer holds when a fused rollout left a_{step_count} behind: the next rollout with
    // the same action source continues from it instead of launching the policy kernel again.
    struct {
        bool valid = false;
        int policy = -1;
        uint64_t seed = 0, index0 = 0;
        int64_t step = -1;
    } act_tag;
    std::vector<hipEvent_t> events;
    int rollout_slice_tiles = 1024;   // 64-env tiles per persistent launch (zenv_set_rollout_slice): one per wave slot pair
    // actor network (zenv_mlp_load)
    void *mlp_mem = nullptr;
    MlpImages mlp{};
    void *mlp_pooled = nullptr;
    float *mlp_mu = nullptr, *mlp_std = nullptr, *mlp_value = nullptr, *mlp_value_sigma = nullptr;
    int *mlp_range_flag = nullptr;      // pinned host word, see mlp_range_check()
    void *mlp_f32_mem = nullptr;        // float32 path (ZENV_MLP_F32): transposed float32 weights
    MlpF32 mlp_f32{};
    bool mlp_ready = false;
    // goal-conditioned variant (zenv_goal_enable)
    bool goal_enabled = false;
    bool order_enabled = false;   // solver-ordered variant (zenv_order_enable)
    int32_t *goal_in = nullptr, *goal_bad = nullptr;
    // experience buffers (zenv_collect)
    ExpBuffers exp{};
    void *exp_mem = nullptr;
    // staging of zenv_bank_update (page-locked host image + its device copy)
    void *refill_host = nullptr, *refill_dev = nullptr;
    size_t refill_cap = 0;
    hipEvent_t refill_done = nullptr;
    bool refill_busy = false;
    // the sharded job's communicator (zenv_comm_init): RCCL over xGMI
    ncclComm_t comm = nullptr;
    int comm_rank = 0, comm_world = 0;
    void *comm_send = nullptr, *comm_recv = nullptr;   // [N] and [world * N] 4-byte elements
    double *comm_scalar = nullptr;                      // device scratch of the barrier / max-reduce
    // zenv_step_many: the chunk's actions on the device (when they came from the host) and its time-major records
    void *chunk_mem = nullptr;
    size_t chunk_cap = 0;           // steps * envs the allocation holds
    float *chunk_actions = nullptr, *chunk_reward = nullptr;
    uint8_t *chunk_done = nullptr;
    int chunk_steps = 0;            // steps of the last zenv_step_many (extent of ZENV_F_CHUNK_*)
    // ZENV_F_EP_RETURN / ZENV_F_EP_LEN as plain arrays: the values live in the HotA records, unpacked by refresh_field()
    double *pub_ep_return = nullptr;
    int32_t *pub_steps = nullptr;
};

namespace {

int validate_config(const zenv_config &c)
{
    if (c.task < ZENV_TASK_TSP || c.task > ZENV_TASK_COLOUR_MATCH) return fail(ZENV_E_ARG, "unknown task %d", c.task);
    if (c.num_zones < 1 || c.num_zones > ZENV_MAX_ZONES)
        return fail(ZENV_E_ARG, "num_zones %d outside [1,%d]", c.num_zones, ZENV_MAX_ZONES);
    if (c.num_steps < 1) return fail(ZENV_E_ARG, "num_steps must be positive");
    if (c.max_cd < 1 || c.max_cd > 255) return fail(ZENV_E_ARG, "max_cd %d outside [1,255]", c.max_cd);
    if (c.frameskip < 1 || c.frameskip > 1000) return fail(ZENV_E_ARG, "bad frameskip %d", c.frameskip);
    if (!(c.mass > 0) || !(c.inertia_zz > 0) || !(c.timestep > 0)) return fail(ZENV_E_ARG, "bad model constants");
    if (c.kernel != ZENV_KERNEL_LANE_PER_ENV && c.kernel != ZENV_KERNEL_WAVE_PER_ENV)
        return fail(ZENV_E_ARG, "unknown kernel layout %d", c.kernel);
    // The step kernels treat "MuJoCo could not simulate this step" (Engine.step's MujocoException path) as
    // "the action holds a NaN".  That is exact while a finite state and a finite control cannot produce a bad qacc
    // (NaN, Inf, |.| > 1e10: mj_checkAcc): every model constant finite, the mass matrix's Schur complement on the
    // hinge positive (I0 > m c^2, the parallel-axis bound of any physical body), forces far below 1e10 * mass.
    for (double v : { c.mass, c.com_x, c.inertia_zz, c.timestep, c.damping[0], c.damping[1], c.damping[2], c.gear,
                      c.forcerange, c.vel_kv, c.reward_exception, c.time_saved_reward, c.zones_size })
        if (!std::isfinite(v)) return fail(ZENV_E_ARG, "non-finite model constant");
    if (c.damping[0] < 0 || c.damping[1] < 0 || c.damping[2] < 0 || c.forcerange < 0)
        return fail(ZENV_E_ARG, "damping and forcerange must be non-negative");
    {
        const double mc = c.mass * c.com_x, bmin = std::min(c.damping[0], c.damping[1]);
        const double schur = c.inertia_zz + c.timestep * c.damping[2] - mc * mc / (c.mass + c.timestep * bmin);
        if (!(schur > 1e-12 * c.inertia_zz)) return fail(ZENV_E_ARG, "inertia_zz must exceed mass * com_x^2");
        if (!(std::fabs(c.gear) * c.forcerange < 1e6 * std::min(c.mass, schur)))
            return fail(ZENV_E_ARG, "actuator force out of proportion to the body's mass / inertia");
        // The kernels turn sin/cos of the hinge angle by d = h * omega per substep with Taylor kernels good to
        // |d| = 0.1.  |omega| stays below the servo's terminal speed plus the overshoot of its chatter (two
        // force-clamped substeps): keep that bound under 0.05 rad per substep (point.xml: 0.0093).
        const double torque = std::fabs(c.gear) * c.forcerange;
        const double w_term = c.damping[2] > 0 ? torque / c.damping[2]
                                               : (c.vel_kv * std::fabs(c.gear) > 0 ? 1.0 / std::fabs(c.gear) : 1e30);
        const double w_servo = c.vel_kv * std::fabs(c.gear) > 0 ? 1.0 / std::fabs(c.gear) : w_term;   // |ctrl| <= 1
        const double w_max = std::min(w_term, w_servo) + 2.0 * c.timestep * torque / schur;
        if (!(c.timestep * w_max < 0.05))
            return fail(ZENV_E_ARG, "the hinge can turn %.3g rad per substep: beyond the small-angle update (0.05)",
                        c.timestep * w_max);
    }
    if (c.n_zones_locations < 0 || c.n_zones_locations > c.num_zones)
        return fail(ZENV_E_ARG, "n_zones_locations %d outside [0, num_zones]", c.n_zones_locations);
    if (c.n_robot_locations < 0 || c.n_robot_locations > 1)
        return fail(ZENV_E_ARG, "n_robot_locations must be 0 or 1");
    if (c.num_zones < 32 && (c.visited0 >> c.num_zones) != 0u) return fail(ZENV_E_ARG, "visited0 names a zone >= num_zones");
    if (c.visited0 && c.task == ZENV_TASK_COLOUR_MATCH) return fail(ZENV_E_ARG, "visited0 is a TSP / TimedTSP setting");
    if (c.visited0 && (c.num_zones >= 32 ? c.visited0 == 0xFFFFFFFFu : c.visited0 == (1u << c.num_zones) - 1u))
        return fail(ZENV_E_ARG, "visited0 leaves no zone to visit");
    return ZENV_OK;
}

void derive_constants(const zenv_config &c, DevParams &p)
{
    p.task = c.task;
    p.kernel = c.kernel;
    p.vis0 = c.visited0;
    p.Z = c.num_zones;
    p.F = zenv_zone_feat(&c);
    p.num_steps = c.num_steps;
    p.max_cd = c.max_cd;
    p.frameskip = c.frameskip;
    p.h = c.timestep;
    p.gear = c.gear;
    p.fmax = c.forcerange;
    p.kv = c.vel_kv;
    p.mc = c.mass * c.com_x;
    p.b0 = c.damping[0];
    p.b1 = c.damping[1];
    p.b2 = c.damping[2];
    const double A00 = c.mass + c.timestep * c.damping[0];
    const double A11 = c.mass + c.timestep * c.damping[1];
    p.A22 = c.inertia_zz + c.timestep * c.damping[2];
    p.inv00 = 1.0 / A00;
    p.inv11 = 1.0 / A11;
    p.iso = c.damping[0] == c.damping[1];
    p.order_fresh = 0;
    p.inv_den = 1.0 / std::fma(-(p.mc * p.mc), p.inv00, p.A22);
    p.kvg = c.vel_kv * c.gear;
    p.hit_d2 = sqrt_threshold(c.zones_size);
    {
        // prefilter shell: +-2e-6 on the radius, i.e. +-8e-7 on r^2 at r = 0.2 (4x the float error bound)
        const double lo = (c.zones_size - 2e-6) * (c.zones_size - 2e-6), hi = (c.zones_size + 2e-6) * (c.zones_size + 2e-6);
        p.d2_lo = std::nextafterf((float)lo, 0.f);
        p.d2_hi = std::nextafterf((float)hi, INFINITY);
    }
    {
        // reset-prefetch hint (k_step_lane): a zone can only be hit at the NEXT step if the robot is within its radius +
        // one env step of travel NOW.  Terminal speed gear * forcerange / damping, 1.5x for transients; undamped: always near
        const double b = std::min(c.damping[0], c.damping[1]);
        const double v = b > 0 ? std::fabs(c.gear) * c.forcerange / b : INFINITY;
        const double reach = c.zones_size + 1.5 * v * c.timestep * c.frameskip + 1e-3;
        p.d2_near = std::isfinite(reach) && reach < 1e3 ? (float)(reach * reach) : INFINITY;
        p.pad_near = 0.f;
    }
    p.tsr = c.time_saved_reward;
    p.reward_exc = c.reward_exception;
    p.inv3 = 1.0 / 3.0;
    p.inv1_5 = 1.0 / 1.5;
    p.d_steps = (double)c.num_steps;
    p.inv_steps = 1.0 / p.d_steps;
    p.d_maxcd = (double)c.max_cd;
    p.inv_maxcd = 1.0 / p.d_maxcd;
}

template <typename T>
void want(zenv *h, T *&slot, size_t count, bool is_state)
{
    h->allocs.push_back({ reinterpret_cast<void **>(&slot), count * sizeof(T), is_state });
}

int use_device(const zenv *h)
{
    HIP_TRY(hipSetDevice(h->device));
    return ZENV_OK;
}

struct FieldInfo {
    void *ptr;
    int64_t bytes;
};

FieldInfo field_info(const zenv *h, int field)
{
    const DevParams &p = h->p;
    const int64_t N = h->n_env;
    switch (field) {
    case ZENV_F_OBS: return { p.obs, N * 8 * 4 };
    case ZENV_F_ZONE_OBS: return { p.zone_obs, N * p.Z * p.F * 4 };
    case ZENV_F_REWARD: return { p.reward, N * 4 };
    case ZENV_F_DONE: return { p.done_out, N };
    case ZENV_F_GOAL_MET: return { p.goal_met, N };
    case ZENV_F_EP_RETURN: return { h->pub_ep_return, N * 8 };      // (refresh_field() first)
    case ZENV_F_EP_LEN: return { h->pub_steps, N * 4 };
    case ZENV_F_LAST_RETURN: return { p.last_return, N * 8 };
    case ZENV_F_LAST_LEN: return { p.last_len, N * 4 };
    case ZENV_F_EPISODES: return { p.episodes, N * 4 };
    case ZENV_F_VISIT_COUNT: return { p.visit_count, N * 4 };
    case ZENV_F_SEED: return { p.seed, N * 8 };
    case ZENV_F_ACTIONS: return { p.actions, N * 2 * 4 };
    case ZENV_F_POLICY_MU: return { h->mlp_mu, h->mlp_mu ? N * 2 * 4 : 0 };
    case ZENV_F_POLICY_STD: return { h->mlp_std, h->mlp_std ? N * 2 * 4 : 0 };
    case ZENV_F_POLICY_VALUE: return { h->mlp_value, h->mlp_value ? N * 4 : 0 };
    case ZENV_F_POLICY_VALUE_SIGMA: return { h->mlp_value_sigma, h->mlp_value_sigma ? N * 4 : 0 };
    case ZENV_F_SHAPED_REWARD: return { p.shaped, p.shaped ? N * 8 : 0 };
    case ZENV_F_NEED_GOAL: return { p.need_goal, p.need_goal ? N : 0 };
    case ZENV_F_AVAILABLE_GOALS: return { p.available, p.available ? N * 4 : 0 };
    case ZENV_F_GOAL: return { p.goal, p.goal ? N * 4 : 0 };
    case ZENV_F_EXP_OBS: return { h->exp.obs, h->exp.obs ? N * h->exp.T * 8 * 4 : 0 };
    case ZENV_F_EXP_ZONE_OBS: return { h->exp.zone_obs, h->exp.obs ? N * h->exp.T * p.Z * p.F * 4 : 0 };
    case ZENV_F_EXP_ACTION: return { h->exp.action, h->exp.obs ? N * h->exp.T * 2 * 4 : 0 };
    case ZENV_F_EXP_LOG_PROB: return { h->exp.log_prob, h->exp.obs ? N * h->exp.T * 2 * 4 : 0 };
    case ZENV_F_EXP_VALUE: return { h->exp.value, h->exp.obs ? N * h->exp.T * 4 : 0 };
    case ZENV_F_EXP_REWARD: return { h->exp.reward, h->exp.obs ? N * h->exp.T * 4 : 0 };
    case ZENV_F_EXP_MASK: return { h->exp.mask, h->exp.obs ? N * h->exp.T * 4 : 0 };
    case ZENV_F_EXP_ADVANTAGE: return { h->exp.advantage, h->exp.obs ? N * h->exp.T * 4 : 0 };
    case ZENV_F_EXP_RETURN: return { h->exp.returnn, h->exp.obs ? N * h->exp.T * 4 : 0 };
    case ZENV_F_ORDER_VAL: return { p.order_val, p.order_val ? N * p.Z * 4 : 0 };
    case ZENV_F_ORDER_POS: return { p.order_pos, p.order_pos ? N * p.Z : 0 };
    case ZENV_F_CHUNK_REWARD: return { h->chunk_reward, h->chunk_reward ? N * h->chunk_steps * 4 : 0 };
    case ZENV_F_CHUNK_DONE: return { h->chunk_done, h->chunk_done ? N * h->chunk_steps : 0 };
    case ZENV_F_CHUNK_ACTIONS: return { h->chunk_actions, h->chunk_actions ? N * h->chunk_steps * 8 : 0 };
    case ZENV_F_EXCEPTION: return { p.exception, N };
    default: return { nullptr, 0 };
    }
}

// ZENV_F_EP_RETURN / ZENV_F_EP_LEN are kept inside the step kernels' 16-byte records: bring the plain arrays a caller
// sees up to date (stream-ordered, behind every step already enqueued)
int refresh_field(zenv *h, int field)
{
    if (field != ZENV_F_EP_RETURN && field != ZENV_F_EP_LEN) return ZENV_OK;
    HIP_TRY(launch_unpack_hot(h->p, h->pub_ep_return, h->pub_steps, h->stream));
    return ZENV_OK;
}

}  // namespace

// ============================================================================ host-only API
extern "C" const char *zenv_last_error(void) { return g_err.c_str(); }

// The -D / -m switches the library was compiled with beyond build.py's fixed set (ZENV_EXTRA_FLAGS: diagnostic
// variants such as -DZENV_EXP=1, -DZENV_STORE_AUX=16, -DZENV_STAMPS); "" for the shipped build.  build.py passes the
// string as -DZENV_BUILD_FLAGS, so a variant library names itself and bench.py can refuse to time it unasked.
#ifndef ZENV_BUILD_FLAGS
#define ZENV_BUILD_FLAGS ""
#endif
extern "C" const char *zenv_build_flags(void) { return ZENV_BUILD_FLAGS; }

extern "C" const char *zenv_version(void)
{
    static const std::string v = std::string("zenv-hip 0.4 (gfx950)") +
                                 (ZENV_BUILD_FLAGS[0] ? std::string(" [variant: ") + ZENV_BUILD_FLAGS + "]" : std::string());
    return v.c_str();
}

extern "C" int zenv_device_count(void)
{
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

extern "C" int zenv_zone_feat(const zenv_config *cfg)
{
    if (!cfg) return 0;
    return cfg->task == ZENV_TASK_TSP ? 6 : 7;   // TSP_env.py:27-29, TTSP_env.py:78-84, colour_match_env.py:70-73
}

extern "C" int zenv_config_size(void) { return (int)sizeof(zenv_config); }

extern "C" int zenv_default_config(int task, int num_zones, zenv_config *c)
{
    if (!c) return fail(ZENV_E_ARG, "null config");
    std::memset(c, 0, sizeof(*c));
    c->task = task;
    c->num_zones = num_zones;
    c->num_steps = 2000;
    c->max_cd = 150;
    c->frameskip = 10;
    c->kernel = ZENV_KERNEL_LANE_PER_ENV;
    c->zones_size = 0.2;
    c->zones_keepout = 0.55;
    c->robot_keepout = 0.4;
    c->extent = 3.0;
    c->placements_margin = 0.0;
    c->time_saved_reward = 0.01;
    c->beta_a = 3.0;
    c->beta_b = 1.5;
    // xmls/point.xml: sphere r=0.1 at the body origin + box half-size 0.05 at (0.1,0,0), density 1
    const double pi = 3.14159265358979323846, density = 1.0;
    const double m_sphere = density * (4.0 / 3.0 * pi * 0.1 * 0.1 * 0.1);
    const double m_box = density * (8.0 * 0.05 * 0.05 * 0.05);
    c->timestep = 0.002;
    c->mass = m_sphere + m_box;
    c->com_x = 0.1 * m_box / c->mass;
    c->inertia_zz = 0.4 * m_sphere * (0.1 * 0.1) + m_box * (0.05 * 0.05 + 0.05 * 0.05) / 3.0 + m_box * (0.1 * 0.1);
    c->damping[0] = 0.01;
    c->damping[1] = 0.01;
    c->damping[2] = 0.005;
    c->gear = 0.3;
    c->forcerange = 0.05;
    c->vel_kv = 1.0;
    c->reward_exception = -10.0;   // [not vendored] Engine.DEFAULT
    return validate_config(*c);
}

extern "C" int zenv_config_for_id(const char *env_id, zenv_config *out)
{
    if (!env_id || !out) return fail(ZENV_E_ARG, "null argument");
    struct Entry {
        const char *id;
        int task, zones, steps;
    };
    // envs/__init__.py:88-141 with config_point / config_point_easy / config_point_colour
    static const Entry table[] = {
        { "PointTSP-v0", ZENV_TASK_TSP, 15, 2000 },        { "PointTSP-v1", ZENV_TASK_TSP, 5, 1000 },
        { "PointTTSP-v0", ZENV_TASK_TIMED_TSP, 15, 2000 }, { "PointTTSP-v1", ZENV_TASK_TIMED_TSP, 5, 1000 },
        { "ColourMatch-v0", ZENV_TASK_COLOUR_MATCH, 6, 2000 },
        { "PointTSP-v4", ZENV_TASK_TSP, 15, 1000 },        { "PointTSP-v5", ZENV_TASK_TSP, 15, 250 },
    };
    for (const Entry &e : table) {
        if (std::strcmp(e.id, env_id) == 0) {
            int rc = zenv_default_config(e.task, e.zones, out);
            if (rc) return rc;
            out->num_steps = e.steps;
            // TSPHardEnv: config_zone_fixed_1 / _2 (envs/__init__.py:52-81)
            if (std::strcmp(env_id, "PointTSP-v4") == 0) {
                static const double loc[5][2] = { { -2.6, -1.6 }, { -0., -0.5 }, { 1., 0.5 }, { 1.8, 1.5 }, { 2.6, 2.6 } };
                out->n_zones_locations = 5;
                std::memcpy(out->zones_locations, loc, sizeof(loc));
                out->visited0 = 0x7FE0u;                 // 'zones_colours': [6] * 5 + [5] * 10 (Cyan, then Yellow)
                out->n_robot_locations = 1;
                out->robot_location[0] = -0.9;
                out->robot_location[1] = -0.9;
                out->robot_rot_fixed = 1;
                out->robot_rot = -1.0;
            } else if (std::strcmp(env_id, "PointTSP-v5") == 0) {
                static const double loc[3][2] = { { -2.6, -2.6 }, { -2, -1.6 }, { 2, 1 } };
                out->n_zones_locations = 3;
                std::memcpy(out->zones_locations, loc, sizeof(loc));
                out->visited0 = 0x7FF8u;                 // [6] * 3 + [5] * 12
                out->n_robot_locations = 1;
                out->robot_location[0] = 0.8;
                out->robot_location[1] = 0.8;
            }
            return validate_config(*out);
        }
    }
    return fail(ZENV_E_ARG, "Unknown environment: %s", env_id);
}

extern "C" int zenv_sample_layout(const zenv_config *cfg, int64_t seed, double *robot_xyrot,
                                  double *zone_xy, int32_t *aux, int32_t *restarts)
{
    if (!cfg) return fail(ZENV_E_ARG, "null config");
    int rc = validate_config(*cfg);
    if (rc) return rc;
    if (seed < 0 || seed + 1 > 0xFFFFFFFFll) return fail(ZENV_E_ARG, "Seed must be between 0 and 2**32 - 1");
    Layout L;
    rc = sample_layout(*cfg, seed, L);
    if (rc) return fail(rc, "Failed to sample layout of objects (seed %lld)", (long long)seed);
    if (robot_xyrot) {
        robot_xyrot[0] = L.robot_x;
        robot_xyrot[1] = L.robot_y;
        robot_xyrot[2] = L.robot_rot;
    }
    for (int z = 0; z < cfg->num_zones; ++z) {
        if (zone_xy) {
            zone_xy[2 * z] = L.zone_xy[z][0];
            zone_xy[2 * z + 1] = L.zone_xy[z][1];
        }
        if (aux) aux[z] = L.aux[z];
    }
    if (restarts) *restarts = L.restarts;
    return ZENV_OK;
}

extern "C" int zenv_fixed_seed_sequence(uint64_t rng_seed, int64_t min_seed, int64_t max_seed, int count,
                                        int64_t *out)
{
    if (!out || count < 0 || max_seed < min_seed) return fail(ZENV_E_ARG, "bad argument");
    if ((uint64_t)(max_seed - min_seed) >= 0xFFFFFFFFull) return fail(ZENV_E_ARG, "seed range too wide");
    Pcg64State s = pcg64_from_seed(rng_seed);
    for (int i = 0; i < count; ++i) out[i] = pcg64_integers(s, min_seed, max_seed + 1);
    return ZENV_OK;
}

// ============================================================================ lifecycle
extern "C" int zenv_create(const zenv_config *cfg, int n_env, int device, zenv_t **out)
{
    if (!cfg || !out) return fail(ZENV_E_ARG, "null argument");
    *out = nullptr;
    int rc = validate_config(*cfg);
    if (rc) return rc;
    if (n_env < 1) return fail(ZENV_E_ARG, "n_env must be >= 1");
    {
        // the kernels index a handle's arrays with 32-bit element offsets: the largest one, zone_obs [N][Z][F], stays
        // below 2^29 floats (2 GiB).  That is 3.5 M envs at Z = 25 (tested at 1 M); a bigger batch takes several handles.
        const int64_t zone_floats = (int64_t)n_env * cfg->num_zones * (cfg->task == ZENV_TASK_TSP ? 6 : 7);
        if (zone_floats > ((int64_t)1 << 29))
            return fail(ZENV_E_ARG, "n_env %d x %d zones is beyond one handle's 32-bit indexing (zone_obs would hold %lld "
                        "floats, the limit is 2^29): split the batch over several handles", n_env, cfg->num_zones,
                        (long long)zone_floats);
    }
    int n_dev = 0;
    hipError_t e = hipGetDeviceCount(&n_dev);
    if (e != hipSuccess || n_dev < 1)
        return fail(ZENV_E_HIP, "no HIP device available (%s); this library has no CPU fallback",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (device < 0 || device >= n_dev) return fail(ZENV_E_ARG, "device %d outside [0,%d)", device, n_dev);
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(ZENV_E_HIP, "device %d is %s; this library ships gfx950 code only", device, prop.gcnArchName);

    zenv *h = new zenv();
    h->cfg = *cfg;
    h->n_env = n_env;
    h->device = device;
    derive_constants(*cfg, h->p);
    DevParams &p = h->p;
    p.N = n_env;
    p.sched_mode = SCHED_SEQUENTIAL;
    p.sched_stride = 0;
    const size_t N = n_env, Z = cfg->num_zones, F = p.F;

    want(h, p.qa, N, true); want(h, p.qb, N, true); want(h, p.qc, N, true);
    want(h, p.fa, N, true); want(h, p.fb, N, true);
    want(h, p.zxy, Z * N, true);
    want(h, p.zpf, ((Z + 1) / 2) * N, true);
    want(h, p.hota, N, true); want(h, p.hotc, N, true); want(h, p.sched, N, true);   // the hot per-env scalars: 16-byte records
    want(h, p.tmax, Z * N, true);
    want(h, p.cooldown, std::max<size_t>(1, (Z + 7) / 8 - 1) * 8 * N, true);      // ColourMatch zones 8 and up (zones 0..7: HotC::cd0)
    want(h, p.last_return, N, true);
    want(h, p.last_len, N, true); want(h, p.episodes, N, true); want(h, p.visit_count, N, true);
    want(h, p.seed, N, true);
    want(h, p.slot_first, N, true);
    want(h, p.pcg, 4 * N, true); want(h, p.pcg_buf, 2 * N, true);
    want(h, p.actions, 2 * N, true);
    {   // results slab, in ZENV_RESULT_* order
        const size_t first = h->allocs.size();
        want(h, p.obs, 8 * N, true); want(h, p.reward, N, true); want(h, p.done_out, N, true);
        want(h, p.goal_met, N, true); want(h, p.exception, N, true); want(h, p.zone_obs, Z * F * N, true);
        int64_t off = 0;
        for (int i = 0; i < ZENV_N_RESULTS; ++i) {
            h->allocs[first + i].slab_off = h->results_off[i] = off;
            off += (int64_t)((h->allocs[first + i].bytes + 255) / 256 * 256);
        }
        h->results_bytes = off;
    }
    want(h, p.dbg, 16 * ((N + 63) / 64), false);
    want(h, h->pub_ep_return, N, false); want(h, h->pub_steps, N, false);   // ZENV_F_EP_RETURN / _EP_LEN, unpacked on request

    hipError_t err = hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking);
    h->stream = h->own_stream;
    if (err != hipSuccess) {
        delete h;
        return fail(ZENV_E_HIP, "hipStreamCreate: %s", hipGetErrorString(err));
    }
    err = hipMalloc(&h->results_slab, (size_t)h->results_bytes);
    if (err == hipSuccess) err = hipMemsetAsync(h->results_slab, 0, (size_t)h->results_bytes, h->stream);
    if (err != hipSuccess) {
        zenv_destroy(h);
        return fail(ZENV_E_HIP, "hipMalloc(%lld): %s", (long long)h->results_bytes, hipGetErrorString(err));
    }
    for (Alloc &a : h->allocs) {
        if (a.slab_off >= 0) {
            *a.slot = static_cast<char *>(h->results_slab) + a.slab_off;
            continue;
        }
        err = hipMalloc(a.slot, a.bytes);
        if (err == hipSuccess) err = hipMemsetAsync(*a.slot, 0, a.bytes, h->stream);
        if (err != hipSuccess) {
            zenv_destroy(h);
            return fail(ZENV_E_HIP, "hipMalloc(%zu): %s", a.bytes, hipGetErrorString(err));
        }
    }
    err = hipMalloc(reinterpret_cast<void **>(&h->d_mask), N);
    if (err == hipSuccess) err = hipMalloc(reinterpret_cast<void **>(&h->d_self), sizeof(DevParams));
    if (err == hipSuccess) p.self = h->d_self;
    if (err == hipSuccess) err = hipStreamSynchronize(h->stream);
    if (err != hipSuccess) {
        zenv_destroy(h);
        return fail(ZENV_E_HIP, "init: %s", hipGetErrorString(err));
    }
    *out = h;
    return ZENV_OK;
}

extern "C" int zenv_destroy(zenv_t *h)
{
    if (!h) return ZENV_OK;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->comm) (void)zenv_comm_destroy(h);
    if (h->host_io_actions) h->p.actions = h->dev_actions;       // the slot below is the device buffer again
    for (Alloc &a : h->allocs)
        if (*a.slot && a.slab_off < 0) (void)hipFree(*a.slot);
    if (h->results_slab) (void)hipFree(h->results_slab);
    if (h->host_io_slab) (void)hipHostFree(h->host_io_slab);
    if (h->host_io_actions) (void)hipHostFree(h->host_io_actions);
    for (void *m : h->bank_mem)
        if (m) (void)hipFree(m);
    if (h->d_mask) (void)hipFree(h->d_mask);
    if (h->d_self) (void)hipFree(h->d_self);
    if (h->chunk_mem) (void)hipFree(h->chunk_mem);
    if (h->refill_host) (void)hipHostFree(h->refill_host);
    if (h->mlp_range_flag) (void)hipHostFree(h->mlp_range_flag);
    if (h->refill_dev) (void)hipFree(h->refill_dev);
    if (h->refill_done) (void)hipEventDestroy(h->refill_done);
    for (void *m : { h->mlp_mem, h->mlp_f32_mem, (void *)h->mlp_value_sigma, h->mlp_pooled, (void *)h->mlp_mu,
                     (void *)h->mlp_std, (void *)h->mlp_value,
                     (void *)h->p.visit_zone, (void *)h->p.term_xy, (void *)h->p.goal, (void *)h->p.goal_last,
                     (void *)h->p.goal_xy, (void *)h->p.shaped, (void *)h->p.need_goal, (void *)h->p.available,
                     (void *)h->goal_in, (void *)h->goal_bad, h->exp_mem, (void *)h->p.order_pos,
                     (void *)h->p.order_val })
        if (m) (void)hipFree(m);
    for (hipEvent_t ev : h->events) (void)hipEventDestroy(ev);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
    return ZENV_OK;
}

extern "C" int zenv_set_stream(zenv_t *h, void *hip_stream)
{
    if (!h) return fail(ZENV_E_ARG, "null handle");
    int rc = use_device(h);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : h->own_stream;
    return ZENV_OK;
}

extern "C" int zenv_num_envs(const zenv_t *h) { return h ? h->n_env : 0; }

extern "C" int zenv_get_config(const zenv_t *h, zenv_config *out)
{
    if (!h || !out) return fail(ZENV_E_ARG, "null argument");
    *out = h->cfg;
    return ZENV_OK;
}

// ============================================================================ layout bank
static int upload_bank(zenv *h, const std::vector<double> &robot4, const std::vector<double> &zone,
                       const std::vector<int32_t> &aux, const std::vector<int64_t> &seeds)
{
    int rc = use_device(h);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    // allocate and fill the new bank first; the handle switches to it only when all of it is on the device (a failure
    // half way leaves the old bank -- and bank_ready -- exactly as they were)
    const size_t S = seeds.size();
    const int32_t S_old = h->p.bank_size;
    const size_t bytes[5] = { robot4.size() * sizeof(double), zone.size() * sizeof(double), aux.size() * sizeof(int32_t),
                              S * sizeof(int64_t), S * 3 * sizeof(float4) };
    const void *src[5] = { robot4.data(), zone.data(), aux.data(), seeds.data(), nullptr };
    void *fresh[5] = { nullptr, nullptr, nullptr, nullptr, nullptr };
    hipError_t err = hipSuccess;
    for (int i = 0; i < 5 && err == hipSuccess; ++i) {
        err = hipMalloc(&fresh[i], bytes[i]);
        if (err == hipSuccess && src[i]) err = hipMemcpy(fresh[i], src[i], bytes[i], hipMemcpyHostToDevice);
    }
    if (err != hipSuccess) {
        for (void *m : fresh)
            if (m) (void)hipFree(m);
        return fail(ZENV_E_HIP, "uploading the layout bank: %s", hipGetErrorString(err));
    }
    for (int i = 0; i < 5; ++i) {
        if (h->bank_mem[i]) (void)hipFree(h->bank_mem[i]);
        h->bank_mem[i] = fresh[i];
    }
    h->p.bank_first = static_cast<const float4 *>(h->bank_mem[4]);
    h->p.bank_robot = static_cast<const double *>(h->bank_mem[0]);
    h->p.bank_zone = static_cast<const double *>(h->bank_mem[1]);
    h->p.bank_aux = static_cast<const int32_t *>(h->bank_mem[2]);
    h->p.bank_seed = static_cast<const int64_t *>(h->bank_mem[3]);
    h->p.bank_size = static_cast<int32_t>(S);
    HIP_TRY(launch_bank_derive(h->p, nullptr, (int)S, h->stream));     // first obs + first greedy action per slot
    h->bank_ready = true;
    if (h->p.sched_mode == SCHED_FIXED_SEEDS && h->p.seed_max - h->p.seed_min + 1 != (int64_t)S)
        h->sched_ready = false;  // that schedule draws slots of the bank it was made for: the next reset starts a default one
    if (h->sched_ready && h->p.sched_mode == SCHED_SEQUENTIAL) {
        // a sequential schedule carries over to a bank of another size: (first + k stride) mod the NEW size
        h->p.sched_stride %= (int32_t)S;
        HIP_TRY(launch_sched_sync(h->p, 0, h->stream));
    } else if (h->sched_ready && h->p.sched_mode == SCHED_RING && (int32_t)S != S_old) {
        h->sched_ready = false;  // a ring's slots belong to the bank it was laid out over
    }
    return ZENV_OK;
}

// Engine.reset's random half for a list of env seeds, on n_threads host threads: robot (x, y, quat w, quat z), zone
// centres, the task's aux column (tmax / colours / route ranks) per seed.
static int sample_bank_rows(const zenv *h, const int64_t *seed_list, size_t S, int n_threads, std::vector<double> &robot4,
                            std::vector<double> &zone, std::vector<int32_t> &aux)
{
    const int Z = h->cfg.num_zones;
    for (size_t i = 0; i < S; ++i)
        if (seed_list[i] < 0 || seed_list[i] + 1 > 0xFFFFFFFFll)
            return fail(ZENV_E_ARG, "Seed must be between 0 and 2**32 - 1");
    robot4.assign(S * 4, 0.0);
    zone.assign(S * Z * 2, 0.0);
    aux.assign(S * Z, 0);
    std::vector<int> status(S, 0);
    n_threads = std::max(1, std::min(n_threads, 256));
    if ((size_t)n_threads > S) n_threads = (int)S;
    auto work = [&](int tid) {
        for (size_t i = tid; i < S; i += n_threads) {
            Layout L;
            status[i] = sample_layout(h->cfg, seed_list[i], L);
            double s, c;
            det_sincos(L.robot_rot / 2, s, c);   // world.py rot2quat: [cos(rot/2), 0, 0, sin(rot/2)]
            robot4[4 * i + 0] = L.robot_x;
            robot4[4 * i + 1] = L.robot_y;
            robot4[4 * i + 2] = c;
            robot4[4 * i + 3] = s;
            for (int z = 0; z < Z; ++z) {
                zone[(i * Z + z) * 2 + 0] = L.zone_xy[z][0];
                zone[(i * Z + z) * 2 + 1] = L.zone_xy[z][1];
                aux[i * Z + z] = L.aux[z];
            }
            // solver-ordered variant: the aux column of a PointTSP bank carries the route (rank per zone)
            if (h->order_enabled && status[i] == 0) route_ranks(L.robot_x, L.robot_y, L.zone_xy, Z, &aux[i * Z]);
        }
    };
    if (n_threads == 1) {
        work(0);
    } else {
        std::vector<std::thread> pool;
        for (int t = 0; t < n_threads; ++t) pool.emplace_back(work, t);
        for (std::thread &t : pool) t.join();
    }
    for (size_t i = 0; i < S; ++i)
        if (status[i])
            return fail(ZENV_E_LAYOUT, "Failed to sample layout of objects (seed %lld)", (long long)seed_list[i]);
    return ZENV_OK;
}

extern "C" int zenv_bank_build_seeds(zenv_t *h, const int64_t *seed_list, int count, int n_threads)
{
    if (!h || !seed_list) return fail(ZENV_E_ARG, "null argument");
    if (count < 1) return fail(ZENV_E_ARG, "count must be >= 1");
    std::vector<double> robot4, zone;
    std::vector<int32_t> aux;
    std::vector<int64_t> seeds(seed_list, seed_list + count);
    int rc = sample_bank_rows(h, seed_list, (size_t)count, n_threads, robot4, zone, aux);
    if (rc) return rc;
    return upload_bank(h, robot4, zone, aux, seeds);
}

// Refill bank slots in place, stream-ordered: the layouts of `seeds` are sampled on the host, staged in page-locked
// memory, copied once and scattered by a small kernel -- behind every step already enqueued, before every later one.
extern "C" int zenv_bank_update(zenv_t *h, const int32_t *slots, const int64_t *seed_list, int count, int n_threads)
{
    if (!h || !slots || !seed_list) return fail(ZENV_E_ARG, "null argument");
    if (!h->bank_ready) return fail(ZENV_E_STATE, "build or set the layout bank first");
    if (count < 0) return fail(ZENV_E_ARG, "count must be >= 0");
    if (count == 0) return ZENV_OK;
    for (int i = 0; i < count; ++i)
        if (slots[i] < 0 || slots[i] >= h->p.bank_size)
            return fail(ZENV_E_ARG, "slot %d outside the bank [0,%d)", slots[i], h->p.bank_size);
    int rc = use_device(h);
    if (rc) return rc;
    std::vector<double> robot4, zone;
    std::vector<int32_t> aux;
    rc = sample_bank_rows(h, seed_list, (size_t)count, n_threads, robot4, zone, aux);
    if (rc) return rc;
    const size_t Z = (size_t)h->cfg.num_zones;
    const size_t rec = (4 + 2 * Z + 1) * 8 + ((Z * 4 + 7) / 8) * 8;
    const size_t need = (size_t)count * (rec + sizeof(int32_t));
    if (h->refill_cap < need) {
        HIP_TRY(hipStreamSynchronize(h->stream));
        if (h->refill_host) (void)hipHostFree(h->refill_host);
        if (h->refill_dev) (void)hipFree(h->refill_dev);
        h->refill_host = h->refill_dev = nullptr;
        h->refill_cap = 0;
        const size_t cap = std::max(need * 2, (size_t)1 << 16);
        HIP_TRY(hipHostMalloc(&h->refill_host, cap, hipHostMallocDefault));
        HIP_TRY(hipMalloc(&h->refill_dev, cap));
        h->refill_cap = cap;
    } else if (h->refill_busy) {
        HIP_TRY(hipEventSynchronize(h->refill_done));      // the previous refill still reads the staging buffer
    }
    char *hb = static_cast<char *>(h->refill_host);
    for (int i = 0; i < count; ++i) {
        double *rd = reinterpret_cast<double *>(hb + (size_t)i * rec);
        std::memcpy(rd, &robot4[4 * (size_t)i], 32);
        std::memcpy(rd + 4, &zone[2 * Z * (size_t)i], 16 * Z);
        std::memcpy(rd + 4 + 2 * Z, &seed_list[i], 8);
        std::memcpy(rd + 4 + 2 * Z + 1, &aux[Z * (size_t)i], 4 * Z);
    }
    std::memcpy(hb + (size_t)count * rec, slots, (size_t)count * sizeof(int32_t));
    HIP_TRY(hipMemcpyAsync(h->refill_dev, h->refill_host, need, hipMemcpyHostToDevice, h->stream));
    const char *db = static_cast<const char *>(h->refill_dev);
    HIP_TRY(launch_bank_scatter(h->p, reinterpret_cast<const int32_t *>(db + (size_t)count * rec), db, (int)rec, count,
                                h->stream));
    HIP_TRY(launch_bank_derive(h->p, reinterpret_cast<const int32_t *>(db + (size_t)count * rec), count, h->stream));
    if (!h->refill_done) HIP_TRY(hipEventCreateWithFlags(&h->refill_done, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(h->refill_done, h->stream));
    h->refill_busy = true;
    return ZENV_OK;
}

extern "C" int zenv_bank_build(zenv_t *h, int64_t seed_first, int count, int n_threads)
{
    if (!h) return fail(ZENV_E_ARG, "null handle");
    if (count < 1) return fail(ZENV_E_ARG, "count must be >= 1");
    std::vector<int64_t> seeds(count);
    for (int i = 0; i < count; ++i) seeds[i] = seed_first + i;
    return zenv_bank_build_seeds(h, seeds.data(), count, n_threads);
}

extern "C" int zenv_bank_set(zenv_t *h, const double *robot_xyrot, const double *zone_xy, const int32_t *aux_in,
                             const int64_t *seeds_in, int count)
{
    if (!h || !robot_xyrot || !zone_xy) return fail(ZENV_E_ARG, "null argument");
    if (count < 1) return fail(ZENV_E_ARG, "count must be >= 1");
    if (!aux_in && h->cfg.task != ZENV_TASK_TSP) return fail(ZENV_E_ARG, "aux required for this task");
    const int Z = h->cfg.num_zones;
    const size_t S = count;
    std::vector<double> robot4(S * 4), zone(zone_xy, zone_xy + S * Z * 2);
    std::vector<int32_t> aux(S * Z, 0);
    std::vector<int64_t> seeds(S, 0);
    for (size_t i = 0; i < S; ++i) {
        double s, c;
        det_sincos(robot_xyrot[3 * i + 2] / 2, s, c);
        robot4[4 * i + 0] = robot_xyrot[3 * i + 0];
        robot4[4 * i + 1] = robot_xyrot[3 * i + 1];
        robot4[4 * i + 2] = c;
        robot4[4 * i + 3] = s;
        if (seeds_in) seeds[i] = seeds_in[i];
    }
    if (aux_in) {
        for (size_t i = 0; i < S * Z; ++i) {
            if (h->cfg.task == ZENV_TASK_COLOUR_MATCH && (aux_in[i] < 0 || aux_in[i] > 2))
                return fail(ZENV_E_ARG, "colour %d outside 0..2", aux_in[i]);
            aux[i] = aux_in[i];
        }
    }
    if (h->order_enabled) {
        // solver-ordered variant: the aux column is the route (rank per zone); the caller's must be a
        // permutation of 0..Z-1, a missing one is filled with the built-in tour
        for (size_t i = 0; i < S; ++i) {
            if (!aux_in) {
                route_ranks(robot_xyrot[3 * i], robot_xyrot[3 * i + 1],
                            reinterpret_cast<const double(*)[2]>(zone_xy + i * Z * 2), Z, &aux[i * Z]);
                continue;
            }
            uint64_t seen = 0;
            for (int z = 0; z < Z; ++z)
                if (aux[i * Z + z] >= 0 && aux[i * Z + z] < Z) seen |= 1ull << aux[i * Z + z];
            if (seen != (Z >= 64 ? ~0ull : (1ull << Z) - 1)) return fail(ZENV_E_ARG, "route %zu is not a permutation of 0..%d", i, Z - 1);
        }
    }
    return upload_bank(h, robot4, zone, aux, seeds);
}

extern "C" int zenv_bank_size(const zenv_t *h) { return h ? h->p.bank_size : 0; }

// ============================================================================ schedule
extern "C" int zenv_schedule_sequential(zenv_t *h, const int32_t *first, int32_t stride)
{
    if (!h) return fail(ZENV_E_ARG, "null handle");
    if (!h->bank_ready) return fail(ZENV_E_STATE, "build or set the layout bank first");
    if (stride < 0) return fail(ZENV_E_ARG, "stride must be >= 0");
    int rc = use_device(h);
    if (rc) return rc;
    const int S = h->p.bank_size;
    std::vector<int32_t> f(h->n_env);
    for (int i = 0; i < h->n_env; ++i) {
        const int32_t v = first ? first[i] : (i % S);
        if (v < 0 || v >= S) return fail(ZENV_E_ARG, "first[%d] = %d outside the bank [0,%d)", i, v, S);
        f[i] = v;
    }
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipMemcpy(h->p.slot_first, f.data(), f.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    h->p.sched_mode = SCHED_SEQUENTIAL;
    h->p.sched_stride = stride % S;
    HIP_TRY(launch_sched_sync(h->p, 1, h->stream));   // episode 0, next_slot = first (on the handle's stream)
    h->sched_ready = true;
    return ZENV_OK;
}

extern "C" int zenv_schedule_ring(zenv_t *h, const int32_t *first, int32_t depth)
{
    if (!h || !first) return fail(ZENV_E_ARG, "null argument");
    if (!h->bank_ready) return fail(ZENV_E_STATE, "build or set the layout bank first");
    if (depth < 1) return fail(ZENV_E_ARG, "ring depth must be >= 1");
    int rc = use_device(h);
    if (rc) return rc;
    const int S = h->p.bank_size;
    for (int i = 0; i < h->n_env; ++i)
        if (first[i] < 0 || (int64_t)first[i] + depth > S)
            return fail(ZENV_E_ARG, "ring of env %d, slots [%d, %d), leaves the bank [0,%d)", i, first[i], first[i] + depth, S);
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipMemcpy(h->p.slot_first, first, (size_t)h->n_env * sizeof(int32_t), hipMemcpyHostToDevice));
    h->p.sched_mode = SCHED_RING;
    h->p.sched_stride = depth;
    HIP_TRY(launch_sched_sync(h->p, 1, h->stream));
    h->sched_ready = true;
    return ZENV_OK;
}

extern "C" int zenv_schedule_fixed_seeds(zenv_t *h, const uint64_t *rng_seeds, int64_t min_seed, int64_t max_seed)
{
    if (!h || !rng_seeds) return fail(ZENV_E_ARG, "null argument");
    if (!h->bank_ready) return fail(ZENV_E_STATE, "build or set the layout bank first");
    if (max_seed < min_seed || max_seed - min_seed + 1 != h->p.bank_size)
        return fail(ZENV_E_ARG, "bank must hold exactly the seeds min_seed..max_seed (%d layouts)", h->p.bank_size);
    int rc = use_device(h);
    if (rc) return rc;
    std::vector<uint64_t> st(4 * (size_t)h->n_env);
    for (int i = 0; i < h->n_env; ++i) {
        const Pcg64State s = pcg64_from_seed(rng_seeds[i]);
        st[4 * (size_t)i + 0] = s.state_hi;
        st[4 * (size_t)i + 1] = s.state_lo;
        st[4 * (size_t)i + 2] = s.inc_hi;
        st[4 * (size_t)i + 3] = s.inc_lo;
    }
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipMemcpy(h->p.pcg, st.data(), st.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMemsetAsync(h->p.pcg_buf, 0, 2 * (size_t)h->n_env * sizeof(uint32_t), h->stream));
    h->p.sched_mode = SCHED_FIXED_SEEDS;
    h->p.seed_min = min_seed;
    h->p.seed_max = max_seed;
    HIP_TRY(launch_sched_sync(h->p, 1, h->stream));
    h->sched_ready = true;
    return ZENV_OK;
}

// ============================================================================ hot path
extern "C" int zenv_reset(zenv_t *h, const uint8_t *mask)
{
    if (!h) return fail(ZENV_E_ARG, "null handle");
    if (!h->bank_ready) return fail(ZENV_E_STATE, "build or set the layout bank first");
    int rc = use_device(h);
    if (rc) return rc;
    if (!h->sched_ready) {
        rc = zenv_schedule_sequential(h, nullptr, 0);
        if (rc) return rc;
    }
    if (mask && !h->was_reset) return fail(ZENV_E_STATE, "the first reset must cover every env (mask == NULL)");
    const uint8_t *dmask = nullptr;
    if (mask) {
        HIP_TRY(hipMemcpyAsync(h->d_mask, mask, h->n_env, hipMemcpyHostToDevice, h->stream));
        dmask = h->d_mask;
    }
    h->act_tag.valid = false;
    HIP_TRY(launch_reset(h->p, dmask, h->stream));
    if (h->goal_enabled) HIP_TRY(launch_goal_clear(h->p, dmask, h->stream));
    if (h->order_enabled) HIP_TRY(launch_order_reset(h->p, dmask, h->stream));
    h->was_reset = true;
    return ZENV_OK;
}

extern "C" int zenv_step(zenv_t *h, const float *actions, int actions_on_device, int auto_reset)
{
    if (!h) return fail(ZENV_E_ARG, "null handle");
    if (!h->was_reset) return fail(ZENV_E_STATE, "Environment must be reset before stepping");
    int rc = use_device(h);
    if (rc) return rc;
    h->act_tag.valid = false;
    const float *d_act = h->p.actions;
    if (actions) {
        if (actions_on_device) {
            d_act = actions;
        } else if (h->host_io_actions) {
            // the kernel reads the page-locked buffer itself; nothing of an earlier step can still be reading it only if
            // the stream has drained -- which every zenv_host_io caller's step ends with (zenv_step_host / _results)
            HIP_TRY(hipStreamSynchronize(h->stream));
            if (actions != h->host_io_actions) std::memcpy(h->host_io_actions, actions, sizeof(float) * 2 * (size_t)h->n_env);
        } else {
            HIP_TRY(hipMemcpyAsync(h->p.actions, actions, sizeof(float) * 2 * (size_t)h->n_env,
                                   hipMemcpyHostToDevice, h->stream));
        }
    }
    HIP_TRY(launch_step(h->p, d_act, auto_reset, no_policy(), h->stream));
    if (h->goal_enabled) HIP_TRY(launch_goal_step(h->p, h->stream));
    if (h->order_enabled) HIP_TRY(launch_order_step(h->p, h->stream));
    h->step_count += 1;
    return ZENV_OK;
}

// ============================================================================ solver-ordered variant
extern "C" int zenv_route_ranks(const double *robot_xy, const double *zone_xy, int num_zones, int32_t *rank)
{
    if (!robot_xy || !zone_xy || !rank) return fail(ZENV_E_ARG, "null argument");
    if (num_zones < 1 || num_zones > ZENV_MAX_ZONES) return fail(ZENV_E_ARG, "num_zones outside [1,%d]", ZENV_MAX_ZONES);
    route_ranks(robot_xy[0], robot_xy[1], reinterpret_cast<const double(*)[2]>(zone_xy), num_zones, rank);
    return ZENV_OK;
}

extern "C" int zenv_order_enable(zenv_t *h)
{
    if (!h) return fail(ZENV_E_ARG, "null handle");
    if (h->cfg.task != ZENV_TASK_TSP) return fail(ZENV_E_ARG, "the solver-ordered variant is a PointTSP variant");
    if (h->goal_enabled) return fail(ZENV_E_STATE, "a handle is goal-conditioned or solver-ordered, not both");
    if (h->bank_ready) return fail(ZENV_E_STATE, "call zenv_order_enable before building the bank (routes ride in it)");
    if (h->order_enabled) return ZENV_OK;
    int rc = use_device(h);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    const size_t N = (size_t)h->n_env, Z = (size_t)h->p.Z;
    DevParams &p = h->p;
    HIP_TRY(hipMalloc((void **)&p.visit_zone, N * 4));
    HIP_TRY(hipMalloc((void **)&p.term_xy, N * 16));
    HIP_TRY(hipMalloc((void **)&p.goal, N * 4));
    HIP_TRY(hipMalloc((void **)&p.goal_last, N * 8));
    HIP_TRY(hipMalloc((void **)&p.goal_xy, N * 16));
    HIP_TRY(hipMalloc((void **)&p.shaped, N * 8));
    HIP_TRY(hipMalloc((void **)&p.order_pos, N * Z));
    HIP_TRY(hipMalloc((void **)&p.order_val, N * Z * 4));
    HIP_TRY(hipMemsetAsync(p.term_xy, 0, N * 16, h->stream));
    HIP_TRY(hipMemsetAsync(p.goal_xy, 0, N * 16, h->stream));
    HIP_TRY(hipMemsetAsync(p.goal_last, 0, N * 8, h->stream));
    HIP_TRY(hipMemsetAsync(p.shaped, 0, N * 8, h->stream));
    HIP_TRY(hipMemsetAsync(p.goal, 0xFF, N * 4, h->stream));
    HIP_TRY(hipMemsetAsync(p.visit_zone, 0xFF, N * 4, h->stream));
    HIP_TRY(hipMemsetAsync(p.order_pos, 0xFF, N * Z, h->stream));
    HIP_TRY(hipMemsetAsync(p.order_val, 0, N * Z * 4, h->stream));
    h->order_enabled = true;
    return ZENV_OK;
}

extern "C" int zenv_order_configure(zenv_t *h, int flags)
{
    if (!h) return fail(ZENV_E_ARG, "null handle");
    if (!h->order_enabled) return fail(ZENV_E_STATE, "zenv_order_enable first");
    if (flags & ~ZENV_ORDER_FRESH_FIRST_OBS) return fail(ZENV_E_ARG, "unknown zenv_order_configure flags 0x%x", flags);
    h->p.order_fresh = (flags & ZENV_ORDER_FRESH_FIRST_OBS) ? 1 : 0;   // by-value kernel argument of the next launch
    return ZENV_OK;
}

// ============================================================================ goal-conditioned variant
extern "C" int zenv_goal_enable(zenv_t *h)
{
    if (!h) return fail(ZENV_E_ARG, "null handle");
    if (h->goal_enabled) return ZENV_OK;
    if (h->order_enabled) return fail(ZENV_E_STATE, "a handle is goal-conditioned or solver-ordered, not both");
    int rc = use_device(h);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    const size_t N = (size_t)h->n_env;
    DevParams &p = h->p;
    HIP_TRY(hipMalloc((void **)&p.visit_zone, N * 4));
    HIP_TRY(hipMalloc((void **)&p.term_xy, N * 16));
    HIP_TRY(hipMalloc((void **)&p.goal, N * 4));
    HIP_TRY(hipMalloc((void **)&p.goal_last, N * 8));
    HIP_TRY(hipMalloc((void **)&p.goal_xy, N * 16));
    HIP_TRY(hipMalloc((void **)&p.shaped, N * 8));
    HIP_TRY(hipMalloc((void **)&p.need_goal, N));
    HIP_TRY(hipMalloc((void **)&p.available, N * 4));
    HIP_TRY(hipMalloc((void **)&h->goal_in, N * 4));
    HIP_TRY(hipMalloc((void **)&h->goal_bad, 4));
    HIP_TRY(hipMemsetAsync(p.term_xy, 0, N * 16, h->stream));
    HIP_TRY(hipMemsetAsync(p.goal_last, 0, N * 8, h->stream));
    HIP_TRY(hipMemsetAsync(p.goal_xy, 0, N * 16, h->stream));
    HIP_TRY(launch_goal_clear(p, nullptr, h->stream));
    h->goal_enabled = true;
    return ZENV_OK;
}

extern "C" int zenv_set_goals(zenv_t *h, const int32_t *goals)
{
    if (!h || !goals) return fail(ZENV_E_ARG, "null argument");
    if (!h->goal_enabled) return fail(ZENV_E_STATE, "zenv_goal_enable first");
    if (!h->was_reset) return fail(ZENV_E_STATE, "reset before setting goals");
    int rc = use_device(h);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(h->goal_in, goals, sizeof(int32_t) * (size_t)h->n_env, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemsetAsync(h->goal_bad, 0, 4, h->stream));
    HIP_TRY(launch_goal_set(h->p, h->goal_in, h->goal_bad, h->stream));
    int32_t bad = 0;
    HIP_TRY(hipMemcpyAsync(&bad, h->goal_bad, 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (bad) return fail(ZENV_E_ARG, "%d goal zone(s) out of range or already visited", bad);
    return ZENV_OK;
}

extern "C" int zenv_solver_goals(zenv_t *h, int32_t *goals)
{
    if (!h || !goals) return fail(ZENV_E_ARG, "null argument");
    if (h->cfg.task != ZENV_TASK_COLOUR_MATCH) return fail(ZENV_E_ARG, "solver_get_next_goal is a ColourMatch function");
    if (!h->goal_enabled) return fail(ZENV_E_STATE, "zenv_goal_enable first");
    if (!h->was_reset) return fail(ZENV_E_STATE, "reset before asking for goals");
    int rc = use_device(h);
    if (rc) return rc;
    HIP_TRY(launch_solver_goal(h->p, h->goal_in, h->stream));
    HIP_TRY(hipMemcpyAsync(goals, h->goal_in, sizeof(int32_t) * (size_t)h->n_env, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return ZENV_OK;
}

// ZENV_MLP_F16X3: the network kernel sets *mlp_range_flag (pinned host memory) when one of its operands left float16's
// range; every entry point that waits for the device looks at it once the stream has drained
static int mlp_range_check(zenv *h)
{
    if (!h->mlp_range_flag || !*(volatile int *)h->mlp_range_flag) return ZENV_OK;
    *(volatile int *)h->mlp_range_flag = 0;
    return fail(ZENV_E_RANGE, "an input or activation of the network reached 65 520, beyond float16 (ZENV_MLP_F16: or an "
                              "observation beyond 64): the actions since the last synchronising call are invalid -- load these "
                              "weights with ZENV_MLP_BF16, ZENV_MLP_BF16X3 or ZENV_MLP_F32");
}

// ============================================================================ actor network
extern "C" int zenv_mlp_load(zenv_t *h, const zenv_mlp_weights *w)
{
    if (!h || !w) return fail(ZENV_E_ARG, "null argument");
    for (const float *t : { w->zone_w1, w->zone_b1, w->zone_w2, w->zone_b2, w->zone_w3, w->zone_b3, w->comb_w, w->comb_b,
                            w->enc_w, w->enc_b, w->mu_w, w->mu_b, w->std_w, w->std_b })
        if (!t) return fail(ZENV_E_ARG, "zenv_mlp_weights has a null tensor");
    const int n_critic = (w->critic_w1 != nullptr) + (w->critic_b1 != nullptr) + (w->critic_w2 != nullptr) +
                         (w->critic_b2 != nullptr);
    if (n_critic != 0 && n_critic != 4) return fail(ZENV_E_ARG, "give all four critic tensors or none");
    const int n_sigma = (w->critic_sigma_w != nullptr) + (w->critic_sigma_b != nullptr);
    if (n_sigma == 1 || (n_sigma == 2 && n_critic == 0))
        return fail(ZENV_E_ARG, "the distributional critic needs critic.0, critic_mu (as critic_w2 / _b2) and critic_sigma");
    if (w->precision < ZENV_MLP_BF16 || w->precision > ZENV_MLP_F16)
        return fail(ZENV_E_ARG, "unknown zenv_mlp_weights.precision %d", w->precision);
    if (w->precision == ZENV_MLP_F16 && w->h_dim >= 1 && w->h_dim < kMlpHP) {
        // single float16 operands.  (i) every weight must be a finite float16; (ii) the zone layers' activations are bounded
        // here, over the rows' absolute sums, for observations up to kMlpF16ObsBound and zone rows up to 2 (positions / 3,
        // flags, timers: bounded by construction) -- the zone kernel has no cycles to spare for watching them; the head
        // kernel watches its own.
        const int hd = w->h_dim, F = h->p.F;
        const struct { const float *t; size_t n; const char *name; } all[] = {
            { w->zone_w1, (size_t)hd * (8 + F), "zone_net_.0.weight" }, { w->zone_b1, (size_t)hd, "zone_net_.0.bias" },
            { w->zone_w2, (size_t)hd * hd, "zone_net_.2.weight" },      { w->zone_b2, (size_t)hd, "zone_net_.2.bias" },
            { w->enc_w, (size_t)hd * hd, "actor.enc_.0.0.weight" },     { w->enc_b, (size_t)hd, "actor.enc_.0.0.bias" },
            { w->mu_w, (size_t)2 * hd, "actor.mu_.weight" },            { w->mu_b, 2, "actor.mu_.bias" },
            { w->std_w, (size_t)2 * hd, "actor.std_.weight" },          { w->std_b, 2, "actor.std_.bias" },
            { w->critic_w1, (size_t)hd * hd, "critic.0.weight" },       { w->critic_b1, (size_t)hd, "critic.0.bias" },
            { w->critic_w2, (size_t)hd, "critic.2.weight" },            { w->critic_b2, 1, "critic.2.bias" },
            { w->critic_sigma_w, (size_t)hd, "critic_sigma.weight" },   { w->critic_sigma_b, 1, "critic_sigma.bias" } };
        for (const auto &a : all)
            for (size_t i = 0; a.t && i < a.n; ++i)
                if (!(std::fabs(a.t[i]) < 65504.0f))
                    return fail(ZENV_E_RANGE, "%s[%zu] = %g is not a finite float16: use ZENV_MLP_BF16 or a split mode",
                                a.name, i, (double)a.t[i]);
        double a1 = 0.0, s2 = 0.0, b2 = 0.0;
        for (int r = 0; r < hd; ++r) {
            double v = std::fabs((double)w->zone_b1[r]), s = 0.0;
            for (int k = 0; k < 8 + F; ++k)
                v += std::fabs((double)w->zone_w1[(size_t)r * (8 + F) + k]) * (k < 8 ? (double)kMlpF16ObsBound : 2.0);
            a1 = std::max(a1, v);
            for (int k = 0; k < hd; ++k) s += std::fabs((double)w->zone_w2[(size_t)r * hd + k]);
            s2 = std::max(s2, s);
            b2 = std::max(b2, std::fabs((double)w->zone_b2[r]));
        }
        if (!(a1 < 65504.0) || !(s2 * a1 + b2 < 65504.0))
            return fail(ZENV_E_RANGE, "zone_net_'s activations are bounded by %.3g / %.3g for observations up to %g: beyond "
                                      "float16 -- use ZENV_MLP_BF16 or a split mode for these weights",
                        a1, s2 * a1 + b2, (double)kMlpF16ObsBound);
    }
    if (w->precision == ZENV_MLP_F16X3 && w->h_dim >= 1 && w->h_dim < kMlpHP) {
        // float16 halves: a weight of 65 520 or more would be inf on the device
        const int hd = w->h_dim, F = h->p.F;
        const struct { const float *t; size_t n; const char *name; } all[] = {
            { w->zone_w1, (size_t)hd * (8 + F), "zone_net_.0.weight" }, { w->zone_b1, (size_t)hd, "zone_net_.0.bias" },
            { w->zone_w2, (size_t)hd * hd, "zone_net_.2.weight" },      { w->zone_b2, (size_t)hd, "zone_net_.2.bias" },
            { w->zone_w3, (size_t)hd * hd, "zone_net_.4.weight" },      { w->zone_b3, (size_t)hd, "zone_net_.4.bias" },
            { w->comb_w, (size_t)hd * (8 + hd), "combine_net_.weight" }, { w->comb_b, (size_t)hd, "combine_net_.bias" },
            { w->enc_w, (size_t)hd * hd, "actor.enc_.0.0.weight" },     { w->enc_b, (size_t)hd, "actor.enc_.0.0.bias" },
            { w->mu_w, (size_t)2 * hd, "actor.mu_.weight" },            { w->mu_b, 2, "actor.mu_.bias" },
            { w->std_w, (size_t)2 * hd, "actor.std_.weight" },          { w->std_b, 2, "actor.std_.bias" },
            { w->critic_w1, (size_t)hd * hd, "critic.0.weight" },       { w->critic_b1, (size_t)hd, "critic.0.bias" },
            { w->critic_w2, (size_t)hd, "critic.2.weight" },            { w->critic_b2, 1, "critic.2.bias" },
            { w->critic_sigma_w, (size_t)hd, "critic_sigma.weight" },   { w->critic_sigma_b, 1, "critic_sigma.bias" } };
        for (const auto &a : all)
            for (size_t i = 0; a.t && i < a.n; ++i)
                if (!(std::fabs(a.t[i]) < 32768.0f))
                    return fail(ZENV_E_RANGE, "%s[%zu] = %g: ZENV_MLP_F16X3 keeps weights as float16 pairs (|w| < 32 768); use "
                                              "ZENV_MLP_BF16X3 or ZENV_MLP_F32", a.name, i, (double)a.t[i]);
    }
    std::vector<uint16_t> img;
    size_t offs[8];
    if (pack_images(*w, h->p.F, img, offs, w->precision == ZENV_MLP_F16) != 0)
        return fail(ZENV_E_ARG, "h_dim %d outside [1, %d]", w->h_dim, kMlpHP - 1);
    if (w->precision == ZENV_MLP_F16)        // zone_net_.4 folded into combine_net_: a product that can leave the range
        for (size_t i = 0; i < img.size(); ++i)
            if ((img[i] & 0x7C00u) == 0x7C00u)
                return fail(ZENV_E_RANGE, "combine_net_ folded over zone_net_.4 leaves float16's range: use ZENV_MLP_BF16 or "
                                          "a split mode for these weights");
    std::vector<float> f32;         // the float32-grade modes' images (packed before the handle is touched: it can refuse)
    size_t fo[30];
    const bool f32_grade = w->precision != ZENV_MLP_BF16 && w->precision != ZENV_MLP_F16;
    if (f32_grade) {
        pack_f32(*w, h->p.F, f32, fo);
        if (w->precision == ZENV_MLP_F16X3) {
            // the float16 images start at fo[24] (zone_net_.2): combine_net_ arrives with zone_net_.4 folded in, a product of
            // two in-range matrices that need not be in range itself
            const uint16_t *hw = reinterpret_cast<const uint16_t *>(f32.data() + fo[24]);
            const size_t n16 = (f32.size() - fo[24]) * 2;
            for (size_t i = 0; i < n16; ++i)
                if ((hw[i] & 0x7C00u) == 0x7C00u)
                    return fail(ZENV_E_RANGE, "combine_net_ folded over zone_net_.4 leaves float16's range: use ZENV_MLP_BF16X3 "
                                              "or ZENV_MLP_F32 for these weights");
        }
    }
    int rc = use_device(h);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    const size_t N = (size_t)h->n_env;
    h->mlp_ready = false;
    if (h->mlp_mem) HIP_TRY(hipFree(h->mlp_mem));   // its size depends on whether there is a critic
    h->mlp_mem = nullptr;
    if (h->mlp_f32_mem) HIP_TRY(hipFree(h->mlp_f32_mem));
    h->mlp_f32_mem = nullptr;
    HIP_TRY(hipMalloc(&h->mlp_mem, img.size() * 2));
    if (!h->mlp_value) HIP_TRY(hipMalloc((void **)&h->mlp_value, N * sizeof(float)));
    HIP_TRY(hipMemsetAsync(h->mlp_value, 0, N * sizeof(float), h->stream));
    if (!h->mlp_value_sigma) HIP_TRY(hipMalloc((void **)&h->mlp_value_sigma, N * sizeof(float)));
    HIP_TRY(hipMemsetAsync(h->mlp_value_sigma, 0, N * sizeof(float), h->stream));
    if (!h->mlp_pooled) HIP_TRY(hipMalloc(&h->mlp_pooled, N * kMlpHP * sizeof(uint16_t)));
    if (!h->mlp_mu) HIP_TRY(hipMalloc((void **)&h->mlp_mu, N * 2 * sizeof(float)));
    if (!h->mlp_std) HIP_TRY(hipMalloc((void **)&h->mlp_std, N * 2 * sizeof(float)));
    HIP_TRY(hipMemcpy(h->mlp_mem, img.data(), img.size() * 2, hipMemcpyHostToDevice));
    const char *base = static_cast<const char *>(h->mlp_mem);
    h->mlp = MlpImages{ base + offs[0], base + offs[1], base + offs[2], base + offs[3], base + offs[4], base + offs[5],
                        n_critic ? base + offs[6] : nullptr, n_critic ? base + offs[7] : nullptr, nullptr,
                        n_sigma == 2 ? 1 : 0, 0, nullptr };
    if (w->precision != ZENV_MLP_BF16 && !h->mlp_range_flag) {
        HIP_TRY(hipHostMalloc((void **)&h->mlp_range_flag, sizeof(int), hipHostMallocDefault));
        *h->mlp_range_flag = 0;
    }
    if (w->precision == ZENV_MLP_F16) {
        h->mlp.elem_f16 = 1;
        h->mlp.range_flag = h->mlp_range_flag;
    }
    if (f32_grade) {
        // (diagnostic: ZENV_MLP_F32_VALU=1 runs the network on the vector ALU, k_mlp_f32, instead of the f32 MFMA)
        // ZENV_MLP_F32_MFMA=1 the MFMA kernel whatever the batch; default: by batch size, see launch_mlp_forward_f32)
        const int on_mfma = std::getenv("ZENV_MLP_F32_VALU") ? 0 : std::getenv("ZENV_MLP_F32_MFMA") ? 2 : 1;
        HIP_TRY(hipMalloc(&h->mlp_f32_mem, f32.size() * sizeof(float)));
        HIP_TRY(hipMemcpy(h->mlp_f32_mem, f32.data(), f32.size() * sizeof(float), hipMemcpyHostToDevice));
        const float *fb = static_cast<const float *>(h->mlp_f32_mem);
        h->mlp_f32 = MlpF32{ w->h_dim, n_sigma == 2 ? 1 : 0, n_critic ? 1 : 0, 0,
                             fb + fo[0], fb + fo[1], fb + fo[2], fb + fo[3], fb + fo[4], fb + fo[5], fb + fo[6],
                             fb + fo[7], fb + fo[8], fb + fo[9], n_critic ? fb + fo[10] : nullptr,
                             n_critic ? fb + fo[11] : nullptr, fb + fo[12], fb + fo[13], fb + fo[14], fb + fo[15], fb + fo[16], fb + fo[17], fb + fo[18],
                             n_critic ? fb + fo[19] : nullptr, n_critic ? fb + fo[20] : nullptr, on_mfma,
                             w->precision == ZENV_MLP_BF16X3 ? 1 : w->precision == ZENV_MLP_F16X3 ? 2 : 0,
                             fb + fo[21], fb + fo[22], fb + fo[23], fb + fo[24], fb + fo[25],
                             n_critic ? fb + fo[26] : nullptr, fb + fo[27], fb + fo[28], n_critic ? fb + fo[29] : nullptr,
                             h->mlp_range_flag };
        h->mlp.f32 = &h->mlp_f32;
    }
    h->mlp_ready = true;
    return ZENV_OK;
}

extern "C" int zenv_mlp_forward(zenv_t *h)
{
    if (!h) return fail(ZENV_E_ARG, "null handle");
    if (!h->mlp_ready) return fail(ZENV_E_STATE, "zenv_mlp_load first");
    if (!h->was_reset) return fail(ZENV_E_STATE, "reset before asking for actions");
    int rc = use_device(h);
    if (rc) return rc;
    HIP_TRY(launch_mlp_forward(h->mlp, h->n_env, h->p.Z, h->p.F, h->p.obs, h->p.zone_obs, h->mlp_pooled, h->mlp_mu,
                               h->mlp_std, h->mlp_value, h->mlp_value_sigma, no_mlp_action(), h->stream));
    return ZENV_OK;
}

static bool policy_known(int policy) { return policy >= ZENV_POLICY_UNIFORM && policy <= ZENV_POLICY_MLP_SAMPLE; }
static bool policy_is_mlp(int policy) { return policy == ZENV_POLICY_MLP_MEAN || policy == ZENV_POLICY_MLP_SAMPLE; }

// a_t = pi(obs_t, t) into pol.out, for every kind of action source
static int run_policy(zenv_t *h, const StepPolicy &pol, const MlpRecord *rec = nullptr)
{
    if (!policy_is_mlp(pol.policy)) {
        HIP_TRY(launch_policy(h->p, pol, h->stream));
        return ZENV_OK;
    }
    if (!h->mlp_ready) return fail(ZENV_E_STATE, "zenv_mlp_load first");
    // the head kernel also turns (mu, std) into the action
    const MlpAction act{ pol.policy == ZENV_POLICY_MLP_SAMPLE ? 1 : 0, pol.step_index, pol.seed, pol.env_index0, pol.out,
                         rec ? *rec : MlpRecord{} };
    HIP_TRY(launch_mlp_forward(h->mlp, h->n_env, h->p.Z, h->p.F, h->p.obs, h->p.zone_obs, h->mlp_pooled, h->mlp_mu,
                               h->mlp_std, h->mlp_value, h->mlp_value_sigma, act, h->stream));
    return ZENV_OK;
}

// ============================================================================ experience collection
extern "C" int zenv_collect(zenv_t *h, int T, uint64_t policy_seed, uint64_t env_index0, float discount, float gae_lambda)
{
    if (!h) return fail(ZENV_E_ARG, "null handle");
    if (T < 1) return fail(ZENV_E_ARG, "frames_per_proc must be positive");
    if (!h->was_reset) return fail(ZENV_E_STATE, "Environment must be reset before stepping");
    if (!h->mlp_ready || !h->mlp.wv1) return fail(ZENV_E_STATE, "zenv_mlp_load with actor and critic weights first");
    if (h->host_io_slab) return fail(ZENV_E_STATE, "zenv_collect records on the device: switch zenv_host_io off first");
    if (h->order_enabled)
        return fail(ZENV_E_STATE, "solver-ordered envs are stepped with zenv_step (their order feature is not part of "
                                  "the network input this call evaluates)");
    int rc = use_device(h);
    if (rc) return rc;
    const size_t N = (size_t)h->n_env, ZF = (size_t)h->p.Z * h->p.F;
    h->act_tag.valid = false;
    if (!h->exp_mem || h->exp.T != T) {
        HIP_TRY(hipStreamSynchronize(h->stream));
        std::vector<float> keep_mask;
        if (h->exp_mem) {       // self.mask survives a change of T
            keep_mask.resize(N);
            HIP_TRY(hipMemcpy(keep_mask.data(), h->exp.cur_mask, N * 4, hipMemcpyDeviceToHost));
            HIP_TRY(hipFree(h->exp_mem));
            h->exp_mem = nullptr;
        }
        const size_t per_slot = 8 + ZF + 2 + 2 + 5;            // floats per (env, t)
        const size_t total = N * (size_t)T * per_slot + N;
        HIP_TRY(hipMalloc(&h->exp_mem, total * sizeof(float)));
        float *f = static_cast<float *>(h->exp_mem);
        ExpBuffers &x = h->exp;
        x.T = T;
        x.obs = f;            f += N * T * 8;
        x.zone_obs = f;       f += N * T * ZF;
        x.action = f;         f += N * T * 2;
        x.log_prob = f;       f += N * T * 2;
        x.value = f;          f += N * T;
        x.reward = f;         f += N * T;
        x.mask = f;           f += N * T;
        x.advantage = f;      f += N * T;
        x.returnn = f;        f += N * T;
        x.cur_mask = f;
        if (keep_mask.empty()) keep_mask.assign(N, 1.0f);      // base.py:96 self.mask = ones
        HIP_TRY(hipMemcpy(x.cur_mask, keep_mask.data(), N * 4, hipMemcpyHostToDevice));
    }
    // The observations are recorded where they are produced: the policy of frame t reads slot t, the step kernel
    // of frame t writes obs_{t+1} into slot t+1 (the last one into the handle's own buffers again).
    struct ObsRedirect {
        zenv_t *h;
        float *obs, *zone_obs;
        ~ObsRedirect() { h->p.obs = obs; h->p.zone_obs = zone_obs; }
    } home{ h, h->p.obs, h->p.zone_obs };
    HIP_TRY(hipMemcpyAsync(h->exp.obs, home.obs, N * 8 * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->exp.zone_obs, home.zone_obs, N * ZF * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
    for (int t = 0; t < T; ++t) {
        h->p.obs = h->exp.obs + (size_t)t * N * 8;
        h->p.zone_obs = h->exp.zone_obs + (size_t)t * N * ZF;
        StepPolicy pol{ ZENV_POLICY_MLP_SAMPLE, (uint32_t)h->step_count, policy_seed, env_index0, h->p.actions };
        // dist, value = acmodel(obs); action = dist.sample(); the head kernel also records frame t (and the reward
        // of frame t-1, still in the env's reward / done buffers)
        const MlpRecord rec{ h->exp.action, h->exp.log_prob, h->exp.value, h->exp.mask, h->exp.reward, h->exp.cur_mask,
                             h->p.reward, h->goal_enabled ? h->p.shaped : nullptr, h->p.done_out, T, t, h->n_env };
        rc = run_policy(h, pol, &rec);
        if (rc) return rc;
        h->p.obs = t + 1 < T ? h->exp.obs + (size_t)(t + 1) * N * 8 : home.obs;
        h->p.zone_obs = t + 1 < T ? h->exp.zone_obs + (size_t)(t + 1) * N * ZF : home.zone_obs;
        HIP_TRY(launch_step(h->p, h->p.actions, 1, no_policy(), h->stream));     // ParallelEnv.step: auto-reset
        if (h->goal_enabled) HIP_TRY(launch_goal_step(h->p, h->stream));
        h->step_count += 1;
    }
    HIP_TRY(launch_exp_reward(h->exp, h->n_env, T - 1, h->p.reward, h->goal_enabled ? h->p.shaped : nullptr,
                              h->p.done_out, h->stream));
    // next_value = value(obs_T) (:177-187), then the GAE recursion
    HIP_TRY(launch_mlp_forward(h->mlp, h->n_env, h->p.Z, h->p.F, h->p.obs, h->p.zone_obs, h->mlp_pooled, h->mlp_mu,
                               h->mlp_std, h->mlp_value, h->mlp_value_sigma, no_mlp_action(), h->stream));
    HIP_TRY(launch_exp_gae(h->exp, h->n_env, h->mlp_value, discount, gae_lambda, h->stream));
    return ZENV_OK;
}


extern "C" int zenv_policy(zenv_t *h, int policy, uint64_t policy_seed, uint64_t env_index0, float *dst_device)
{
    if (!h) return fail(ZENV_E_ARG, "null handle");
    if (!h->was_reset) return fail(ZENV_E_STATE, "reset before asking for actions");
    if (!policy_known(policy)) return fail(ZENV_E_ARG, "unknown policy %d", policy);
    int rc = use_device(h);
    if (rc) return rc;
    const StepPolicy pol{ policy, (uint32_t)h->step_count, policy_seed, env_index0,
                          dst_device ? dst_device : h->p.actions };
    h->act_tag.valid = false;
    return run_policy(h, pol);
}

// steps per launch of the persistent kernel (diagnostic: ZENV_ROLLOUT_CHUNK_EXP=<n> shortens it, e.g. to time the
// persistent kernel at one step per launch against the per-step kernel)
static const int kRolloutChunk = [] {
    const char *e = std::getenv("ZENV_ROLLOUT_CHUNK_EXP");
    const int v = e ? std::atoi(e) : 0;
    return v >= 1 && v <= ZENV_ROLLOUT_CHUNK ? v : ZENV_ROLLOUT_CHUNK;
}();
extern "C" int zenv_rollout_chunk(void) { return kRolloutChunk; }

// Bring the device copy of the parameter block up to date (it changes with the bank, the schedule, a redirected
// output buffer ...: rarely, so one comparison per rollout call and a copy only when something did change).
static int ensure_self(zenv_t *h)
{
    if (h->self_valid && std::memcmp(&h->p, &h->self_shadow, sizeof(DevParams)) == 0) return ZENV_OK;
    h->self_shadow = h->p;
    HIP_TRY(hipMemcpyAsync(h->d_self, &h->self_shadow, sizeof(DevParams), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));   // the source is host memory that the next change overwrites
    h->self_valid = true;
    return ZENV_OK;
}

// A ring schedule is only as deep as its `depth` slots per env, and only the HOST refills them (zenv_bank_update), between
// calls: an env ends at most one episode per step, so a call of up to `depth` auto-resetting steps cannot outrun its
// ring; a longer one could wrap onto maps it has already played -- silently.  Refused instead.
static int ring_guard(const zenv *h, int steps, int auto_reset_every_step)
{
    if (h->p.sched_mode == SCHED_RING && auto_reset_every_step && steps > h->p.sched_stride)
        return fail(ZENV_E_STATE, "a ring schedule of depth %d is refilled by the host between calls: %d auto-resetting "
                    "steps in one call could replay maps (use calls of at most `depth` steps with zenv_bank_update in "
                    "between, a deeper ring, or a sequential schedule)", h->p.sched_stride, steps);
    return ZENV_OK;
}

// ============================================================================ action chunks
// K steps of caller-supplied actions.  With the persistent kernel available (lane layout, a compiled zone count, no
// goal / order post-kernels) a chunk is one launch per ZENV_ROLLOUT_CHUNK steps and slice; otherwise it is the plain
// sequence of single-step launches -- same results either way (tests/test_gpu_chunk.py).
static int chunk_reserve(zenv *h, int steps, const float *device_actions)
{
    const size_t N = (size_t)h->n_env, cells = (size_t)steps * N;
    if (cells > h->chunk_cap) {
        // growing frees the old allocation: a caller that replays ZENV_F_CHUNK_ACTIONS (zenv_device_ptr) must not do it
        // with more steps than the buffer was sized for
        const char *a = reinterpret_cast<const char *>(device_actions), *m = static_cast<const char *>(h->chunk_mem);
        if (a && m && a >= m && a < m + h->chunk_cap * (8 + 4 + 1) + 256)
            return fail(ZENV_E_ARG, "the actions lie inside the handle's own chunk buffer, which %d steps would outgrow "
                                    "(size it with one zenv_step_many of host actions first)", steps);
        HIP_TRY(hipStreamSynchronize(h->stream));
        if (h->chunk_mem) (void)hipFree(h->chunk_mem);
        h->chunk_mem = nullptr;
        h->chunk_actions = h->chunk_reward = nullptr;
        h->chunk_done = nullptr;
        h->chunk_cap = 0;
        HIP_TRY(hipMalloc(&h->chunk_mem, cells * (8 + 4 + 1) + 256));
        h->chunk_actions = static_cast<float *>(h->chunk_mem);                    // [K][N][2] float32
        h->chunk_reward = h->chunk_actions + 2 * cells;                           // [K][N] float32
        h->chunk_done = reinterpret_cast<uint8_t *>(h->chunk_reward + cells);     // [K][N] uint8
        h->chunk_cap = cells;
    } else {
        // the records of a shorter chunk sit at the front of the same allocation, re-laid-out for this call's K
        const size_t cap = h->chunk_cap;
        h->chunk_reward = h->chunk_actions + 2 * cap;
        h->chunk_done = reinterpret_cast<uint8_t *>(h->chunk_reward + cap);
    }
    return ZENV_OK;
}

extern "C" int zenv_step_many(zenv_t *h, const float *actions, int actions_on_device, int n_steps, int reset_mode)
{
    if (!h || !actions) return fail(ZENV_E_ARG, "null argument");
    if (!h->was_reset) return fail(ZENV_E_STATE, "Environment must be reset before stepping");
    if (n_steps < 1) return fail(ZENV_E_ARG, "n_steps must be >= 1");
    if (reset_mode != ZENV_CHUNK_NO_RESET && reset_mode != ZENV_CHUNK_RESET_EVERY && reset_mode != ZENV_CHUNK_RESET_LAST)
        return fail(ZENV_E_ARG, "unknown reset mode %d", reset_mode);
    if ((int64_t)n_steps * h->n_env > ((int64_t)1 << 31))
        return fail(ZENV_E_ARG, "%d steps x %d envs: split the chunk (the time-major records are indexed with 32 bits)", n_steps, h->n_env);
    if (int rr = ring_guard(h, n_steps, reset_mode == ZENV_CHUNK_RESET_EVERY)) return rr;
    int rc = use_device(h);
    if (rc) return rc;
    h->act_tag.valid = false;
    rc = chunk_reserve(h, n_steps, actions_on_device ? actions : nullptr);
    if (rc) return rc;
    const size_t N = (size_t)h->n_env;
    const float *d_act = actions;
    if (!actions_on_device) {
        HIP_TRY(hipMemcpyAsync(h->chunk_actions, actions, sizeof(float) * 2 * N * (size_t)n_steps, hipMemcpyHostToDevice, h->stream));
        d_act = h->chunk_actions;
        // like n_steps zenv_step() calls with host actions, the handle's action buffer ends up holding the last step's
        if (!h->host_io_actions)
            HIP_TRY(hipMemcpyAsync(h->p.actions, h->chunk_actions + 2 * N * (size_t)(n_steps - 1), sizeof(float) * 2 * N,
                                   hipMemcpyDeviceToDevice, h->stream));
    }
    h->chunk_steps = n_steps;
    const bool persistent = rollout_kernel_available(h->p) && !h->goal_enabled && !h->order_enabled;
    if (persistent) {
        rc = ensure_self(h);
        if (rc) return rc;
        const int all_tiles = rollout_tiles(h->p);
        const int slice = h->rollout_slice_tiles > 0 ? std::min(h->rollout_slice_tiles, all_tiles) : all_tiles;
        for (int t = 0; t < n_steps; t += kRolloutChunk) {
            const int k = std::min(kRolloutChunk, n_steps - t);
            const bool last = t + k == n_steps;
            const int mode = reset_mode == ZENV_CHUNK_RESET_EVERY ? 1 : (reset_mode == ZENV_CHUNK_RESET_LAST && last ? 2 : 0);
            const ChunkIO io{ reinterpret_cast<const float2 *>(d_act) + (size_t)t * N, h->chunk_reward + (size_t)t * N,
                              h->chunk_done + (size_t)t * N };
            for (int tile0 = 0; tile0 < all_tiles; tile0 += slice)
                HIP_TRY(launch_rollout_actions(h->p, k, mode, io, h->stream, nullptr, nullptr, tile0,
                                               std::min(slice, all_tiles - tile0)));
            h->step_count += k;
        }
        return ZENV_OK;
    }
    for (int t = 0; t < n_steps; ++t) {
        const int ar = reset_mode == ZENV_CHUNK_RESET_EVERY || (reset_mode == ZENV_CHUNK_RESET_LAST && t == n_steps - 1);
        HIP_TRY(launch_step(h->p, d_act + 2 * N * (size_t)t, ar, no_policy(), h->stream));
        if (h->goal_enabled) HIP_TRY(launch_goal_step(h->p, h->stream));
        if (h->order_enabled) HIP_TRY(launch_order_step(h->p, h->stream));
        HIP_TRY(hipMemcpyAsync(h->chunk_reward + (size_t)t * N, h->p.reward, N * sizeof(float), hipMemcpyDefault, h->stream));
        HIP_TRY(hipMemcpyAsync(h->chunk_done + (size_t)t * N, h->p.done_out, N, hipMemcpyDefault, h->stream));
        h->step_count += 1;
    }
    return ZENV_OK;
}

extern "C" int zenv_rollout(zenv_t *h, int steps, int policy, uint64_t policy_seed, uint64_t env_index0,
                            int auto_reset, int flags, int event_stride, float *ms_total, float *ms_step_kernel_avg)
{
    if (!h) return fail(ZENV_E_ARG, "null handle");
    if (!h->was_reset) return fail(ZENV_E_STATE, "Environment must be reset before stepping");
    if (steps < 0) return fail(ZENV_E_ARG, "steps must be >= 0");
    if (!policy_known(policy)) return fail(ZENV_E_ARG, "unknown policy %d", policy);
    if (policy_is_mlp(policy) && !h->mlp_ready) return fail(ZENV_E_STATE, "zenv_mlp_load first");
    if (h->goal_enabled || h->order_enabled)
        return fail(ZENV_E_STATE, "goal-conditioned / solver-ordered envs are stepped with zenv_step");
    if (int rr = ring_guard(h, steps, auto_reset)) return rr;
    int rc = use_device(h);
    if (rc) return rc;
    // the actor network is its own launch sequence: policy, then step, every step
    // ... and so is every policy of the wave-per-env layout (K1w has no fused policy)
    const bool fused = (flags & ZENV_ROLLOUT_UNFUSED) == 0 && !policy_is_mlp(policy) &&
                       h->p.kernel == ZENV_KERNEL_LANE_PER_ENV;
    const bool persistent = fused && (flags & ZENV_ROLLOUT_PER_STEP) == 0 && rollout_kernel_available(h->p);
    const bool per_kernel = ms_step_kernel_avg != nullptr && steps > 0;
    if (event_stride < 1) event_stride = 1;
    // persistent: one launch covers up to kRolloutChunk steps and every launch is timed
    // ... in slices of the batch (zenv_set_rollout_slice): a launch of more workgroups than the chip holds at once runs
    // its later workgroups wherever a slot frees up, and the placement "one env wave + one stream wave per SIMD" that
    // the first 1 024 tiles get is lost (measured: one launch over 131 072 envs takes 6.9 us per 65 536 env-steps, two
    // launches over 65 536 each 5.3)
    const int all_tiles = rollout_tiles(h->p);
    const int slice = h->rollout_slice_tiles > 0 ? std::min(h->rollout_slice_tiles, all_tiles) : all_tiles;
    const int n_slices = (all_tiles + slice - 1) / slice;
    const int n_sampled = !per_kernel ? 0
                          : persistent ? n_slices * ((steps + kRolloutChunk - 1) / kRolloutChunk)
                                       : (steps + event_stride - 1) / event_stride;
    const size_t need = 2 + 2 * (size_t)n_sampled;
    while (h->events.size() < need) {
        hipEvent_t ev;
        HIP_TRY(hipEventCreate(&ev));
        h->events.push_back(ev);
    }
    // a_0 is already in the action buffer when the previous fused rollout of the same action source ended here
    const bool have_a0 = fused && h->act_tag.valid && h->act_tag.policy == policy && h->act_tag.seed == policy_seed &&
                         h->act_tag.index0 == env_index0 && h->act_tag.step == h->step_count;
    h->act_tag.valid = false;
    if (persistent && steps > 0) {
        rc = ensure_self(h);
        if (rc) return rc;
    }
    // persistent launches that carry their own begin/end events need no bracket events: the span from the first
    // begin to the last end is the loop (two marker packets less on the stream -- it shows on a 20-step call)
    const bool own_bracket = !(persistent && per_kernel);
    if (own_bracket) HIP_TRY(hipEventRecord(h->events[0], h->stream));
    if (persistent && steps > 0) {
        StepPolicy pol{ policy, (uint32_t)h->step_count, policy_seed, env_index0, h->p.actions };
        if (!have_a0) HIP_TRY(launch_policy(h->p, pol, h->stream));   // a_0; every launch leaves the next action behind
        for (int t = 0, c = 0; t < steps; t += kRolloutChunk) {
            const int k = std::min(kRolloutChunk, steps - t);
            pol.step_index = (uint32_t)(h->step_count + 1);
            for (int tile0 = 0; tile0 < all_tiles; tile0 += slice, ++c) {
                hipEvent_t e0 = per_kernel ? h->events[2 + 2 * c] : nullptr;
                hipEvent_t e1 = per_kernel ? h->events[3 + 2 * c] : nullptr;
                HIP_TRY(launch_rollout(h->p, k, auto_reset, pol, h->stream, e0, e1, tile0, std::min(slice, all_tiles - tile0)));
            }
            h->step_count += k;
        }
    }
    for (int t = 0; t < steps && !persistent; ++t) {
        StepPolicy pol{ policy, (uint32_t)h->step_count, policy_seed, env_index0, h->p.actions };
        // a_t = pi(obs_t, t): one stand-alone policy launch per step (unfused) or only before
        // the first step (fused: every step kernel then leaves a_{t+1} in the action buffer)
        if (!fused || (t == 0 && !have_a0)) {
            rc = run_policy(h, pol);
            if (rc) return rc;
        }
        StepPolicy next = no_policy();
        if (fused) {
            next = pol;
            next.step_index = (uint32_t)(h->step_count + 1);
        }
        const bool sampled = per_kernel && (t % event_stride == 0);
        hipEvent_t e0 = sampled ? h->events[2 + 2 * (t / event_stride)] : nullptr;
        hipEvent_t e1 = sampled ? h->events[3 + 2 * (t / event_stride)] : nullptr;
        HIP_TRY(launch_step(h->p, h->p.actions, auto_reset, next, h->stream, e0, e1));
        h->step_count += 1;
    }
    if (own_bracket) HIP_TRY(hipEventRecord(h->events[1], h->stream));
    if (fused && (steps > 0 || have_a0)) {
        h->act_tag.valid = true;
        h->act_tag.policy = policy;
        h->act_tag.seed = policy_seed;
        h->act_tag.index0 = env_index0;
        h->act_tag.step = h->step_count;
    }
    if (flags & ZENV_ROLLOUT_ASYNC) {
        // enqueue only: the caller overlaps host work (bank refills, a PPO update) with the launches and collects with
        // zenv_query() / zenv_sync(); no times are reported
        if (ms_total) *ms_total = -1.f;
        if (ms_step_kernel_avg) *ms_step_kernel_avg = -1.f;
        return ZENV_OK;
    }
    hipEvent_t ev_first = own_bracket ? h->events[0] : h->events[2];
    hipEvent_t ev_last = own_bracket ? h->events[1] : h->events[3 + 2 * (n_sampled - 1)];
    HIP_TRY(hipEventSynchronize(ev_last));
    if (int rr = mlp_range_check(h)) return rr;
    if (ms_total) HIP_TRY(hipEventElapsedTime(ms_total, ev_first, ev_last));
    if (per_kernel) {
        double sum = 0.0;
        for (int i = 0; i < n_sampled; ++i) {
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, h->events[2 + 2 * i], h->events[3 + 2 * i]));
            sum += ms;
        }
        // persistent: the launches cover all steps, so this is the per-step share of kernel time
        *ms_step_kernel_avg = (float)(persistent ? sum / steps : sum / n_sampled);
    } else if (ms_step_kernel_avg) {
        *ms_step_kernel_avg = 0.f;
    }
    return ZENV_OK;
}

// ============================================================================ results
extern "C" int64_t zenv_field_bytes(const zenv_t *h, int field) { return h ? field_info(h, field).bytes : 0; }

extern "C" int zenv_get(zenv_t *h, int field, void *dst, int dst_on_device)
{
    if (!h || !dst) return fail(ZENV_E_ARG, "null argument");
    const FieldInfo f = field_info(h, field);
    if (!f.ptr) return fail(ZENV_E_ARG, "unknown field %d", field);
    int rc = use_device(h);
    if (rc) return rc;
    if (int rf = refresh_field(h, field)) return rf;
    HIP_TRY(hipMemcpyAsync(dst, f.ptr, f.bytes, hipMemcpyDefault, h->stream));   // (a zenv_host_io field is host memory)
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (int rr = mlp_range_check(h)) return rr;
    return ZENV_OK;
}

extern "C" int zenv_get_rows(zenv_t *h, int field, int first_env, int count, void *dst)
{
    if (!h || !dst) return fail(ZENV_E_ARG, "null argument");
    const FieldInfo f = field_info(h, field);
    if (!f.ptr) return fail(ZENV_E_ARG, "unknown field %d", field);
    if (first_env < 0 || count < 0 || (int64_t)first_env + count > h->n_env)
        return fail(ZENV_E_ARG, "envs [%d, %d) outside the batch of %d", first_env, first_env + count, h->n_env);
    if (f.bytes % h->n_env) return fail(ZENV_E_ARG, "field %d is not laid out per env", field);
    int rc = use_device(h);
    if (rc) return rc;
    const int64_t per_env = f.bytes / h->n_env;
    if (int rf = refresh_field(h, field)) return rf;
    HIP_TRY(hipMemcpyAsync(dst, static_cast<const char *>(f.ptr) + per_env * first_env, (size_t)(per_env * count),
                           hipMemcpyDefault, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (int rr = mlp_range_check(h)) return rr;
    return ZENV_OK;
}

extern "C" void *zenv_host_alloc(int64_t bytes)
{
    void *p = nullptr;
    if (bytes <= 0 || hipHostMalloc(&p, (size_t)bytes, hipHostMallocDefault) != hipSuccess) {
        fail(ZENV_E_HIP, "hipHostMalloc(%lld) failed", (long long)bytes);
        return nullptr;
    }
    return p;
}

extern "C" int zenv_host_free(void *ptr)
{
    if (ptr) HIP_TRY(hipHostFree(ptr));
    return ZENV_OK;
}

extern "C" int zenv_get_many(zenv_t *h, int n_fields, const int *fields, void *const *dst)
{
    if (!h || !fields || !dst || n_fields < 0) return fail(ZENV_E_ARG, "bad argument");
    int rc = use_device(h);
    if (rc) return rc;
    for (int i = 0; i < n_fields; ++i) {
        const FieldInfo f = field_info(h, fields[i]);
        if (!f.ptr || !dst[i]) return fail(ZENV_E_ARG, "unknown field %d or null destination", fields[i]);
        if (int rf = refresh_field(h, fields[i])) return rf;
        HIP_TRY(hipMemcpyAsync(dst[i], f.ptr, f.bytes, hipMemcpyDefault, h->stream));
    }
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (int rr = mlp_range_check(h)) return rr;
    return ZENV_OK;
}

extern "C" int64_t zenv_results_layout(const zenv_t *h, int64_t *offsets)
{
    if (!h) return 0;
    if (offsets)
        for (int i = 0; i < ZENV_N_RESULTS; ++i) offsets[i] = h->results_off[i];
    return h->results_bytes;
}

extern "C" int zenv_step_results(zenv_t *h, const float *actions, int auto_reset, void *host_slab)
{
    if (!h || !host_slab) return fail(ZENV_E_ARG, "null argument");
    if (actions) {
        const int rc = zenv_step(h, actions, 0, auto_reset);
        if (rc) return rc;
    } else {
        const int rc = use_device(h);
        if (rc) return rc;
    }
    if (h->host_io_slab) {                       // the kernels wrote the results into host memory themselves
        HIP_TRY(hipStreamSynchronize(h->stream));
        if (host_slab != h->host_io_slab) std::memcpy(host_slab, h->host_io_slab, (size_t)h->results_bytes);
    } else {
        HIP_TRY(hipMemcpyAsync(host_slab, h->results_slab, (size_t)h->results_bytes, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
    }
    if (int rr = mlp_range_check(h)) return rr;
    return ZENV_OK;
}

extern "C" int zenv_host_io(zenv_t *h, int enable, void **results, float **actions)
{
    if (!h) return fail(ZENV_E_ARG, "null handle");
    int rc = use_device(h);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    const size_t act_bytes = sizeof(float) * 2 * (size_t)h->n_env;
    if (enable && !h->host_io_slab) {
        void *slab = nullptr, *act = nullptr;
        HIP_TRY(hipHostMalloc(&slab, (size_t)h->results_bytes, hipHostMallocDefault));
        hipError_t err = hipHostMalloc(&act, act_bytes, hipHostMallocDefault);
        if (err == hipSuccess) err = hipMemcpy(slab, h->results_slab, (size_t)h->results_bytes, hipMemcpyDefault);
        if (err == hipSuccess) err = hipMemcpy(act, h->p.actions, act_bytes, hipMemcpyDefault);
        if (err != hipSuccess) {
            (void)hipHostFree(slab);
            if (act) (void)hipHostFree(act);
            return fail(ZENV_E_HIP, "zenv_host_io: %s", hipGetErrorString(err));
        }
        h->host_io_slab = slab;
        h->host_io_actions = static_cast<float *>(act);
        h->dev_actions = h->p.actions;
        h->p.actions = h->host_io_actions;
        for (Alloc &a : h->allocs)
            if (a.slab_off >= 0) *a.slot = static_cast<char *>(slab) + a.slab_off;
        h->act_tag.valid = false;
    } else if (!enable && h->host_io_slab) {
        HIP_TRY(hipMemcpy(h->results_slab, h->host_io_slab, (size_t)h->results_bytes, hipMemcpyDefault));
        HIP_TRY(hipMemcpy(h->dev_actions, h->host_io_actions, act_bytes, hipMemcpyDefault));
        h->p.actions = h->dev_actions;
        for (Alloc &a : h->allocs)
            if (a.slab_off >= 0) *a.slot = static_cast<char *>(h->results_slab) + a.slab_off;
        (void)hipHostFree(h->host_io_slab);
        (void)hipHostFree(h->host_io_actions);
        h->host_io_slab = nullptr;
        h->host_io_actions = h->dev_actions = nullptr;
        h->act_tag.valid = false;
    }
    if (results) *results = h->host_io_slab;
    if (actions) *actions = h->host_io_actions;
    return ZENV_OK;
}

extern "C" int zenv_step_host(zenv_t *h, int auto_reset)
{
    if (!h) return fail(ZENV_E_ARG, "null handle");
    if (!h->host_io_slab) return fail(ZENV_E_STATE, "zenv_host_io(h, 1, ...) first");
    const int rc = zenv_step(h, h->host_io_actions, 0, auto_reset);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (int rr = mlp_range_check(h)) return rr;
    return ZENV_OK;
}

extern "C" int zenv_device_ptr(zenv_t *h, int field, void **ptr)
{
    if (!h || !ptr) return fail(ZENV_E_ARG, "null argument");
    const FieldInfo f = field_info(h, field);
    if (!f.ptr) return fail(ZENV_E_ARG, "unknown field %d", field);
    if (field == ZENV_F_EP_RETURN || field == ZENV_F_EP_LEN) {
        // these two live inside the step kernels' records: the pointer is to a plain copy brought up to date by THIS
        // call (stream-ordered), not a live view
        int rc = use_device(h);
        if (rc) return rc;
        if (int rf = refresh_field(h, field)) return rf;
    }
    *ptr = f.ptr;
    return ZENV_OK;
}

// ============================================================================ multi-GPU: the one collective, native RCCL
// Envs shard trivially over GPUs (one process per GPU, no exchange on the step path); a job's only collective is the
// all-gather of per-env episodic figures after a rollout.  This replaces the reference's Pipe star
// (main/src/torch_ac/torch_utils/penv.py:26-40) at N > 1.  librccl is loaded on first use -- a single-GPU caller never
// maps its half gigabyte -- preferring a copy the process already holds (a torch in the same process brings its own).
namespace {

struct RcclApi {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
    std::string path;
    std::string load_error;   // the loader's message when no copy could be opened
};

void rccl_load(RcclApi &api);
RcclApi g_rccl;

RcclApi *rccl_api()
{
    static std::once_flag once;          // two handles may set up communicators from different threads
    std::call_once(once, [] { rccl_load(g_rccl); });
    return g_rccl.lib ? &g_rccl : nullptr;
}

void rccl_load(RcclApi &api)
{
    const char *names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
    void *lib = nullptr;
    for (const char *n : names)                       // a copy that is already mapped wins (torch's bundled one)
        if (!lib && (lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_LOCAL))) api.path = std::string(n) + " (already loaded)";
    if (const char *e = std::getenv("ZENV_RCCL_PATH"))
        if (!lib && (lib = dlopen(e, RTLD_NOW | RTLD_LOCAL))) api.path = e;
    for (const char *n : names) {
        if (lib) break;
        if ((lib = dlopen(n, RTLD_NOW | RTLD_LOCAL))) {
            api.path = n;
        } else if (const char *e = dlerror()) {         // dlerror() clears itself: read once, keep the text
            api.load_error = e;
        }
    }
    if (!lib) return;
#define ZENV_SYM(name) api.name = reinterpret_cast<decltype(api.name)>(dlsym(lib, "nccl" #name))
    ZENV_SYM(GetUniqueId); ZENV_SYM(CommInitRank); ZENV_SYM(CommDestroy); ZENV_SYM(AllGather); ZENV_SYM(AllReduce);
    ZENV_SYM(GetErrorString); ZENV_SYM(GetVersion);
#undef ZENV_SYM
    if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllGather || !api.AllReduce || !api.GetErrorString) {
        api.load_error = api.path + " lacks an nccl* entry point this library binds";
        return;
    }
    api.lib = lib;
}

const char *rccl_load_error() { return g_rccl.load_error.empty() ? "no loader message" : g_rccl.load_error.c_str(); }

#define RCCL_TRY(api, expr)                                                                                   \
    do {                                                                                                      \
        ncclResult_t _r = (expr);                                                                             \
        if (_r != ncclSuccess)                                                                                \
            return fail(ZENV_E_HIP, "%s failed: %s (%s:%d)", #expr, (api)->GetErrorString(_r), __FILE__, __LINE__); \
    } while (0)

}  // namespace

extern "C" int zenv_comm_unique_id(void *id_out)
{
    if (!id_out) return fail(ZENV_E_ARG, "null argument");
    static_assert(sizeof(ncclUniqueId) == ZENV_COMM_ID_BYTES, "ZENV_COMM_ID_BYTES must be sizeof(ncclUniqueId)");
    RcclApi *api = rccl_api();
    if (!api) return fail(ZENV_E_HIP, "librccl not found (%s)", rccl_load_error());
    ncclUniqueId id;
    RCCL_TRY(api, api->GetUniqueId(&id));
    std::memcpy(id_out, &id, sizeof(id));
    return ZENV_OK;
}

extern "C" int zenv_comm_init(zenv_t *h, int rank, int world, const void *unique_id)
{
    if (!h || !unique_id) return fail(ZENV_E_ARG, "null argument");
    if (world < 1 || rank < 0 || rank >= world) return fail(ZENV_E_ARG, "rank %d outside [0,%d)", rank, world);
    if (h->comm) return fail(ZENV_E_STATE, "the handle already has a communicator");
    RcclApi *api = rccl_api();
    if (!api) return fail(ZENV_E_HIP, "librccl not found (%s)", rccl_load_error());
    int rc = use_device(h);
    if (rc) return rc;
    ncclUniqueId id;
    std::memcpy(&id, unique_id, sizeof(id));
    RCCL_TRY(api, api->CommInitRank(&h->comm, world, id, rank));
    h->comm_rank = rank;
    h->comm_world = world;
    const size_t N = (size_t)h->n_env;
    hipError_t err = hipMalloc(&h->comm_send, N * 4);
    if (err == hipSuccess) err = hipMalloc(&h->comm_recv, N * 4 * (size_t)world);
    if (err == hipSuccess) err = hipMalloc(reinterpret_cast<void **>(&h->comm_scalar), 2 * sizeof(double));
    if (err != hipSuccess) {
        (void)zenv_comm_destroy(h);      // no half-built communicator: the next allgather would launch on null buffers
        return fail(ZENV_E_HIP, "zenv_comm_init: hipMalloc: %s", hipGetErrorString(err));
    }
    return ZENV_OK;
}

extern "C" int zenv_comm_destroy(zenv_t *h)
{
    if (!h) return fail(ZENV_E_ARG, "null handle");
    if (!h->comm) return ZENV_OK;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    RcclApi *api = rccl_api();
    if (api) (void)api->CommDestroy(h->comm);
    h->comm = nullptr;
    h->comm_world = 0;
    for (void **m : { &h->comm_send, &h->comm_recv, reinterpret_cast<void **>(&h->comm_scalar) }) {
        if (*m) (void)hipFree(*m);
        *m = nullptr;
    }
    return ZENV_OK;
}

extern "C" int zenv_comm_info(const zenv_t *h, int *rank, int *world, const char **library)
{
    if (!h) return fail(ZENV_E_ARG, "null handle");
    if (rank) *rank = h->comm_rank;
    if (world) *world = h->comm_world;
    if (library) {
        RcclApi *api = h->comm ? rccl_api() : nullptr;
        *library = api ? api->path.c_str() : "";
    }
    return ZENV_OK;
}

extern "C" int zenv_allgather(zenv_t *h, int field, void *dst, int dst_on_device)
{
    if (!h || !dst) return fail(ZENV_E_ARG, "null argument");
    if (!h->comm) return fail(ZENV_E_STATE, "zenv_comm_init first");
    // the documented per-env figures only: float64 narrowed to float32, float32 and int32 as they are (a float2 action
    // or an int64 seed is N * 8 bytes too, and would be gathered as garbage)
    const bool as_f64 = field == ZENV_F_LAST_RETURN || field == ZENV_F_EP_RETURN;
    const bool as_f32 = field == ZENV_F_REWARD;
    const bool as_i32 = field == ZENV_F_EPISODES || field == ZENV_F_LAST_LEN || field == ZENV_F_VISIT_COUNT || field == ZENV_F_EP_LEN;
    if (!as_f64 && !as_f32 && !as_i32)
        return fail(ZENV_E_ARG, "field %d is not gathered: LAST_RETURN, EP_RETURN, REWARD, EPISODES, LAST_LEN, VISIT_COUNT, EP_LEN are", field);
    const FieldInfo f = field_info(h, field);
    const int64_t N = h->n_env;
    if (!f.ptr || f.bytes != N * (as_f64 ? 8 : 4)) return fail(ZENV_E_ARG, "field %d has an unexpected extent", field);
    if (!h->comm_send || !h->comm_recv) return fail(ZENV_E_STATE, "the communicator has no buffers (zenv_comm_init failed)");
    RcclApi *api = rccl_api();
    int rc = use_device(h);
    if (rc) return rc;
    const bool is_f32 = as_f64 || as_f32;
    if (int rf = refresh_field(h, field)) return rf;
    HIP_TRY(launch_gather_prep(f.ptr, (int)(f.bytes / N), h->comm_send, h->n_env, h->stream));
    RCCL_TRY(api, api->AllGather(h->comm_send, h->comm_recv, (size_t)N, is_f32 ? ncclFloat32 : ncclInt32, h->comm, h->stream));
    HIP_TRY(hipMemcpyAsync(dst, h->comm_recv, (size_t)N * 4 * (size_t)h->comm_world,
                           dst_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return ZENV_OK;
}

extern "C" int zenv_comm_barrier(zenv_t *h)
{
    if (!h) return fail(ZENV_E_ARG, "null handle");
    if (!h->comm) return fail(ZENV_E_STATE, "zenv_comm_init first");
    RcclApi *api = rccl_api();
    int rc = use_device(h);
    if (rc) return rc;
    // every rank's stream reaches this point, then a one-element sum goes round: nobody leaves before everybody came
    RCCL_TRY(api, api->AllReduce(h->comm_scalar, h->comm_scalar + 1, 1, ncclFloat64, ncclSum, h->comm, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return ZENV_OK;
}

extern "C" int zenv_comm_allreduce_max(zenv_t *h, double *value)
{
    if (!h || !value) return fail(ZENV_E_ARG, "null argument");
    if (!h->comm) return fail(ZENV_E_STATE, "zenv_comm_init first");
    RcclApi *api = rccl_api();
    int rc = use_device(h);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(h->comm_scalar, value, sizeof(double), hipMemcpyHostToDevice, h->stream));
    RCCL_TRY(api, api->AllReduce(h->comm_scalar, h->comm_scalar + 1, 1, ncclFloat64, ncclMax, h->comm, h->stream));
    HIP_TRY(hipMemcpyAsync(value, h->comm_scalar + 1, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return ZENV_OK;
}

// measurement utility: the attainable rate of a write-only row stream of the step kernels' shape on this device
extern "C" int zenv_probe_store_stream(int device, int64_t n_tiles, int tile_bytes, int steps, int cache_policy,
                                       int reps, float *us_per_step)
{
    if (!us_per_step) return fail(ZENV_E_ARG, "null argument");
    if (n_tiles < 1 || n_tiles > (1ll << 24) || tile_bytes < 1024 || tile_bytes % 16 || steps < 1 || reps < 1)
        return fail(ZENV_E_ARG, "bad probe shape");
    if (cache_policy != 0 && cache_policy != 2 && cache_policy != 16)
        return fail(ZENV_E_ARG, "cache_policy is 0 (plain), 2 (nt) or 16 (sc1)");
    HIP_TRY(hipSetDevice(device));
    float *buf = nullptr;
    hipStream_t s = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&buf), (size_t)n_tiles * (size_t)tile_bytes));
    hipError_t err = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    if (err == hipSuccess) err = hipEventCreate(&e0);
    if (err == hipSuccess) err = hipEventCreate(&e1);
    float best = 0.f;
    for (int r = 0; r <= reps && err == hipSuccess; ++r) {          // r == 0: untimed warm-up
        err = launch_probe_store(buf, n_tiles, tile_bytes, steps, cache_policy, s, e0, e1);
        if (err == hipSuccess) err = hipEventSynchronize(e1);
        float ms = 0.f;
        if (err == hipSuccess) err = hipEventElapsedTime(&ms, e0, e1);
        if (r >= 1 && (r == 1 || ms < best)) best = ms;
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (s) (void)hipStreamDestroy(s);
    (void)hipFree(buf);
    if (err != hipSuccess) return fail(ZENV_E_HIP, "store-stream probe: %s", hipGetErrorString(err));
    *us_per_step = best * 1e3f / (float)steps;
    return ZENV_OK;
}

// diagnostic: copy the per-block phase stamps of the last step launch (ZENV_STAMPS builds)
extern "C" int zenv_debug_stamps(zenv_t *h, unsigned long long *dst, int64_t count)
{
    if (!h || !dst) return fail(ZENV_E_ARG, "null argument");
    const int64_t have = 16 * (((int64_t)h->n_env + 63) / 64);
    if (count != have) return fail(ZENV_E_ARG, "stamp buffer holds %lld entries", (long long)have);
    int rc = use_device(h);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipMemcpy(dst, h->p.dbg, have * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemsetAsync(h->p.dbg, 0, have * sizeof(unsigned long long), h->stream));
    return ZENV_OK;
}

extern "C" int zenv_sync(zenv_t *h)
{
    if (!h) return fail(ZENV_E_ARG, "null handle");
    int rc = use_device(h);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (int rr = mlp_range_check(h)) return rr;
    return ZENV_OK;
}

extern "C" int zenv_set_rollout_slice(zenv_t *h, int envs_per_launch)
{
    if (!h) return fail(ZENV_E_ARG, "null handle");
    if (envs_per_launch < 0) return fail(ZENV_E_ARG, "envs_per_launch must be >= 0 (0: the whole batch in one launch)");
    h->rollout_slice_tiles = (envs_per_launch + 63) / 64;
    return ZENV_OK;
}

extern "C" int zenv_query(zenv_t *h)
{
    if (!h) return fail(ZENV_E_ARG, "null handle");
    int rc = use_device(h);
    if (rc) return rc;
    const hipError_t e = hipStreamQuery(h->stream);
    if (e == hipSuccess) return 1;
    if (e == hipErrorNotReady) {
        (void)hipGetLastError();      // not an error: clear the sticky code
        return 0;
    }
    return fail(ZENV_E_HIP, "hipStreamQuery: %s", hipGetErrorString(e));
}

extern "C" int64_t zenv_step_count(const zenv_t *h) { return h ? h->step_count : 0; }

// ============================================================================ snapshots
extern "C" int64_t zenv_state_bytes(const zenv_t *h)
{
    if (!h) return 0;
    int64_t total = 16;   // step_count + flags
    for (const Alloc &a : h->allocs)
        if (a.is_state) total += (int64_t)a.bytes;
    return total;
}

extern "C" int zenv_get_state(zenv_t *h, void *dst, int64_t bytes)
{
    if (!h || !dst) return fail(ZENV_E_ARG, "null argument");
    if (bytes != zenv_state_bytes(h)) return fail(ZENV_E_ARG, "state blob is %lld bytes", (long long)zenv_state_bytes(h));
    int rc = use_device(h);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    char *out = static_cast<char *>(dst);
    int64_t head[2] = { h->step_count, (int64_t)h->p.sched_mode | ((int64_t)h->was_reset << 8) };
    std::memcpy(out, head, 16);
    out += 16;
    for (const Alloc &a : h->allocs) {
        if (!a.is_state) continue;
        HIP_TRY(hipMemcpy(out, *a.slot, a.bytes, hipMemcpyDefault));
        if (a.slot == (void **)&h->p.sched) {
            // the reset hint rides in the schedule record but is a prefetch hint, not state (the persistent kernel
            // does not maintain it): snapshots of equal states are equal bytes
            Sched *sc = reinterpret_cast<Sched *>(out);
            for (int i = 0; i < h->n_env; ++i) sc[i].reset_hint = -1;
        }
        out += a.bytes;
    }
    return ZENV_OK;
}

extern "C" int zenv_set_state(zenv_t *h, const void *src, int64_t bytes)
{
    if (!h || !src) return fail(ZENV_E_ARG, "null argument");
    if (bytes != zenv_state_bytes(h)) return fail(ZENV_E_ARG, "state blob is %lld bytes", (long long)zenv_state_bytes(h));
    int rc = use_device(h);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    const char *in = static_cast<const char *>(src);
    int64_t head[2];
    std::memcpy(head, in, 16);
    in += 16;
    {
        // MuJoCo resets a state that holds a NaN / Inf (mj_checkPos / mj_checkVel) and the kernels rely on a finite one
        // (include/zenv.h, exception path): a blob with a non-finite joint state or placement is refused
        const char *scan = in;
        for (const Alloc &a : h->allocs) {
            if (!a.is_state) continue;
            void **s0 = a.slot;
            if (s0 == (void **)&h->p.qa || s0 == (void **)&h->p.qb || s0 == (void **)&h->p.qc ||
                s0 == (void **)&h->p.fa || s0 == (void **)&h->p.fb) {
                const double *d = reinterpret_cast<const double *>(scan);
                for (size_t i = 0; i < a.bytes / sizeof(double); ++i)
                    if (!std::isfinite(d[i])) return fail(ZENV_E_ARG, "state blob holds a non-finite joint state");
                // the kernels turn sin / cos of the hinge angle by h * omega per substep with Taylor kernels that
                // validate_config keeps below 0.05 rad for every state the model can REACH; a blob can hold any
                // finite velocity, so the same bound is applied to it (qc = (v1, v2): omega is the second double)
                if (s0 == (void **)&h->p.qc)
                    for (size_t i = 1; i < a.bytes / sizeof(double); i += 2)
                        if (!(h->cfg.timestep * std::fabs(d[i]) < 0.05))
                            return fail(ZENV_E_ARG, "state blob: env %zu turns %.3g rad per substep, beyond the small-angle "
                                        "update (0.05)", i / 2, h->cfg.timestep * std::fabs(d[i]));
            }
            scan += a.bytes;
        }
    }
    for (const Alloc &a : h->allocs) {
        if (!a.is_state) continue;
        HIP_TRY(hipMemcpy(*a.slot, in, a.bytes, hipMemcpyDefault));
        in += a.bytes;
    }
    h->step_count = head[0];
    h->was_reset = ((head[1] >> 8) & 1) != 0;
    h->act_tag.valid = false;
    return ZENV_OK;
}

extern "C" int zenv_debug_state(zenv_t *h, double *qpos, double *qvel, int32_t *zone_state, int32_t *cooldown,
                                int32_t *steps)
{
    if (!h) return fail(ZENV_E_ARG, "null handle");
    int rc = use_device(h);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    const size_t N = h->n_env, Z = h->cfg.num_zones;
    const DevParams &p = h->p;
    if (qpos || qvel) {
        std::vector<double2> qa(N), qb(N), qc(N);
        HIP_TRY(hipMemcpy(qa.data(), p.qa, N * sizeof(double2), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(qb.data(), p.qb, N * sizeof(double2), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(qc.data(), p.qc, N * sizeof(double2), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < N; ++i) {
            if (qpos) { qpos[3 * i] = qa[i].x; qpos[3 * i + 1] = qa[i].y; qpos[3 * i + 2] = qb[i].x; }
            if (qvel) { qvel[3 * i] = qb[i].y; qvel[3 * i + 1] = qc[i].x; qvel[3 * i + 2] = qc[i].y; }
        }
    }
    std::vector<HotA> ha(N);
    std::vector<HotC> hc(N);
    HIP_TRY(hipMemcpy(ha.data(), p.hota, N * sizeof(HotA), hipMemcpyDefault));
    HIP_TRY(hipMemcpy(hc.data(), p.hotc, N * sizeof(HotC), hipMemcpyDefault));
    if (zone_state) {
        for (size_t i = 0; i < N; ++i)
            for (size_t z = 0; z < Z; ++z)
                zone_state[i * Z + z] = h->cfg.task == ZENV_TASK_COLOUR_MATCH ? (int32_t)((hc[i].colpack >> (2 * z)) & 3ull)
                                                                               : (int32_t)((ha[i].vis >> z) & 1u);
    }
    if (cooldown) {
        const size_t ZW = (Z + 7) / 8;
        std::vector<uint8_t> cd(std::max<size_t>(1, ZW - 1) * 8 * N);      // zones 8 and up
        HIP_TRY(hipMemcpy(cd.data(), p.cooldown, cd.size(), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < N; ++i)
            for (size_t z = 0; z < Z; ++z)
                cooldown[i * Z + z] = z < 8 ? (int32_t)((hc[i].cd0 >> (8 * z)) & 0xFFull) : cd[(((z >> 3) - 1) * N + i) * 8 + (z & 7)];
    }
    if (steps)
        for (size_t i = 0; i < N; ++i) steps[i] = ha[i].steps;
    return ZENV_OK;
}

/*
 * zenv.h -- C ABI of the MI355X-native batched zone-env step path.
 *
 * Drop-in boundary for the env.step()/reset() hot path of
 * andrewli77/combinatorial-rl-tasks (paths below are relative to the reference root):
 *   main/envs/TSP_env.py, main/envs/TTSP_env.py, main/envs/colour_match_env.py,
 *   main/envs/zone_envs/ZoneEnvBase.py, main/envs/wrappers.py (FixedSeedsWrapper,
 *   ZoneWrapper), main/src/torch_ac/torch_utils/penv.py (ParallelEnv) and, beneath them,
 *   safety_gym Engine.step/reset + MuJoCo mj_step for xmls/point.xml (not vendored).
 *
 * Conventions: every entry point is extern "C", takes plain pointers and sizes, returns
 * 0 on success or a negative ZENV_E_* code (text from zenv_last_error()); no exception
 * crosses the boundary.  The caller owns every buffer it passes; the library owns the
 * handle and its device memory until zenv_destroy().  A handle is bound to one device and
 * one HIP stream and is not thread-safe.  zenv_step()/zenv_reset()/zenv_policy()/zenv_bank_update() are
 * asynchronous on the handle's stream; zenv_get*(), zenv_sync() and -- unless ZENV_ROLLOUT_ASYNC is set --
 * zenv_rollout() synchronise.
 *
 * There is no CPU fallback: every compute entry point fails with ZENV_E_HIP when no
 * gfx950 device is usable.
 */
#ifndef ZENV_H
#define ZENV_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZENV_MAX_ZONES 32
#define ZENV_OBS_DIM 8

/* tasks: envs/__init__.py:88-141 registry ids PointTSP-v0/v1, PointTTSP-v0/v1, ColourMatch-v0 */
enum { ZENV_TASK_TSP = 0, ZENV_TASK_TIMED_TSP = 1, ZENV_TASK_COLOUR_MATCH = 2 };

/* error codes */
enum {
    ZENV_OK = 0,
    ZENV_E_ARG = -1,        /* bad argument */
    ZENV_E_HIP = -2,        /* HIP runtime / no device */
    ZENV_E_STATE = -3,      /* call order (e.g. step before bank/reset) */
    ZENV_E_LAYOUT = -4,     /* ResamplingError: no layout in 10000 tries */
    ZENV_E_DONE = -5,       /* single-env semantics: 'Environment must be reset before stepping' */
    ZENV_E_RANGE = -6       /* ZENV_MLP_F16X3: a weight, input or activation of the network beyond float16's range */
};

/* zenv_get()/zenv_device_ptr() selectors */
enum {
    ZENV_F_OBS = 0,         /* float32 [N,8]   wrappers.py:136-142 'obs' = remaining,pos,dir,velp,velr */
    ZENV_F_ZONE_OBS = 1,    /* float32 [N,Z,F] wrappers.py:137 'zone_obs' */
    ZENV_F_REWARD = 2,      /* float32 [N] */
    ZENV_F_DONE = 3,        /* uint8   [N]     done flag returned by the last step */
    ZENV_F_GOAL_MET = 4,    /* uint8   [N]     info['goal_met'] (evaluate.py:65) */
    ZENV_F_EP_RETURN = 5,   /* float64 [N]     undiscounted return of the running episode */
    ZENV_F_EP_LEN = 6,      /* int32   [N]     steps of the running episode */
    ZENV_F_LAST_RETURN = 7, /* float64 [N]     return of the last finished episode (evaluate.py:62-72) */
    ZENV_F_LAST_LEN = 8,    /* int32   [N] */
    ZENV_F_EPISODES = 9,    /* int32   [N]     episodes finished so far */
    ZENV_F_VISIT_COUNT = 10,/* int32   [N]     visited zones (TSP/Timed) or goal_dist (ColourMatch) */
    ZENV_F_SEED = 11,       /* int64   [N]     env seed of the running episode */
    ZENV_F_ACTIONS = 12,    /* float32 [N,2]   internal action buffer (zenv_policy target) */
    ZENV_F_POLICY_MU = 13,  /* float32 [N,2]   mean of the actor's Normal (after zenv_mlp_forward) */
    ZENV_F_POLICY_STD = 14, /* float32 [N,2]   its standard deviation */
    ZENV_F_POLICY_VALUE = 15, /* float32 [N]   the critic's value (when critic weights were loaded) */
    /* goal-conditioned variant, after zenv_goal_enable(): */
    ZENV_F_SHAPED_REWARD = 16,   /* float64 [N] info['shaped_reward'] (TSP_next_city_env.py:60-66) */
    ZENV_F_NEED_GOAL = 17,       /* uint8   [N] info['need_next_goal'] / env.goal_zone is None (:69-75) */
    ZENV_F_AVAILABLE_GOALS = 18, /* uint32  [N] get_available_goals() as a bit mask, bit z = zone z unvisited */
    ZENV_F_GOAL = 19,            /* int32   [N] goal zone, -1 = none */
    /* experience buffers of the last zenv_collect(), time-major [T][N][...]; their [N][T] transposes, flattened, are
     * exps.* of base.py:125-128, :211-227 */
    ZENV_F_EXP_OBS = 20,         /* float32 [T,N,8]    every ZENV_F_EXP_* buffer is time-major (frame t of all envs is
                                  *                    contiguous); the observations are written in place by the step kernel */
    ZENV_F_EXP_ZONE_OBS = 21,    /* float32 [T,N,Z,F] */
    ZENV_F_EXP_ACTION = 22,      /* float32 [T,N,2] */
    ZENV_F_EXP_LOG_PROB = 23,    /* float32 [T,N,2]  Normal(mu, std).log_prob(action) */
    ZENV_F_EXP_VALUE = 24,       /* float32 [T,N] */
    ZENV_F_EXP_REWARD = 25,      /* float32 [T,N]    shaped_reward when the handle is goal-conditioned (:153-159) */
    ZENV_F_EXP_MASK = 26,        /* float32 [T,N]    1 - done of the previous step (:149-150) */
    ZENV_F_EXP_ADVANTAGE = 27,   /* float32 [T,N]    GAE (:190-196) */
    ZENV_F_EXP_RETURN = 28,      /* float32 [T,N]    value + advantage (:226) */
    ZENV_F_ORDER_VAL = 29,       /* float32 [N,Z]    TSPOrderEnv's 7th row feature 0.5^(position in the route), 0 when visited */
    ZENV_F_POLICY_VALUE_SIGMA = 31, /* float32 [N]   the distributional critic's sigma (flat_model.py:57-60) */
    ZENV_F_EXCEPTION = 30,       /* uint8   [N]      info['exception'] of the env's LAST FINISHED episode: 1 = it was ended by
                                  *                    Engine.step's MujocoException path (valid once done; see zenv_step) */
    ZENV_F_ORDER_POS = 32,       /* int8    [N,Z]    TSPOrderEnv's self.route: position of every zone in the remaining route, -1 =
                                  *                    not in it (visited) */
    /* time-major records of the last zenv_step_many(): every step of the chunk */
    ZENV_F_CHUNK_REWARD = 33,    /* float32 [K,N] */
    ZENV_F_CHUNK_DONE = 34,      /* uint8   [K,N] */
    ZENV_F_CHUNK_ACTIONS = 35,   /* float32 [K,N,2]  the device copy of the last chunk whose actions came from the host (a caller that
                                  *                    replays it passes zenv_device_ptr() of this field back, actions_on_device = 1) */
    ZENV_F_COUNT = 36
};

/* scripted on-device action sources (the build's own; used by bench/tests) */
enum {
    ZENV_POLICY_UNIFORM = 0,
    ZENV_POLICY_GREEDY = 1,
    ZENV_POLICY_MLP_MEAN = 2,   /* a = mu of the loaded actor network (zenv_mlp_load) */
    ZENV_POLICY_MLP_SAMPLE = 3  /* a ~ Normal(mu, std), the reference's dist.sample() (utils/agent.py:41-44) */
};

/* kernel layouts */
enum {
    ZENV_KERNEL_LANE_PER_ENV = 0,  /* SoA state, one lane per env, LDS-transposed obs tile */
    ZENV_KERNEL_WAVE_PER_ENV = 1   /* one wave64 per env, lane z owns zone z (north-star layout): same results,
                                    * ~10x the step time; kept selectable for comparison */
};

/* Replaces the config dicts of envs/__init__.py:7-50 merged into Engine.DEFAULT
 * (ZoneEnvBase.py:42-53) and the model constants of xmls/point.xml. */
typedef struct zenv_config {
    int32_t task;              /* ZENV_TASK_* */
    int32_t num_zones;         /* 'num_cities' */
    int32_t num_steps;         /* 'num_steps' */
    int32_t max_cd;            /* colour_match_env.py:16 */
    int32_t frameskip;         /* Engine frameskip_binom_n (p = 1.0) */
    int32_t kernel;            /* ZENV_KERNEL_* */
    double zones_size;         /* ZoneEnvBase.py:51 */
    double zones_keepout;      /* ZoneEnvBase.py:50 */
    double robot_keepout;
    double extent;             /* ZoneEnvBase.py:41 */
    double placements_margin;
    double time_saved_reward;  /* TSP_env.py:15 */
    double beta_a, beta_b;     /* TTSP_env.py:13 */
    double timestep;           /* point.xml <option timestep> */
    double mass, com_x, inertia_zz;
    double damping[3];
    double gear, forcerange, vel_kv;
    double reward_exception;   /* Engine.DEFAULT 'reward_exception' (-10.0): the reward of a step MuJoCo could not simulate */
    /* Fixed placements and a pre-coloured start: config_zone_fixed_1/_2 of envs/__init__.py:52-81 for TSPHardEnv
     * (TSP_hard_env.py:11-29, ids PointTSP-v4 / -v5).  [not vendored] Engine.placements_dict_from_object gives object i
     * with a fixed location (x, y) the placement box (x-k, y-k, x+k, y+k), k = keepout + 1e-9, which draw_placement
     * shrinks by the keepout again: the object lands within 1e-9 of (x, y) and still consumes two uniform draws. */
    int32_t n_zones_locations; /* 'zones_locations': the first n zones are fixed, the rest sampled as usual */
    int32_t n_robot_locations; /* 'robot_locations': 0 or 1 entry */
    int32_t robot_rot_fixed;   /* != 0: 'robot_rot' is given (no random_rot() draw for the robot) */
    uint32_t visited0;         /* 'zones_colours': bit z set = zone z starts visited (Yellow); TSP / TimedTSP only */
    double robot_rot;
    double robot_location[2];
    double zones_locations[ZENV_MAX_ZONES][2];
} zenv_config;

typedef struct zenv zenv_t;

/* ---- configuration / introspection (host only; usable without a GPU) ---- */
const char *zenv_last_error(void);
const char *zenv_version(void);
/* Benchmark integrity: the extra compiler switches this library was built with ("" = the shipped build; anything else
 * is a diagnostic variant, e.g. "-DZENV_EXP=1" compiles the row flush out) and the steps one persistent launch really
 * covers (ZENV_ROLLOUT_CHUNK unless the diagnostic environment variable ZENV_ROLLOUT_CHUNK_EXP shortened it).  bench.py
 * prints both and refuses to time a variant unless asked to. */
int zenv_device_count(void);                  /* usable HIP devices (0 without a GPU); does not create a context */
const char *zenv_build_flags(void);
int zenv_rollout_chunk(void);
/* env_id: "PointTSP-v0", "PointTSP-v1", "PointTSP-v4", "PointTSP-v5" (TSPHardEnv), "PointTTSP-v0", "PointTTSP-v1",
 * "ColourMatch-v0" (envs/__init__.py:88-141); unknown id -> ZENV_E_ARG (make_env.py:18 RuntimeError). */
int zenv_config_for_id(const char *env_id, zenv_config *out);
int zenv_default_config(int task, int num_zones, zenv_config *out);
int zenv_zone_feat(const zenv_config *cfg);   /* F: 6 (TSP) or 7 */
int zenv_config_size(void);                   /* sizeof(zenv_config): lets a binding check its mirror */

/* Host layout sampler = Engine.reset()'s random half for env.seed(seed); reset():
 * aux from RandomState(seed) (TTSP_env.py:19-21 tmax / colour_match_env.py:57-68 colours),
 * layout + robot_rot from RandomState(seed+1).  robot_xyrot[3], zone_xy[Z*2], aux[Z]. */
int zenv_sample_layout(const zenv_config *cfg, int64_t seed, double *robot_xyrot,
                       double *zone_xy, int32_t *aux, int32_t *restarts);
/* FixedSeedsWrapper.reset (wrappers.py:20-23): the seed sequence drawn by
 * np.random.default_rng(rng_seed).integers(min_seed, max_seed+1) -- PCG64 + SeedSequence. */
int zenv_fixed_seed_sequence(uint64_t rng_seed, int64_t min_seed, int64_t max_seed,
                             int count, int64_t *out);

/* ---- lifecycle ---- */
/* n_env: 1 .. the largest batch whose zone_obs [N][Z][F] stays below 2^29 floats (3.5 M envs at Z = 25; measured at 1 M:
 * the same 11.9 G env-steps/s); ZENV_E_ARG beyond it -- use several handles. */
int zenv_create(const zenv_config *cfg, int n_env, int device, zenv_t **out);
int zenv_destroy(zenv_t *h);
int zenv_num_envs(const zenv_t *h);
int zenv_get_config(const zenv_t *h, zenv_config *out);

/* ---- layout bank (pre-sampled episodes in HBM) ---- */
/* Sample layouts for env seeds seed_first .. seed_first+count-1 on n_threads host threads
 * and upload them.  Replaces the per-episode XML rebuild of Engine.reset(). */
int zenv_bank_build(zenv_t *h, int64_t seed_first, int count, int n_threads);
/* Same for an arbitrary list of env seeds (slot j <-> seeds[j]). */
int zenv_bank_build_seeds(zenv_t *h, const int64_t *seeds, int count, int n_threads);
/* Upload caller-provided layouts: robot_xyrot [S,3], zone_xy [S,Z,2], aux [S,Z] (may be NULL
 * for TSP), seeds [S]. */
int zenv_bank_set(zenv_t *h, const double *robot_xyrot, const double *zone_xy,
                  const int32_t *aux, const int64_t *seeds, int count);
int zenv_bank_size(const zenv_t *h);
/* Refill bank slots in place: slot slots[j] <- the layout of env seed seeds[j] (sampled here, on n_threads host
 * threads).  Stream-ordered: it lands behind every step already enqueued and before every later one.  With
 * zenv_schedule_ring this is Engine.reset's unbounded seed stream (_seed += 1 at every reset, [not vendored]
 * Engine.reset; make_env.py:20-35 make_test_env never runs out of maps): once env i has taken episode k from its ring,
 * the host puts episode k + depth into the slot that held it. */
int zenv_bank_update(zenv_t *h, const int32_t *slots, const int64_t *seeds, int count, int n_threads);

/* ---- episode schedule: which bank slot env i uses for its k-th episode ---- */
/* slot = (first[i] + k*stride) mod S; first == NULL -> i mod S. */
int zenv_schedule_sequential(zenv_t *h, const int32_t *first, int32_t stride);
/* Ring: env i owns the slots first[i] .. first[i] + depth - 1 and takes slot first[i] + (k mod depth) for its k-th
 * episode; the host keeps the ring ahead of the env with zenv_bank_update. */
int zenv_schedule_ring(zenv_t *h, const int32_t *first, int32_t depth);
/* FixedSeedsWrapper semantics on device: env i draws seeds from its own PCG64 stream
 * default_rng(rng_seeds[i]).integers(min_seed, max_seed+1); the bank must hold
 * min_seed..max_seed in order (make_env.py:3-18,37-51). */
int zenv_schedule_fixed_seeds(zenv_t *h, const uint64_t *rng_seeds, int64_t min_seed,
                              int64_t max_seed);

/* ---- the hot path ---- */
/* ParallelEnv.reset (penv.py:46-50) / masked re-reset: mask uint8[N] host, NULL = all. */
int zenv_reset(zenv_t *h, const uint8_t *mask);
/* ParallelEnv.step / step_no_reset (penv.py:52-66): actions float32 [N,2]
 * (NULL = internal action buffer written by zenv_policy); auto_reset != 0 resets finished
 * envs in the same launch and returns the new episode's first observation with the terminal
 * reward/done/goal_met (penv.py:8-11).  With auto_reset == 0 a finished env is a masked
 * no-op: zero obs, reward 0, done 1 (WaitWrapper, wrappers.py:34-45); the first auto_reset != 0 step after that
 * brings it back -- reward 0, done 1 and the next episode's first observation -- as the worker's
 * `if done: obs = env.reset()` does after WaitWrapper's no-op (the fixed-length-skill loop,
 * torch_ac/algos/hier_base.py:179-183: skill_len - 1 step_no_reset calls, then one step).
 * Exception path ([not vendored] Engine.step: `except MujocoException: done = True; reward = reward_exception;
 * info['exception'] = True`): mujoco-py raises it when MuJoCo warns that qacc / qpos / qvel hold a NaN, an Inf or a
 * value beyond 1e10 (mj_checkAcc -> mjWARN_BADQACC, after which MuJoCo has reset the data to qpos0, qvel = 0).  With
 * the actuator forces clamped and a validated config the state stays finite, so the one trigger is a NaN action
 * (np.clip keeps a NaN): that step ends the episode with reward_exception, no goal test, joint state zeroed,
 * ZENV_F_EXCEPTION = 1.  The zone visit of that step's first set_mocaps() still counts (it ran before sim.step()). */
int zenv_step(zenv_t *h, const float *actions, int actions_on_device, int auto_reset);
/* An action chunk: n_steps steps of caller-supplied actions [n_steps][N][2] (host, or device memory of the handle's
 * device) -- the same results as n_steps zenv_step() calls, as ONE launch of the persistent kernel per
 * ZENV_ROLLOUT_CHUNK steps (env state in registers, a_{t+1} prefetched under step t; zone counts 5, 6, 10, 15, 20, 25 in
 * the lane layout -- other handles run the single-step launches).  Asynchronous on the handle's stream like zenv_step.
 *   ZENV_CHUNK_NO_RESET     n_steps x step_no_reset: a finished env idles as WaitWrapper's no-op (wrappers.py:34-45)
 *   ZENV_CHUNK_RESET_EVERY  n_steps x step (penv.py:52-59)
 *   ZENV_CHUNK_RESET_LAST   n_steps - 1 x step_no_reset, then one step: the fixed-length-skill loop of
 *                           main/src/torch_ac/algos/_hier_policy_opt.py:68-71 / hier_base.py:179-183 -- an env that
 *                           finishes inside the chunk waits (zero obs, reward 0, done) and comes back at the boundary
 * A NaN action takes Engine.step's exception branch as in zenv_step.  Afterwards obs / zone_obs / reward / done /
 * goal_met hold the LAST step's results; every step's reward and done flag is in ZENV_F_CHUNK_REWARD / _DONE
 * ([n_steps][N], valid until the next zenv_step_many).  On a ring schedule RESET_EVERY is limited to `depth` steps
 * per call (ZENV_E_STATE beyond: the host refills the ring between calls). */
enum { ZENV_CHUNK_NO_RESET = 0, ZENV_CHUNK_RESET_EVERY = 1, ZENV_CHUNK_RESET_LAST = 2 };
int zenv_step_many(zenv_t *h, const float *actions, int actions_on_device, int n_steps, int reset_mode);
/* Scripted action source -> internal action buffer (or dst_device if non-NULL). */
int zenv_policy(zenv_t *h, int policy, uint64_t policy_seed, uint64_t env_index0,
                float *dst_device);
/* K closed-loop steps a_t = policy(obs_t, t); step(a_t) on the handle's stream, HIP-event
 * timed.  Default (flags 0): the persistent rollout kernel -- one launch advances every env by
 * up to ZENV_ROLLOUT_CHUNK steps with the env state in registers, publishing obs / zone_obs / reward / done /
 * goal_met to memory on every step (zone counts 5, 6, 10, 15, 20, 25; other counts use the next mode).
 * ZENV_ROLLOUT_PER_STEP: one step-kernel launch per step, the kernel of step t also emitting
 * a_{t+1} (fused action source).  ZENV_ROLLOUT_UNFUSED: per-step launches with the stand-alone
 * policy kernel before each.  Results are identical in all three.  ms_total: whole loop (events
 * on the stream); ms_step_kernel_avg (may be NULL): kernel time per step -- persistent: begin/end
 * events of every launch, summed, / steps; otherwise the mean over every event_stride-th
 * step-kernel dispatch. */
/* zenv_rollout() is SYNCHRONOUS by default: it returns when its last launch has finished (it reports their times).
 * ZENV_ROLLOUT_ASYNC: enqueue and return at once (ms_* come back as -1); the host overlaps its own work -- refilling
 * bank slots (zenv_bank_update), a policy update -- and collects with zenv_query() (1 = the stream is idle, 0 = still
 * running, < 0 error) or zenv_sync(). */
#define ZENV_ROLLOUT_UNFUSED 1
#define ZENV_ROLLOUT_PER_STEP 2
#define ZENV_ROLLOUT_ASYNC 4
#define ZENV_ROLLOUT_CHUNK 256   /* most steps one persistent launch covers */
int zenv_rollout(zenv_t *h, int steps, int policy, uint64_t policy_seed, uint64_t env_index0,
                 int auto_reset, int flags, int event_stride, float *ms_total,
                 float *ms_step_kernel_avg);
/* How many envs ONE persistent launch covers (default 65 536 = 1 024 tiles: the chip's 1 024 SIMDs each hold one env
 * wave and one stream wave; a larger batch is stepped slice by slice, every slice ZENV_ROLLOUT_CHUNK steps at a time).
 * A launch over more workgroups than are resident at once loses that placement -- one launch over 131 072 envs takes
 * 6.9 us per 65 536 env-steps against 5.3 for two launches over 65 536 each -- and a slice's per-step output (42 MB at
 * Z = 25) stays in the Infinity Cache while it is rewritten.  0: the whole batch in one launch.  Results do not
 * depend on it. */
int zenv_set_rollout_slice(zenv_t *h, int envs_per_launch);

/* ---- goal-conditioned variant: TSPNextCityEnv, main/envs/zone_envs/
 * TSP_next_city_env.py:41-109, and TimedTSPNextCityEnv, zone-goals/envs/TTSP_next_city_env.py:40-51, as the
 * vector calls of zone-goals/src/torch_ac/torch_utils/penv.py:76-99 (set_goal / needs_goal / available_goals).
 * ColourMatchNextCityEnv (zone-goals/envs/colour_match_next_city_env.py) likewise: any zone may be a goal, and
 * cycling a zone other than the goal costs 1.  After zenv_goal_enable() every env needs a goal (ZENV_F_NEED_GOAL = 1);
 * zenv_set_goals() takes int32 goals[N] from the host (-1 = leave that env alone) and fails with ZENV_E_ARG
 * when a goal zone is out of range or (TSP / TimedTSP) already visited (set_goal's assert, :86); every zenv_step() then also produces
 * shaped_reward = last_dist_to_goal - dist_to_goal (0 in the step that reaches the goal), need_next_goal
 * (goal reached, or episode over) and the available-goals mask.  zenv_rollout() is refused on such a handle;
 * the goal arrays are not part of zenv_get_state(). */
int zenv_goal_enable(zenv_t *h);
int zenv_set_goals(zenv_t *h, const int32_t *goals);
/* ColourMatchSolverEnv.solver_get_next_goal (zone-goals/envs/colour_match_solver_env.py:57-97, id ColourMatch-v2) for
 * every env of a goal-conditioned ColourMatch handle: the nearest zone that a cheapest recolouring plan has to cycle
 * (ties: lowest index) into goals[N] (host).  Feed the entries of the envs that need a goal to zenv_set_goals(). */
int zenv_solver_goals(zenv_t *h, int32_t *goals);

/* ---- solver-ordered variant: TSPOrderEnv, main/envs/TSP_order_env.py:13-113 (PointTSP-v2) ----
 * TSP handles only; call before building the bank.  The route of an episode is the bank's aux column (rank of
 * every zone in the visiting order): zenv_bank_build* fill it with zenv_route_ranks()'s tour (the solver's problem,
 * solved without OR-tools, which the reference calls at :49-50 and which is not available), zenv_bank_set takes the caller's.
 * Every zenv_step() then also yields info['shaped_reward'] (ZENV_F_SHAPED_REWARD, :63-72) and the order feature
 * of every zone (ZENV_F_ORDER_VAL, :37-47) -- the reference's (Z,7) row is [zone_obs row (6), order value].
 * Exclusive with zenv_goal_enable; zenv_rollout() is refused on such a handle. */
int zenv_order_enable(zenv_t *h);
/* First observation of an episode.  Default (flags 0) = the reference's: TSPOrderEnv.reset() builds init_obs BEFORE
 * generate_route() (TSP_order_env.py:108-113), so the observation returned by reset() -- and by ParallelEnv's auto-reset,
 * penv.py:8-11 -- carries the order feature of the route the env object was left with: all zeros before the first
 * episode and after a finished one, the unvisited rest of the previous route (indexed by zone number, on the NEW map's
 * rows) after a time-limit end.  From the first step on the feature follows the new route; shaped_reward /
 * last_dist_to_goal use the new route from the start (:112).  ZENV_ORDER_FRESH_FIRST_OBS: the first observation
 * already shows the new episode's route (not what the reference does). */
#define ZENV_ORDER_FRESH_FIRST_OBS 1
int zenv_order_configure(zenv_t *h, int flags);
/* The built-in route of a layout (host only): rank[z] = position of zone z in the tour.  The problem is the one
 * TSP_Solver.get_optim_route (main/src/utils/TSP_Solver.py:24-62) hands to OR-tools -- closed tour from the robot, arc
 * cost int64(10 x distance), first solution PATH_CHEAPEST_ARC, greedy-descent local search (relocate, exchange, 2-opt,
 * or-opt) to a local optimum -- solved without the library: the same kind of tour, not its bit-exact route. */
int zenv_route_ranks(const double *robot_xy, const double *zone_xy, int num_zones, int32_t *rank);

/* ---- the reference's actor network on the device ----
 * ZoneEnvModel (main/src/env_model.py:48-79) + the actor of ACModel (flat_model.py:24-37,
 * policy_network.py:12-53, Box action space), evaluated on bf16 MFMA with float32 accumulation from
 * the handle's own obs / zone_obs buffers.  Pointers are host float32 tensors in the state_dict's
 * layout (row-major [out][in]); F = zenv_zone_feat(cfg).  h_dim <= 191 (the reference uses 185). */
enum {
    ZENV_MLP_BF16 = 0,  /* bf16 MFMA, float32 accumulation: ~20x faster, mu / std within 4e-2 of the reference's float32 */
    ZENV_MLP_F32 = 1,   /* float32 throughout (f32 MFMA / FMA): mu / std / value within 1e-5 of the reference's torch float32 --
                         * the mode in which evaluate() with a checkpoint reproduces the reference's arithmetic */
    ZENV_MLP_BF16X3 = 2, /* the two zone layers -- 96 % of the arithmetic -- as three bf16 products per k-step on hi / lo split
                         * operands (16 significant bits, float32's range), float32 accumulation, the per-env head on the
                         * float32 matrix instruction: within 2e-5 of torch float32 (measured: up to 1.1e-5) at a third of
                         * ZENV_MLP_F32's time */
    ZENV_MLP_F16X3 = 3, /* the same with float16 halves (22 significant bits: within 3e-6 of torch float32, float32's own
                         * rounding noise) on every layer -- the fastest of the float32-grade modes (0.30 of ZENV_MLP_F32).  float16's range applies to every
                         * operand: weights of 32 768 or more are refused by zenv_mlp_load (ZENV_E_RANGE); an input or
                         * activation that reaches 65 520 is caught on the device and reported as ZENV_E_RANGE by the next
                         * call that waits for it (zenv_get*, zenv_sync, zenv_rollout, zenv_step_results).
                         * Batches too small for the matrix kernel (< 2 048 envs) run the float32 vector kernel in both. */
    ZENV_MLP_F16 = 4    /* ZENV_MLP_BF16's two kernels with float16 operands (one product per k-step, float32 accumulation): 5 %
                         * slower than bf16 (the wider multipliers draw more power), 11 significant bits instead of 8 -- mu / std within 1e-3 of the reference's float32
                         * (bf16: within 4e-2 by contract, ~4e-3 measured).  float16's range is guaranteed, not assumed:
                         * zenv_mlp_load refuses weights of 65 504 or more and weights whose zone-layer activations could
                         * leave the range for observations up to 64 (a bound over the rows' absolute sums: ZENV_E_RANGE, use
                         * ZENV_MLP_BF16 or a split mode); the head kernel watches its own activations, and the zone kernel
                         * the observations' magnitude, at run time (ZENV_E_RANGE from the next call that waits). */
};
typedef struct zenv_mlp_weights {
    int32_t h_dim;
    int32_t precision;              /* ZENV_MLP_* */
    const float *zone_w1, *zone_b1; /* env_model.zone_net_.0  [h, 8+F], [h]   input = [obs, zone row] */
    const float *zone_w2, *zone_b2; /* env_model.zone_net_.2  [h, h],   [h] */
    const float *zone_w3, *zone_b3; /* env_model.zone_net_.4  [h, h],   [h] */
    const float *comb_w, *comb_b;   /* env_model.combine_net_ [h, 8+h], [h]   input = [obs, zone_emb] */
    const float *enc_w, *enc_b;     /* actor.enc_.0.0         [h, h],   [h] */
    const float *mu_w, *mu_b;       /* actor.mu_              [2, h],   [2] */
    const float *std_w, *std_b;     /* actor.std_             [2, h],   [2] */
    /* critic of flat_model.ACModel (:43-47, non-distributional): all four NULL = no value head */
    const float *critic_w1, *critic_b1; /* critic.0           [h, h],   [h] */
    const float *critic_w2, *critic_b2; /* critic.2           [1, h],   [1]   (distributional: critic_mu) */
    /* distributional critic (ACModel(distributional_value=True), flat_model.py:35-41,57-60): both non-NULL ->
     * value = critic_mu(relu(critic.0(x))) in ZENV_F_POLICY_VALUE and softplus_{beta=0.3}(critic_sigma(.)) + 1e-3 in
     * ZENV_F_POLICY_VALUE_SIGMA; collect_experiences uses the mean only (base.py:140-141,193-194) */
    const float *critic_sigma_w, *critic_sigma_b; /*         [1, h],   [1] */
} zenv_mlp_weights;
int zenv_mlp_load(zenv_t *h, const zenv_mlp_weights *w);
/* mu = 2 (sigmoid(mu_(x)) - 0.5), std = sigmoid(std_(x)) + 1e-3 of the current observations into
 * ZENV_F_POLICY_MU / ZENV_F_POLICY_STD (and value = critic(x) into ZENV_F_POLICY_VALUE).  zenv_policy() / zenv_rollout() with ZENV_POLICY_MLP_* call it
 * and turn it into actions (rollouts then run one launch sequence per step). */
int zenv_mlp_forward(zenv_t *h);

/* ---- one PPO rollout on the device: BaseAlgo.collect_experiences, main/src/torch_ac/algos/base.py:131-227 ----
 * T times: (dist, value) = acmodel(obs) [zenv_mlp_forward]; action = dist.sample(); record obs, action, value,
 * log_prob, mask; step the envs (auto-reset); record the reward.  Then next_value = value(obs_T) and the GAE
 * recursion.  Needs actor AND critic weights (zenv_mlp_load).  The buffers (ZENV_F_EXP_*) stay valid until the
 * next call with a different T or zenv_destroy; self.mask is carried from call to call like the reference's. */
int zenv_collect(zenv_t *h, int frames_per_proc, uint64_t policy_seed, uint64_t env_index0, float discount,
                 float gae_lambda);

/* ---- results ---- */
int zenv_get(zenv_t *h, int field, void *dst, int dst_on_device);
/* The rows of envs [first_env, first_env + count) of an env-major field (obs, zone_obs, reward, counters ...; not the
 * time-major ZENV_F_EXP_* buffers) into host memory: what a caller that looks at a few envs of a multi-GB batch uses. */
int zenv_get_rows(zenv_t *h, int field, int first_env, int count, void *dst);
int zenv_device_ptr(zenv_t *h, int field, void **ptr);  /* zero-copy for GPU consumers */
int64_t zenv_field_bytes(const zenv_t *h, int field);
int zenv_sync(zenv_t *h);
int zenv_query(zenv_t *h);   /* non-blocking: 1 = everything enqueued on the handle's stream has finished, 0 = not yet */
/* Host-policy surface (a CPU-resident policy: actions up, observations down, every step): page-locked host
 * memory for the caller's buffers, so that zenv_step()'s action upload and zenv_get()'s downloads run as DMA
 * at PCIe rate instead of through a pageable bounce buffer.  zenv_get_many() enqueues several downloads and
 * synchronises once.  Free with zenv_host_free (any time before process exit). */
void *zenv_host_alloc(int64_t bytes);
int zenv_host_free(void *ptr);
int zenv_get_many(zenv_t *h, int n_fields, const int *fields, void *const *dst);
/* The per-step results a host policy reads back -- what one `worker` of the reference pickles into its Pipe after
 * env.step() (main/src/torch_ac/torch_utils/penv.py:8-12, 52-59) -- live in ONE device allocation, 256-byte aligned
 * pieces in ZENV_RESULT_* order, so that a step of a small batch (the reference trains with 16 envs,
 * scripts/train_ppo.py:29-30) costs one upload, one launch, one download and one synchronisation:
 * zenv_results_layout() returns the slab's size and the piece offsets (offsets may be NULL);
 * zenv_step_results() = zenv_step(actions on the host) + download of the whole slab into host_slab (page-locked
 * memory from zenv_host_alloc for DMA) + synchronise; actions == NULL only downloads (after zenv_reset). */
enum {
    ZENV_RESULT_OBS = 0,        /* float32 [N][8]            */
    ZENV_RESULT_REWARD = 1,     /* float32 [N]               */
    ZENV_RESULT_DONE = 2,       /* uint8   [N]               */
    ZENV_RESULT_GOAL_MET = 3,   /* uint8   [N]               */
    ZENV_RESULT_EXCEPTION = 4,  /* uint8   [N]               */
    ZENV_RESULT_ZONE_OBS = 5,   /* float32 [N][Z][F]         */
    ZENV_N_RESULTS = 6
};
int64_t zenv_results_layout(const zenv_t *h, int64_t *offsets /* [ZENV_N_RESULTS] or NULL */);
int zenv_step_results(zenv_t *h, const float *actions, int auto_reset, void *host_slab);

/* Host-resident I/O for small batches driven from the host (the reference's own shape: 16 worker envs and a policy on
 * the host, train_ppo.py:29-30 / penv.py:63-77).  enable = 1 moves the results slab (zenv_results_layout) and the action
 * buffer into page-locked host memory that the kernels write and read THEMSELVES over the bus: a step is then one kernel
 * launch and one wait -- no upload, no download (zenv_step_results: two copy enqueues that cost more than the kernel at
 * this size).  *results / *actions receive the two buffers (NULL when switched off); write the actions, call
 * zenv_step_host, read the results in place.  Every other entry point keeps working (zenv_get*, zenv_step_results,
 * snapshots), device-side readers of the observations (the network kernels) then read them over the bus: meant for
 * batches of a few hundred envs at most.  zenv_collect is refused while it is on.  enable = 0 moves everything back. */
int zenv_host_io(zenv_t *h, int enable, void **results, float **actions);
/* zenv_step with the actions already in the zenv_host_io buffer; returns when the results are in theirs. */
int zenv_step_host(zenv_t *h, int auto_reset);
/* Enqueue everything from now on onto the caller's HIP stream (hipStream_t passed as void*; NULL =
 * back to the handle's own stream; the null stream is named by hipStreamLegacy).  The handle first
 * drains the stream it was using.  This is how
 * a device-resident policy (base.py:139-145 without the .cpu().numpy() round trip) shares one
 * stream with the env: step, read the obs buffers of zenv_device_ptr(), compute actions, step. */
int zenv_set_stream(zenv_t *h, void *hip_stream);
int64_t zenv_step_count(const zenv_t *h);   /* batched steps executed so far */

/* ---- multi-GPU: env shards and the job's one collective ----
 * The path shards trivially: one process per GPU, rank r owns global envs [r*N, (r+1)*N), nothing is exchanged on the
 * step path.  After a rollout the per-env episodic figures are all-gathered over RCCL (xGMI) -- what replaces the
 * reference's per-env Pipe star, main/src/torch_ac/torch_utils/penv.py:26-40, at N > 1.  No PyTorch involved: librccl is
 * dlopen()ed by the first of these calls (a copy already in the process wins; ZENV_RCCL_PATH names another).
 *   zenv_comm_unique_id   rank 0: ncclGetUniqueId into id[ZENV_COMM_ID_BYTES]; the host hands it to every rank
 *                         (any side channel: a file, the launcher's store)
 *   zenv_comm_init        ncclCommInitRank on the handle's device; collective: every rank calls it
 *   zenv_allgather        field = one 4- or 8-byte figure per env (ZENV_F_LAST_RETURN, _EP_RETURN: float64 narrowed to
 *                         float32; ZENV_F_REWARD float32; ZENV_F_EPISODES, _LAST_LEN, _VISIT_COUNT, _EP_LEN int32) ->
 *                         dst [world * N] 4-byte elements ordered by global env index, on the handle's stream,
 *                         synchronised on return
 *   zenv_comm_barrier     the handle's stream drained on every rank (one-element all-reduce + synchronise)
 *   zenv_comm_allreduce_max  *value = max over ranks (the bench's max-over-ranks step time) */
#define ZENV_COMM_ID_BYTES 128
int zenv_comm_unique_id(void *id_out);
int zenv_comm_init(zenv_t *h, int rank, int world, const void *unique_id);
int zenv_comm_destroy(zenv_t *h);
int zenv_comm_info(const zenv_t *h, int *rank, int *world, const char **library);
int zenv_allgather(zenv_t *h, int field, void *dst, int dst_on_device);
int zenv_comm_barrier(zenv_t *h);
int zenv_comm_allreduce_max(zenv_t *h, double *value);

/* Measurement utility (bench.py `store_stream_ceiling`; not part of the env path): n_tiles waves each rewrite their
 * own contiguous tile_bytes (a multiple of 16, >= 1024) `steps` times with 1 KiB dwordx4 bursts under cache_policy
 * (0 plain, 2 non-temporal, 16 sc1 = write-through) -- the row stream of the step kernels with the env taken away.
 * us_per_step = best of `reps` timed launches (dispatch begin/end events) / steps, after one untimed launch. */
int zenv_probe_store_stream(int device, int64_t n_tiles, int tile_bytes, int steps, int cache_policy, int reps,
                            float *us_per_step);

/* ---- state snapshot (tests / checkpointing; the reference never checkpoints env state) ---- */
int64_t zenv_state_bytes(const zenv_t *h);
int zenv_get_state(zenv_t *h, void *dst, int64_t bytes);
int zenv_set_state(zenv_t *h, const void *src, int64_t bytes);
/* Per-env dynamic state for parity tests: qpos[N,3], qvel[N,3] float64; zone_state int32 [N,Z]
 * (visited 0/1, or colour 0..2); cooldown int32 [N,Z]; steps int32 [N]. Any may be NULL. */
int zenv_debug_state(zenv_t *h, double *qpos, double *qvel, int32_t *zone_state,
                     int32_t *cooldown, int32_t *steps);

#ifdef __cplusplus
}
#endif
#endif /* ZENV_H */

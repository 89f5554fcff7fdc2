/*
 * zenv_oracle.h -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * Plain-C float64 restatement of the PointTSP / TimedTSP / ColourMatch
 * env.step()/reset() path of andrewli77/combinatorial-rl-tasks
 * (main/envs/{TSP,TTSP,colour_match}_env.py, zone_envs/ZoneEnvBase.py) and of
 * the un-vendored Safety-Gym Engine + MuJoCo "point.xml" dynamics beneath it.
 *
 * PARITY UNPINNED: the reference ships no tests / golden vectors and its
 * arithmetic lives in third-party packages (safety-gym, mujoco-py 2.0.2.9 /
 * MuJoCo 2.0, gym) that are absent from /root/reference and from this image.
 * Only the RNG/layout half is pinned (against numpy's legacy RandomState,
 * which is the reference's actual dependency); the dynamics half is pinned
 * by analytic known-answers only.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The shipped HIP path never links or calls it.
 */
#ifndef ZENV_ORACLE_H
#define ZENV_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_Z 32

enum { ORC_TASK_TSP = 0, ORC_TASK_TIMED = 1, ORC_TASK_COLOUR = 2 };
enum { ORC_POLICY_UNIFORM = 0, ORC_POLICY_GREEDY = 1 };

typedef struct orc_config {
    int32_t task;              /* ORC_TASK_* */
    int32_t num_zones;         /* num_cities, envs/__init__.py:9 */
    int32_t num_steps;         /* envs/__init__.py:13 */
    int32_t max_cd;            /* colour_match_env.py:16 */
    int32_t frameskip;         /* Engine frameskip_binom_n (p = 1) */
    int32_t pad0;
    double zones_size;         /* ZoneEnvBase.py:51 */
    double zones_keepout;      /* ZoneEnvBase.py:50 */
    double robot_keepout;      /* Engine DEFAULT */
    double extent;             /* ZoneEnvBase.py:41 */
    double placements_margin;  /* Engine DEFAULT 0.0 */
    double time_saved_reward;  /* TSP_env.py:15 */
    double beta_a, beta_b;     /* TTSP_env.py:13 */
    /* Point robot (safety_gym/xmls/point.xml, [UPSTREAM-RECALL]) */
    double timestep;
    double mass;               /* total body mass */
    double com_x;              /* COM offset along body x */
    double inertia_zz;         /* about the hinge axis (body origin) */
    double damping[3];         /* slide x, slide y, hinge z */
    double gear;               /* both actuators */
    double forcerange;         /* |force| clamp of both actuators */
    double vel_kv;             /* velocity actuator gain */
    double reward_exception;   /* [UPSTREAM] Engine.DEFAULT 'reward_exception': -10.0 */
    /* TSPHardEnv (TSP_hard_env.py:11-29) configs config_zone_fixed_1/_2, envs/__init__.py:52-81 */
    int32_t n_zones_locations; /* 'zones_locations': the first n zones have fixed locations */
    int32_t n_robot_locations; /* 'robot_locations': 0 or 1 */
    int32_t robot_rot_fixed;   /* 'robot_rot' given */
    uint32_t visited0;         /* 'zones_colours': bit z = zone z starts as zone.Yellow (visited) */
    double robot_rot;
    double robot_location[2];
    double zones_locations[ORC_MAX_Z][2];
} orc_config;

typedef struct orc_env {
    orc_config cfg;
    /* episode-static */
    int64_t seed;              /* value given to env.seed() */
    double x0, y0, rot;        /* robot placement */
    double bq0, bq3;           /* body quaternion (w, z) */
    double zone_xy[ORC_MAX_Z][2];
    int32_t tmax[ORC_MAX_Z];
    /* dynamic */
    double qpos[3], qvel[3];
    double xpos[2];            /* world position as of the last forward */
    double xvelp[2], xvelr;    /* world velocities as of the last forward */
    double xquat0, xquat3;
    int32_t visited[ORC_MAX_Z];   /* TSP/Timed */
    int32_t colour[ORC_MAX_Z];    /* 0 Blue 1 Green 2 Red */
    int32_t cooldown[ORC_MAX_Z];
    int32_t goal_dist;
    int32_t steps;
    int32_t done;
    int32_t layout_restarts;   /* diagnostics */
    /* goal-conditioned variant (TSP_next_city_env.py) */
    int32_t goal_zone;         /* -1 = None */
    int32_t last_visit;        /* zone visited by the last step, -1 = none */
    double last_dist;          /* last_dist_to_goal */
    /* solver-ordered variant (TSP_order_env.py): route_len zones still to visit, in order */
    int32_t route[ORC_MAX_Z];
    int32_t route_len;
    int32_t exception;         /* info['exception'] of the last step ([UPSTREAM] Engine.step) */
    /* self.route as the most recent obs() call saw it.  It differs from `route` only between reset() and the first
     * step: TSPOrderEnv.reset builds its observation BEFORE generate_route() (TSP_order_env.py:108-113) */
    int32_t obs_route[ORC_MAX_Z];
    int32_t obs_route_len;
    int32_t pad_order;
} orc_env;

/* ---- numpy-legacy RandomState restatement (exposed for pinning tests) ---- */
typedef struct orc_rs {
    uint32_t key[624];
    int pos;
    int has_gauss;
    double gauss;
} orc_rs;
void orc_rs_seed(orc_rs *rs, uint32_t seed);
uint32_t orc_rs_u32(orc_rs *rs);
double orc_rs_double(orc_rs *rs);
double orc_rs_uniform(orc_rs *rs, double lo, double hi);
int64_t orc_rs_choice(orc_rs *rs, int64_t n);
double orc_rs_beta(orc_rs *rs, double a, double b);

/* deterministic sin/cos used by the dynamics (exposed for tests) */
void orc_sincos(double x, double *s, double *c);

/* ---- configuration ---- */
void orc_default_config(int task, int num_zones, orc_config *out);

/* ---- single env ---- */
int orc_reset(orc_env *e, const orc_config *cfg, int64_t seed);
/* step: returns 0, or -1 when the env is done (must be reset first). */
int orc_step(orc_env *e, const float action[2], double *reward, int *done, int *goal_met);
/* obs: obs8[8], zone_obs[Z*F] as float32 (the dtype the consumer casts to, format.py:27-28) */
void orc_obs(const orc_env *e, float *obs8, float *zone_obs);
int orc_zone_feat(const orc_config *cfg);

/* ---- scripted policies (deterministic; same definition as the device K3 kernels) ---- */
void orc_policy(int policy, const orc_config *cfg, const float *obs8, const float *zone_obs,
                uint64_t env_index, uint32_t step_index, uint64_t policy_seed, float action[2]);

/* ---- batch driver (cpu_baseline / parity traces) ----
 * Runs n_env envs for n_steps closed-loop steps with auto-reset; env i starts on
 * seed seeds0[i]; episode k of env i uses seed seeds0[i] + k*seed_stride.
 * Outputs (any may be NULL): per-env sum of rewards, episodes finished, last
 * finished episode return/len, final obs. Returns total env-steps executed. */
int64_t orc_rollout(const orc_config *cfg, int n_env, int n_steps, int policy,
                    const int64_t *seeds0, int64_t seed_stride, uint64_t policy_seed,
                    uint64_t env_index0, int n_threads,
                    double *reward_sum, int32_t *episodes, double *last_return,
                    int32_t *last_len, float *final_obs8, float *final_zone_obs);

/* ---- goal-conditioned variant: TSPNextCityEnv main/envs/zone_envs/TSP_next_city_env.py:41-109 and
 * TimedTSPNextCityEnv zone-goals/envs/TTSP_next_city_env.py:40-51 (TSP / TimedTSP tasks) ----
 * orc_set_goal: -1 when the zone is out of range or visited (set_goal's assert).  orc_step_goal: -2
 * without a goal (step's assert); otherwise orc_step plus info['shaped_reward'] / info['need_next_goal']. */
int orc_set_goal(orc_env *e, int goal);
/* ColourMatchSolverEnv.solver_get_next_goal, zone-goals/envs/colour_match_solver_env.py:57-97 (ColourMatch task) */
int orc_solver_next_goal(const orc_env *e);
int orc_step_goal(orc_env *e, const float action[2], double *reward, int *done, int *goal_met,
                  double *shaped_reward, int *need_next_goal);

/* ---- solver-ordered variant: TSPOrderEnv main/envs/TSP_order_env.py:13-113 (TSP task) ----
 * orc_reset_order: TSPOrderEnv.reset() in the reference's statement order (:108-113) on an env object that LIVES ON
 * across episodes (`e` must be zero-initialised before its first reset: self.route = [] of __init__, :27):
 *   zones = unvisited; init_obs = super().reset()   <- the observation is built here, from self.route AS IT STOOD: the
 *                                                      previous episode's leftover (empty after a finished episode)
 *   generate_route()                                <- rank[z] = position of zone z in the new route (the solver,
 *                                                      OR-tools, is outside the tree: the caller supplies its answer)
 *   last_dist_to_goal = dist_to_goal()
 * fresh_first_obs != 0: the build's opt-out -- the first observation already shows the new route.
 * orc_step_order: orc_step plus info['shaped_reward'] (:63-75).  orc_order_vals: the 7th row feature of the most
 * recent observation, 0.5^index in the route that obs() saw, 0 when not in it (:41-45).  orc_order_route: self.route. */
int orc_reset_order(orc_env *e, const orc_config *cfg, int64_t seed, const int32_t *rank, int fresh_first_obs);
int orc_order_route(const orc_env *e, int32_t *route);   /* returns its length */
int orc_step_order(orc_env *e, const float action[2], double *reward, int *done, int *goal_met, double *shaped_reward);
void orc_order_vals(const orc_env *e, float *vals);

/* Same, with a bank of seed_period maps per env replayed in order: episode k uses seed
 * seeds0[i] + (k % seed_period)*seed_stride (seed_period = 0: no wrap). */
int64_t orc_rollout_wrapped(const orc_config *cfg, int n_env, int n_steps, int policy,
                            const int64_t *seeds0, int64_t seed_stride, int32_t seed_period,
                            uint64_t policy_seed, uint64_t env_index0, int n_threads,
                            double *reward_sum, int32_t *episodes, double *last_return,
                            int32_t *last_len, float *final_obs8, float *final_zone_obs);

#ifdef __cplusplus
}
#endif
#endif

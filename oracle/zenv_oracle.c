/*
 * zenv_oracle.c -- CPU ORACLE (test infrastructure, NOT the product).  PARITY UNPINNED
 * for the dynamics half (see zenv_oracle.h).  Build: gcc -O2 -ffp-contract=off.
 *
 * Each function cites the reference file:line (relative to /root/reference/) whose
 * behaviour it restates; [UPSTREAM] marks behaviour of safety-gym / mujoco-py /
 * MuJoCo 2.0 / numpy code that the reference calls but does not vendor.
 */
#include "zenv_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* One correctly rounded multiply-add, spelled out wherever the arithmetic uses one: the GPU kernels
 * issue the same v_fma_f64 in the same places, and -ffp-contract=off keeps the compiler from adding
 * or removing any. */
#define FMA(a, b, c) __builtin_fma((a), (b), (c))

/* =====================================================================================
 * numpy legacy RandomState  (reference call sites: TTSP_env.py:20-21,
 * colour_match_env.py:60-62, [UPSTREAM] Engine.reset: RandomState(self._seed))
 * ===================================================================================== */

void orc_rs_seed(orc_rs *rs, uint32_t seed)
{
    /* numpy mt19937_seed == Knuth init_genrand */
    for (int i = 0; i < 624; i++) {
        rs->key[i] = seed;
        seed = 1812433253u * (seed ^ (seed >> 30)) + (uint32_t)i + 1u;
    }
    rs->pos = 624;
    rs->has_gauss = 0;
    rs->gauss = 0.0;
}

static void rs_refill(orc_rs *rs)
{
    uint32_t *mt = rs->key;
    int kk;
    uint32_t y;
    for (kk = 0; kk < 624 - 397; kk++) {
        y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
        mt[kk] = mt[kk + 397] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    for (; kk < 623; kk++) {
        y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
        mt[kk] = mt[kk + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    y = (mt[623] & 0x80000000u) | (mt[0] & 0x7fffffffu);
    mt[623] = mt[396] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    rs->pos = 0;
}

uint32_t orc_rs_u32(orc_rs *rs)
{
    if (rs->pos == 624) rs_refill(rs);
    uint32_t y = rs->key[rs->pos++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

double orc_rs_double(orc_rs *rs)
{
    /* numpy mt19937_next_double: 53-bit from two draws */
    int32_t a = (int32_t)(orc_rs_u32(rs) >> 5);
    int32_t b = (int32_t)(orc_rs_u32(rs) >> 6);
    return (a * 67108864.0 + b) / 9007199254740992.0;
}

double orc_rs_uniform(orc_rs *rs, double lo, double hi)
{
    /* numpy legacy uniform: low + (high-low)*double */
    double range = hi - lo;
    return lo + range * orc_rs_double(rs);
}

int64_t orc_rs_choice(orc_rs *rs, int64_t n)
{
    /* RandomState.choice(n) -> randint(0, n) -> masked rejection on 32-bit draws */
    uint64_t rng = (uint64_t)(n - 1);
    if (rng == 0) return 0;
    uint64_t mask = rng;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4;
    mask |= mask >> 8; mask |= mask >> 16; mask |= mask >> 32;
    uint32_t val;
    do {
        val = orc_rs_u32(rs) & (uint32_t)mask;
    } while (val > rng);
    return (int64_t)val;
}

static double rs_gauss(orc_rs *rs)
{
    if (rs->has_gauss) {
        double t = rs->gauss;
        rs->has_gauss = 0;
        rs->gauss = 0.0;
        return t;
    }
    double f, x1, x2, r2;
    do {
        x1 = 2.0 * orc_rs_double(rs) - 1.0;
        x2 = 2.0 * orc_rs_double(rs) - 1.0;
        r2 = x1 * x1 + x2 * x2;
    } while (r2 >= 1.0 || r2 == 0.0);
    f = sqrt(-2.0 * log(r2) / r2);
    rs->gauss = f * x1;
    rs->has_gauss = 1;
    return f * x2;
}

static double rs_std_gamma_gt1(orc_rs *rs, double shape)
{
    /* Marsaglia-Tsang, shape > 1 branch of numpy legacy_standard_gamma */
    double b = shape - 1. / 3.;
    double c = 1. / sqrt(9 * b);
    for (;;) {
        double X, V, U;
        do {
            X = rs_gauss(rs);
            V = 1.0 + c * X;
        } while (V <= 0.0);
        V = V * V * V;
        U = orc_rs_double(rs);
        if (U < 1.0 - 0.0331 * (X * X) * (X * X)) return b * V;
        if (log(U) < 0.5 * X * X + b * (1. - V + log(V))) return b * V;
    }
}

double orc_rs_beta(orc_rs *rs, double a, double b)
{
    /* numpy legacy_beta, a > 1 or b > 1 branch (reference uses a=3, b=1.5: TTSP_env.py:13) */
    double Ga = rs_std_gamma_gt1(rs, a);
    double Gb = rs_std_gamma_gt1(rs, b);
    return Ga / (Ga + Gb);
}

/* =====================================================================================
 * Deterministic sin/cos: Cody-Waite 3-term reduction + minimax kernels, only IEEE
 * + - * and fma (each correctly rounded, written out explicitly; the compiler contracts
 * nothing) so CPU and GPU produce identical bits.  Valid |x| < ~1e6.
 * ===================================================================================== */
void orc_sincos(double x, double *s_out, double *c_out)
{
    const double TWO_OVER_PI = 6.36619772367581382433e-01;
    const double P1  = 1.57079632673412561417e+00;
    const double P2  = 6.07710050630396597660e-11;
    const double P3  = 2.02226624871116645580e-21;
    const double P3T = 8.47842766036889956997e-32;
    const double MAGIC = 6755399441055744.0; /* 1.5 * 2^52: round-to-nearest-even */
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;

    double fn = FMA(x, TWO_OVER_PI, MAGIC) - MAGIC;
    double r = FMA(-fn, P1, x);
    r = FMA(-fn, P2, r);
    r = FMA(-fn, P3, r);
    r = FMA(-fn, P3T, r);
    int32_t n = (int32_t)fn;   /* |x| < 2^31 * pi/2; one v_cvt_i32_f64 on the GPU */

    double z = r * r;
    double ps = FMA(z, S6, S5);
    ps = FMA(z, ps, S4);
    ps = FMA(z, ps, S3);
    ps = FMA(z, ps, S2);
    ps = FMA(z, ps, S1);
    double sn = FMA(r, z * ps, r);
    double pc = FMA(z, C6, C5);
    pc = FMA(z, pc, C4);
    pc = FMA(z, pc, C3);
    pc = FMA(z, pc, C2);
    pc = FMA(z, pc, C1);
    double hz = 0.5 * z;
    double w = 1.0 - hz;
    double cs = w + FMA(z * z, pc, (1.0 - w) - hz);

    switch (n & 3) {
    case 0: *s_out = sn;  *c_out = cs;  break;
    case 1: *s_out = cs;  *c_out = -sn; break;
    case 2: *s_out = -sn; *c_out = -cs; break;
    default: *s_out = -cs; *c_out = sn; break;
    }
}

/* =====================================================================================
 * Configuration (envs/__init__.py:7-50; ZoneEnvBase.py:39-53; point.xml [UPSTREAM])
 * ===================================================================================== */
void orc_default_config(int task, int num_zones, orc_config *c)
{
    memset(c, 0, sizeof(*c));
    c->task = task;
    c->num_zones = num_zones;
    c->num_steps = 2000;            /* envs/__init__.py:13,49 */
    c->max_cd = 150;                /* colour_match_env.py:16 */
    c->frameskip = 10;              /* [UPSTREAM] frameskip_binom_n, p = 1.0 */
    c->zones_size = 0.2;            /* ZoneEnvBase.py:51 */
    c->zones_keepout = 0.55;        /* ZoneEnvBase.py:50 */
    c->robot_keepout = 0.4;         /* [UPSTREAM] Engine.DEFAULT */
    c->extent = 3.0;                /* ZoneEnvBase.py:41,52 */
    c->placements_margin = 0.0;     /* [UPSTREAM] */
    c->time_saved_reward = 0.01;    /* TSP_env.py:15 */
    c->beta_a = 3.0;                /* TTSP_env.py:13 */
    c->beta_b = 1.5;
    /* point.xml [UPSTREAM-RECALL]: density 1; sphere r=.1 at origin; box half .05 at (.1,0,0) */
    const double PI = 3.14159265358979323846;
    const double density = 1.0;
    double m_s = density * (4.0 / 3.0 * PI * 0.1 * 0.1 * 0.1);
    double m_b = density * (8.0 * 0.05 * 0.05 * 0.05);
    c->timestep = 0.002;
    c->mass = m_s + m_b;
    c->com_x = 0.1 * m_b / c->mass;
    c->inertia_zz = 0.4 * m_s * (0.1 * 0.1) + m_b * (0.05 * 0.05 + 0.05 * 0.05) / 3.0 + m_b * (0.1 * 0.1);
    c->damping[0] = 0.01;
    c->damping[1] = 0.01;
    c->damping[2] = 0.005;
    c->gear = 0.3;
    c->forcerange = 0.05;
    c->vel_kv = 1.0;
    c->reward_exception = -10.0;    /* [UPSTREAM] Engine.DEFAULT */
}

int orc_zone_feat(const orc_config *cfg)
{
    /* TSP_env.py:27-29 (6); TTSP_env.py:78-84 (7); colour_match_env.py:70-73 (7) */
    return cfg->task == ORC_TASK_TSP ? 6 : 7;
}

/* =====================================================================================
 * reset  ([UPSTREAM] Engine.reset/build_layout/sample_layout/draw_placement/random_rot;
 * TSP_env.py:74-77; TTSP_env.py:19-21,73-76; colour_match_env.py:57-68,125-127)
 * ===================================================================================== */

static int hamming(const orc_env *e)
{
    /* colour_match_env.py:38-55 */
    int nb = 0, ng = 0, nr = 0;
    for (int i = 0; i < e->cfg.num_zones; i++) {
        if (e->colour[i] == 0) nb++;
        else if (e->colour[i] == 1) ng++;
        else nr++;
    }
    int to_blue = ng * 2 + nr;
    int to_green = nr * 2 + nb;
    int to_red = nb * 2 + ng;
    int d = to_blue < to_green ? to_blue : to_green;
    return d < to_red ? d : to_red;
}

static void forward(orc_env *e)
{
    /* [UPSTREAM] mj_kinematics for body "robot": slides act in the body's initial frame
     * (rotated by rot), hinge adds qpos[2]; velocities are Jacobian * qvel. */
    double cr = FMA(e->bq0, e->bq0, -(e->bq3 * e->bq3));   /* mju_quat2Mat entries */
    double sr = 2.0 * (e->bq0 * e->bq3);
    e->xpos[0] = FMA(cr, e->qpos[0], FMA(-sr, e->qpos[1], e->x0));
    e->xpos[1] = FMA(sr, e->qpos[0], FMA(cr, e->qpos[1], e->y0));
    e->xvelp[0] = FMA(cr, e->qvel[0], -(sr * e->qvel[1]));
    e->xvelp[1] = FMA(sr, e->qvel[0], cr * e->qvel[1]);
    e->xvelr = e->qvel[2];
    double hs, hc;
    orc_sincos(0.5 * e->qpos[2], &hs, &hc);
    e->xquat0 = FMA(e->bq0, hc, -(e->bq3 * hs));
    e->xquat3 = FMA(e->bq0, hs, e->bq3 * hc);
}

static int sample_layout(orc_env *e, orc_rs *rs)
{
    /* [UPSTREAM] Engine.sample_layout: robot first, then zone0..zone{Z-1}
     * (ZoneEnvBase.py:118-122 appends zones after the parent's entries). */
    const orc_config *c = &e->cfg;
    int Z = c->num_zones;
    double px[ORC_MAX_Z + 1] = { 0 }, py[ORC_MAX_Z + 1] = { 0 }, pk[ORC_MAX_Z + 1] = { 0 };
    for (int obj = 0; obj <= Z; obj++) {
        double keepout = obj == 0 ? c->robot_keepout : c->zones_keepout;
        /* draw_placement -> constrain_placement: (xmin + k, ymin + k, xmax - k, ymax - k) of the extents ... */
        double xlo = -c->extent + keepout, xhi = c->extent - keepout, ylo = xlo, yhi = xhi;
        /* ... or, [UPSTREAM] placements_dict_from_object for object i < len(<name>s_locations): the placement
         * (x - k, y - k, x + k, y + k) with k = keepout + 1e-9, constrained the same way (config_zone_fixed_1/_2,
         * envs/__init__.py:52-81); one placement -> no rs.choice draw, two uniform draws as usual */
        const double *fixed = 0;
        if (obj == 0 && c->n_robot_locations > 0) fixed = c->robot_location;
        if (obj > 0 && obj - 1 < c->n_zones_locations) fixed = c->zones_locations[obj - 1];
        if (fixed) {
            double k = keepout + 1e-9;
            xlo = (fixed[0] - k) + keepout; xhi = (fixed[0] + k) - keepout;
            ylo = (fixed[1] - k) + keepout; yhi = (fixed[1] + k) - keepout;
        }
        int conflicted = 1;
        double x = 0, y = 0;
        for (int t = 0; t < 100; t++) {
            x = orc_rs_uniform(rs, xlo, xhi);
            y = orc_rs_uniform(rs, ylo, yhi);
            int ok = 1;
            for (int j = 0; j < obj; j++) {
                double dx = x - px[j], dy = y - py[j];
                double dist = sqrt(dx * dx + dy * dy);
                if (dist < pk[j] + c->placements_margin + keepout) { ok = 0; break; }
            }
            if (ok) { conflicted = 0; break; }
        }
        if (conflicted) return 0;
        px[obj] = x; py[obj] = y; pk[obj] = keepout;
    }
    e->x0 = px[0]; e->y0 = py[0];
    for (int z = 0; z < Z; z++) { e->zone_xy[z][0] = px[z + 1]; e->zone_xy[z][1] = py[z + 1]; }
    return 1;
}

int orc_reset(orc_env *e, const orc_config *cfg, int64_t seed)
{
    if (cfg->num_zones < 1 || cfg->num_zones > ORC_MAX_Z) return -2;
    memset(e, 0, sizeof(*e));
    e->cfg = *cfg;
    e->goal_zone = -1;
    e->last_visit = -1;
    e->seed = seed;
    int Z = cfg->num_zones;
    orc_rs rs;

    /* task state is drawn from RandomState(self._seed) BEFORE Engine.reset increments it */
    if (cfg->task == ORC_TASK_TIMED) {
        /* TTSP_env.py:19-21 */
        orc_rs_seed(&rs, (uint32_t)seed);
        for (int z = 0; z < Z; z++)
            e->tmax[z] = (int32_t)(orc_rs_beta(&rs, cfg->beta_a, cfg->beta_b) * cfg->num_steps);
    } else if (cfg->task == ORC_TASK_COLOUR) {
        /* colour_match_env.py:57-68: the retry loop re-seeds identically each try */
        for (int t = 0; t < 100; t++) {
            orc_rs_seed(&rs, (uint32_t)seed);
            for (int z = 0; z < Z; z++) e->colour[z] = (int32_t)orc_rs_choice(&rs, 3);
            for (int z = 0; z < Z; z++) e->cooldown[z] = 0;
            e->goal_dist = hamming(e);
            if (e->goal_dist > 0) break;
        }
    }
    /* TSP_env.py:75: zones = [unvisited]*Z (memset above); TSP_hard_env.py:27-29: zones = zones_colours */
    if (cfg->task != ORC_TASK_COLOUR)
        for (int z = 0; z < Z; z++) e->visited[z] = (int32_t)((cfg->visited0 >> z) & 1u);

    /* [UPSTREAM] Engine.reset: _seed += 1; rs = RandomState(_seed) */
    orc_rs_seed(&rs, (uint32_t)(seed + 1));
    int ok = 0;
    for (int t = 0; t < 10000; t++) {
        if (sample_layout(e, &rs)) { ok = 1; break; }
        e->layout_restarts++;
    }
    if (!ok) return -3; /* ResamplingError */
    /* build_world_config: robot_rot = random_rot() = rs.uniform(0, 2*pi) unless the config gives 'robot_rot' */
    e->rot = cfg->robot_rot_fixed ? cfg->robot_rot : orc_rs_uniform(&rs, 0.0, 2 * 3.141592653589793);
    /* (one cosmetic random_rot per zone follows, ZoneEnvBase.py:132: unobservable) */
    /* [UPSTREAM] world.py rot2quat: [cos(rot/2), 0, 0, sin(rot/2)] */
    orc_sincos(e->rot / 2, &e->bq3, &e->bq0);
    e->steps = 0;
    e->done = 0;
    forward(e);
    return 0;
}

/* =====================================================================================
 * step  (TSP_env.py:45-49; TTSP_env.py:62-71; colour_match_env.py:95-101;
 * [UPSTREAM] Engine.step; MuJoCo mj_step for point.xml -- SURVEY.md Appendix A.4/B)
 * ===================================================================================== */

static double clampd(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }

/* [UPSTREAM] mju_isBad (engine_util_misc.c): NaN or beyond mjMAXVAL = 1e10 */
static int is_bad(double x) { return x != x || x > 1e10 || x < -1e10; }

/* (s, k) = sin/cos(theta) -> sin/cos(theta + d) for the small angle d = h * omega that one substep turns the hinge by:
 * Taylor kernels of sin d through d^9 and of cos d through d^10 (truncation < 3e-19 for |d| <= 0.1; the product's
 * config validation keeps |d| below 0.05, point.xml reaches 0.01) and the angle-sum formulas.  MuJoCo evaluates
 * sin/cos of the joint angle afresh in every mj_kinematics; re-anchoring on the exact sincos once per env step
 * (mj_env_step) keeps this within a few 1e-16 of that -- tests/test_dynamics_independent.py measures it. */
static void rotate_small(double d, double *s, double *k)
{
    double z = d * d;
    double ps = FMA(z, 1.0 / 362880.0, -1.0 / 5040.0);
    ps = FMA(z, ps, 1.0 / 120.0);
    ps = FMA(z, ps, -1.0 / 6.0);
    double sd = FMA(d, z * ps, d);
    double pc = FMA(z, -1.0 / 3628800.0, 1.0 / 40320.0);
    pc = FMA(z, pc, -1.0 / 720.0);
    pc = FMA(z, pc, 1.0 / 24.0);
    pc = FMA(z, pc, -0.5);
    double cd = FMA(z, pc, 1.0);
    double s2 = FMA(*k, sd, *s * cd);
    double k2 = FMA(-*s, sd, *k * cd);
    *s = s2;
    *k = k2;
}

/* The frameskip mj_step calls of one Engine.step for point.xml: 3 dof (slide x, slide y, hinge z), offset COM,
 * implicit joint damping, no active constraints.  Returns 1 when mj_checkAcc would have
 * raised mjWARN_BADQACC in some substep (a qacc is bad): MuJoCo then resets the data (qpos = qpos0, qvel = 0), and
 * mujoco-py turns the warning into a MujocoException once the call returns.
 *
 * (M + h diag(b)) qacc = qfrc with M = [[m,0,-mc s],[0,m,mc k],[-mc s,mc k,I0]] is solved by the Schur complement
 * on the hinge row.  With equal damping on the two slides (point.xml: 0.01 and 0.01) that complement,
 * I0 + h b2 - (mc)^2 (s^2 + k^2) / (m + h b), does not depend on the angle: one host-side reciprocal replaces the
 * division of every substep.  Operation order is what the HIP kernels execute, token for token. */
static int mj_env_step(orc_env *e, const double ctrl[2])
{
    const orc_config *c = &e->cfg;
    const double h = c->timestep, g = c->gear, F = c->forcerange;
    const double mc = c->mass * c->com_x;
    const double A00 = c->mass + h * c->damping[0];
    const double A11 = c->mass + h * c->damping[1];
    const double A22 = c->inertia_zz + h * c->damping[2];
    const double inv00 = 1.0 / A00, inv11 = 1.0 / A11;
    const int iso = c->damping[0] == c->damping[1];
    const double inv_den = 1.0 / FMA(-(mc * mc), inv00, A22);
    const double kvg = c->vel_kv * g;
    double *q = e->qpos, *v = e->qvel;

    /* sin/cos of the hinge angle from the half angle mj_kinematics' quaternion needs anyway */
    double hs, hc;
    orc_sincos(0.5 * q[2], &hs, &hc);
    double s = 2.0 * (hs * hc);
    double k = FMA(hc, hc, -(hs * hs));
    /* actuators: motor on site (gear .3 0 0 0 0 0); velocity servo on hinge (kv, gear .3); both force-limited */
    const double gf0 = g * clampd(ctrl[0], -F, F);
    const double kvc1 = c->vel_kv * ctrl[1];
    for (int i = 0; i < c->frameskip; i++) {
        double mcs = mc * s, mck = mc * k;
        double w2 = v[2] * v[2];
        double f1 = clampd(FMA(-kvg, v[2], kvc1), -F, F);
        /* qfrc = passive(-b v) - bias(centrifugal of the offset COM) + actuator */
        double rhs0 = FMA(-c->damping[0], v[0], FMA(mck, w2, gf0 * k));
        double rhs1 = FMA(-c->damping[1], v[1], FMA(mcs, w2, gf0 * s));
        double rhs2 = FMA(-c->damping[2], v[2], g * f1);
        double t0 = rhs0 * inv00, t1 = rhs1 * inv11;
        double num = FMA(-mck, t1, FMA(mcs, t0, rhs2));
        double a2;
        if (iso) {
            a2 = num * inv_den;
        } else {
            double den = FMA(-(mck * mck), inv11, FMA(-(mcs * mcs), inv00, A22));
            a2 = num / den;
        }
        double a0 = FMA(mcs, a2, rhs0) * inv00;
        double a1 = FMA(-mck, a2, rhs1) * inv11;
        if (is_bad(a0) || is_bad(a1) || is_bad(a2)) {
            q[0] = q[1] = q[2] = 0.0;       /* mj_resetData */
            v[0] = v[1] = v[2] = 0.0;
            return 1;
        }
        v[0] = FMA(h, a0, v[0]);
        v[1] = FMA(h, a1, v[1]);
        v[2] = FMA(h, a2, v[2]);
        q[0] = FMA(h, v[0], q[0]);
        q[1] = FMA(h, v[1], q[1]);
        q[2] = FMA(h, v[2], q[2]);
        /* the next substep's sin/cos: turn by d = h omega */
        rotate_small(h * v[2], &s, &k);
    }
    return 0;
}

int orc_step(orc_env *e, const float action[2], double *reward, int *done, int *goal_met)
{
    const orc_config *c = &e->cfg;
    const int Z = c->num_zones;
    if (e->done) return -1; /* 'Environment must be reset before stepping' */
    int event = 0;
    e->last_visit = -1;
    e->exception = 0;
    *goal_met = 0;

    /* colour_match_env.py:98-100: cooldowns tick before anything else */
    if (c->task == ORC_TASK_COLOUR)
        for (int z = 0; z < Z; z++)
            if (e->cooldown[z] > 0) e->cooldown[z] -= 1;

    /* set_mocaps() of the first substep: TSP_env.py:54-69 / colour_match_env.py:106-120.
     * Uses xpos of the last forward (= end of previous step); lowest index wins; one per step. */
    for (int z = 0; z < Z; z++) {
        int eligible = c->task == ORC_TASK_COLOUR ? (e->cooldown[z] == 0) : !e->visited[z];
        if (!eligible) continue;
        double dx = e->zone_xy[z][0] - e->xpos[0], dy = e->zone_xy[z][1] - e->xpos[1];
        double dist = sqrt(dx * dx + dy * dy);
        if (dist <= c->zones_size) {
            if (c->task == ORC_TASK_COLOUR) {
                e->colour[z] = (e->colour[z] + 1) % 3;   /* colour_match_env.py:26-36 */
                e->cooldown[z] = c->max_cd;
            } else {
                e->visited[z] = 1;
            }
            event = 1;
            e->last_visit = z;
            break;
        }
    }

    /* [UPSTREAM] Engine.step: ctrl = clip(action, ctrlrange) ; frameskip x mj_step ; forward */
    double ctrl[2];
    ctrl[0] = clampd((double)action[0], -1.0, 1.0);
    ctrl[1] = clampd((double)action[1], -1.0, 1.0);
    /* np.clip keeps a NaN action; `try: set_mocaps(); sim.step() except MujocoException: exception = True; break` */
    e->exception = mj_env_step(e, ctrl);
    forward(e);   /* exception: obs() still runs sim.forward() on the reset data */

    if (e->exception) {
        /* `if exception: self.done = True; reward = self.reward_exception; info['exception'] = True`
         * -- no reward(), no goal_met(); then steps += 1 and the time limit as always */
        e->done = 1;
        e->steps += 1;
        *reward = c->reward_exception;
        *done = 1;
        return 0;
    }

    /* reward(): TSP_env.py:41-42 / colour_match_env.py:86-93 */
    double r = 0.0;
    if (c->task == ORC_TASK_COLOUR) {
        if (event) {
            int nd = hamming(e);
            r = (double)(e->goal_dist - nd);
            e->goal_dist = nd;
        }
    } else {
        r = event ? 1.0 : 0.0;
    }

    /* goal_met(): TSP_env.py:71-72 / colour_match_env.py:122-123; reward_goal uses pre-increment steps */
    int goal;
    if (c->task == ORC_TASK_COLOUR) {
        goal = e->goal_dist == 0;
    } else {
        goal = 1;
        for (int z = 0; z < Z; z++) if (!e->visited[z]) { goal = 0; break; }
    }
    if (goal) {
        r += (double)(c->num_steps - e->steps) * c->time_saved_reward;  /* TSP_env.py:37-39 */
        e->done = 1;
        *goal_met = 1;
    }
    e->steps += 1;
    if (e->steps >= c->num_steps) e->done = 1;

    /* TTSP_env.py:62-71: timeout of any unvisited zone ends the episode (after the base step) */
    if (c->task == ORC_TASK_TIMED && !e->done) {
        for (int z = 0; z < Z; z++)
            if (!e->visited[z] && (e->tmax[z] - e->steps) <= 0) { e->done = 1; break; }
    }
    *reward = r;
    *done = e->done;
    return 0;
}

/* ---- solver-ordered variant ---- */
static double order_dist(const orc_env *e)
{
    /* TSP_order_env.py:52-57 */
    if (e->route_len == 0) return 0.0;
    int g = e->route[0];
    double dx = e->zone_xy[g][0] - e->xpos[0], dy = e->zone_xy[g][1] - e->xpos[1];
    return sqrt(dx * dx + dy * dy);
}

static void order_obs_sees_route(orc_env *e)
{
    for (int i = 0; i < e->route_len; i++) e->obs_route[i] = e->route[i];
    e->obs_route_len = e->route_len;
}

int orc_reset_order(orc_env *e, const orc_config *cfg, int64_t seed, const int32_t *rank, int fresh_first_obs)
{
    const int Z = cfg->num_zones;
    if (Z < 1 || Z > ORC_MAX_Z) return -2;
    for (int z = 0; z < Z; z++)
        if (rank[z] < 0 || rank[z] >= Z) return -1;
    /* self.route belongs to the env object, not to the episode: it survives Engine.reset() */
    int32_t left[ORC_MAX_Z];
    const int n_left = e->route_len;
    for (int i = 0; i < n_left; i++) left[i] = e->route[i];
    /* :109-110  self.zones = [unvisited] * num_cities; init_obs = super().reset() -> ... -> obs() -> obs_zones() (:37-47)
     * with the route above */
    int rc = orc_reset(e, cfg, seed);
    if (rc) return rc;
    for (int i = 0; i < n_left; i++) e->obs_route[i] = left[i];
    e->obs_route_len = n_left;
    /* :111  generate_route() */
    for (int z = 0; z < Z; z++) e->route[rank[z]] = z;
    e->route_len = Z;
    /* :112  last_dist_to_goal = dist_to_goal() */
    e->last_dist = order_dist(e);
    if (fresh_first_obs) order_obs_sees_route(e);
    return 0;
}

int orc_step_order(orc_env *e, const float action[2], double *reward, int *done, int *goal_met, double *shaped_reward)
{
    int rc = orc_step(e, action, reward, done, goal_met);
    if (rc) return rc;
    if (e->last_visit >= 0) {
        /* set_mocaps: self.route.remove(h_index) (:90) */
        int k = 0;
        for (int i = 0; i < e->route_len; i++)
            if (e->route[i] != e->last_visit) e->route[k++] = e->route[i];
        e->route_len = k;
        e->last_dist = order_dist(e);                  /* shaped_reward(): :64-66 */
        *shaped_reward = 0.0;
    } else {
        double d = order_dist(e);                      /* :68-71 */
        *shaped_reward = e->last_dist - d;
        e->last_dist = d;
    }
    order_obs_sees_route(e);                           /* Engine.step's obs() ran after set_mocaps() */
    return 0;
}

void orc_order_vals(const orc_env *e, float *vals)
{
    /* obs_zones: np.power(0.5, self.route.index(i)) if i in self.route else 0 (:41-45) */
    for (int z = 0; z < e->cfg.num_zones; z++) vals[z] = 0.f;
    for (int i = 0; i < e->obs_route_len; i++) vals[e->obs_route[i]] = (float)ldexp(1.0, -i);
}

int orc_order_route(const orc_env *e, int32_t *route)
{
    for (int i = 0; i < e->route_len; i++) route[i] = e->route[i];
    return e->route_len;
}

/* ---- goal-conditioned variant ---- */
static double dist_to_goal(const orc_env *e)
{
    /* TSP_next_city_env.py:41-45: sqrt(sum(square(goal_pos - robot_pos))) on the current world position */
    double dx = e->zone_xy[e->goal_zone][0] - e->xpos[0], dy = e->zone_xy[e->goal_zone][1] - e->xpos[1];
    return sqrt(dx * dx + dy * dy);
}

int orc_set_goal(orc_env *e, int goal)
{
    /* :86; ColourMatchNextCityEnv.set_goal only asserts the range */
    if (goal < 0 || goal >= e->cfg.num_zones || (e->cfg.task != ORC_TASK_COLOUR && e->visited[goal])) return -1;
    e->goal_zone = goal;
    e->last_dist = dist_to_goal(e);                                             /* :87-88 */
    return 0;
}

int orc_step_goal(orc_env *e, const float action[2], double *reward, int *done, int *goal_met,
                  double *shaped_reward, int *need_next_goal)
{
    if (e->goal_zone < 0) return -2;                                            /* :54 */
    int rc = orc_step(e, action, reward, done, goal_met);
    if (rc) return rc;
    int reached = e->last_visit == e->goal_zone;    /* new_city_reached and zones[goal_zone] == visited */
    if (reached) {
        *shaped_reward = 0.0;                                                   /* :60-61 */
    } else {
        double d = dist_to_goal(e);
        *shaped_reward = e->last_dist - d;                                      /* :63-66 */
        e->last_dist = d;
        /* colour_match_next_city_env.py step: a zone other than the goal changed colour */
        if (e->cfg.task == ORC_TASK_COLOUR && e->last_visit >= 0) *shaped_reward -= 1.0;
    }
    if (reached || *done) {                                                     /* :69-72; TTSP_next_city_env.py:46-49 */
        *need_next_goal = 1;
        e->goal_zone = -1;
    } else {
        *need_next_goal = 0;
    }
    return 0;
}

int orc_solver_next_goal(const orc_env *e)
{
    /* zone-goals/envs/colour_match_solver_env.py:57-97 */
    const int Z = e->cfg.num_zones;
    int n[3] = { 0, 0, 0 };
    for (int z = 0; z < Z; z++) n[e->colour[z]]++;
    int to[3] = { n[1] * 2 + n[2], n[2] * 2 + n[0], n[0] * 2 + n[1] };   /* dist_to_blue, _green, _red */
    int mn = to[0] < to[1] ? to[0] : to[1];
    if (to[2] < mn) mn = to[2];
    int best = -1;
    double bd = 0.0;
    for (int z = 0; z < Z; z++) {
        int c = e->colour[z];
        if (!(to[(c + 1) % 3] == mn || to[(c + 2) % 3] == mn)) continue;
        double dx = e->zone_xy[z][0] - e->xpos[0], dy = e->zone_xy[z][1] - e->xpos[1];
        double d = sqrt(dx * dx + dy * dy);
        if (best < 0 || d < bd) { best = z; bd = d; }                      /* candidate_zones.sort()[0]: (dist, index) */
    }
    return best;
}

/* =====================================================================================
 * obs  (ZoneEnvBase.py:190-192,217-224; TSP_env.py:31-35; TTSP_env.py:23-27,86-92;
 * colour_match_env.py:75-80; wrappers.py:136-142).  Values are the reference's float64
 * results (numpy 1.21.1 promotion rules, requirements.txt:4) cast to float32, which is
 * what the consumer does at src/utils/format.py:27-28.
 * ===================================================================================== */
void orc_obs(const orc_env *e, float *o, float *zo)
{
    const orc_config *c = &e->cfg;
    const int Z = c->num_zones, F = orc_zone_feat(c);
    o[0] = (float)(1.0 - (double)e->steps / (double)c->num_steps);
    o[1] = (float)(e->xpos[0] / 3.0);
    o[2] = (float)(e->xpos[1] / 3.0);
    /* robot_quat.astype(float32); float32**2 and 2*float32 promote to float64 under numpy 1.21 */
    double q0 = (double)(float)e->xquat0, q3 = (double)(float)e->xquat3;
    o[3] = (float)(q0 * q0 - q3 * q3);
    o[4] = (float)((2.0 * q0) * q3);
    o[5] = (float)(e->xvelp[0] / 1.5);
    o[6] = (float)(e->xvelp[1] / 1.5);
    o[7] = (float)(e->xvelr / 3.0);
    for (int z = 0; z < Z; z++) {
        float *row = zo + (size_t)z * F;
        row[0] = (float)(e->zone_xy[z][0] / 3.0);
        row[1] = (float)(e->zone_xy[z][1] / 3.0);
        if (c->task == ORC_TASK_COLOUR) {
            /* Blue [0,0,1], Green [0,1,0], Red [1,0,0]  (ZoneEnvBase.py:68-77) */
            row[2] = e->colour[z] == 2 ? 1.f : 0.f;
            row[3] = e->colour[z] == 1 ? 1.f : 0.f;
            row[4] = e->colour[z] == 0 ? 1.f : 0.f;
            row[5] = 0.25f;
            row[6] = (float)((double)(float)e->cooldown[z] / (double)c->max_cd);
        } else {
            /* Cyan [0,1,1] unvisited, Yellow [1,1,0] visited  (TSP_env.py:9-10) */
            row[2] = e->visited[z] ? 1.f : 0.f;
            row[3] = 1.f;
            row[4] = e->visited[z] ? 0.f : 1.f;
            row[5] = 0.25f;
            if (c->task == ORC_TASK_TIMED)
                row[6] = e->visited[z] ? 1.f
                                       : (float)((double)(e->tmax[z] - e->steps) / (double)c->num_steps);
        }
    }
}

/* =====================================================================================
 * Scripted policies (the build's own deterministic action sources; not in the reference).
 * Functions of the float32 observation only, so any consumer can reproduce them.
 * ===================================================================================== */

static void philox4x32_10(uint32_t ctr[4], const uint32_t key_in[2])
{
    uint32_t k0 = key_in[0], k1 = key_in[1];
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * ctr[0];
        uint64_t p1 = (uint64_t)0xCD9E8D57u * ctr[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ ctr[1] ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ ctr[3] ^ k1;
        uint32_t n3 = (uint32_t)p0;
        ctr[0] = n0; ctr[1] = n1; ctr[2] = n2; ctr[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

void orc_policy(int policy, const orc_config *cfg, const float *o, const float *zo,
                uint64_t env_index, uint32_t step_index, uint64_t policy_seed, float action[2])
{
    const int Z = cfg->num_zones, F = orc_zone_feat(cfg);
    if (policy == ORC_POLICY_UNIFORM) {
        uint32_t ctr[4] = { (uint32_t)env_index, (uint32_t)(env_index >> 32), step_index, 0u };
        uint32_t key[2] = { (uint32_t)policy_seed, (uint32_t)(policy_seed >> 32) };
        philox4x32_10(ctr, key);
        action[0] = 2.0f * ((float)(ctr[0] >> 8) * 5.9604644775390625e-08f) - 1.0f;
        action[1] = 2.0f * ((float)(ctr[1] >> 8) * 5.9604644775390625e-08f) - 1.0f;
        return;
    }
    /* greedy: steer to the nearest eligible zone */
    double px = 3.0 * (double)o[1], py = 3.0 * (double)o[2];   /* exact: 3 * float has <= 26 bits */
    double hx = (double)o[3], hy = (double)o[4];
    int target_colour = -1;
    if (cfg->task == ORC_TASK_COLOUR) {
        int cnt[3] = { 0, 0, 0 };
        for (int z = 0; z < Z; z++) {
            const float *row = zo + (size_t)z * F;
            int col = row[4] != 0.f ? 0 : (row[3] != 0.f ? 1 : 2);
            cnt[col]++;
        }
        target_colour = 0;
        if (cnt[1] > cnt[target_colour]) target_colour = 1;
        if (cnt[2] > cnt[target_colour]) target_colour = 2;
    }
    int best = -1;
    double bd2 = 0.0, bdx = 0.0, bdy = 0.0;
    for (int z = 0; z < Z; z++) {
        const float *row = zo + (size_t)z * F;
        int eligible;
        if (cfg->task == ORC_TASK_COLOUR) {
            int col = row[4] != 0.f ? 0 : (row[3] != 0.f ? 1 : 2);
            eligible = (row[6] == 0.f) && (col != target_colour);
        } else {
            eligible = row[2] == 0.f;
        }
        if (!eligible) continue;
        double dx = FMA(3.0, (double)row[0], -px), dy = FMA(3.0, (double)row[1], -py);
        double d2 = FMA(dx, dx, dy * dy);
        if (best < 0 || d2 < bd2) { best = z; bd2 = d2; bdx = dx; bdy = dy; }
    }
    float a0 = 0.f, a1 = 0.f;
    if (best >= 0 && bd2 > 1e-18) {
        double n = sqrt(bd2);
        double cs = FMA(hx, bdx, hy * bdy) / n;
        double sn = FMA(hx, bdy, -(hy * bdx)) / n;
        if (cs < 0.0) a1 = sn >= 0.0 ? 1.f : -1.f;
        else a1 = (float)clampd(4.0 * sn, -1.0, 1.0);
        a0 = cs > 0.8 ? 1.f : 0.f;
    }
    action[0] = a0;
    action[1] = a1;
}

/* =====================================================================================
 * Batch driver: N independent envs, closed loop, auto-reset as penv.py:7-11.
 * ===================================================================================== */
int64_t orc_rollout_wrapped(const orc_config *cfg, int n_env, int n_steps, int policy,
                    const int64_t *seeds0, int64_t seed_stride, int32_t seed_period, uint64_t policy_seed,
                    uint64_t env_index0, int n_threads,
                    double *reward_sum, int32_t *episodes, double *last_return,
                    int32_t *last_len, float *final_obs8, float *final_zone_obs)
{
    const int Z = cfg->num_zones, F = orc_zone_feat(cfg);
    int64_t total = 0;
    (void)n_threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(n_threads > 0 ? n_threads : 1) reduction(+:total)
#endif
    for (int i = 0; i < n_env; i++) {
        orc_env e;
        float o[8], zo[ORC_MAX_Z * 7], a[2];
        int64_t seed = seeds0[i];
        int32_t n_ep = 0, ep_len = 0, l_len = 0;
        double ep_ret = 0.0, l_ret = 0.0, rs = 0.0;
        if (orc_reset(&e, cfg, seed) != 0) continue;
        orc_obs(&e, o, zo);
        for (int t = 0; t < n_steps; t++) {
            double r; int d, g;
            orc_policy(policy, cfg, o, zo, env_index0 + (uint64_t)i, (uint32_t)t, policy_seed, a);
            orc_step(&e, a, &r, &d, &g);
            total++;
            rs += r; ep_ret += r; ep_len++;
            if (d) {
                l_ret = ep_ret; l_len = ep_len; n_ep++;
                ep_ret = 0.0; ep_len = 0;
                /* bank of seed_period maps per env, replayed in order (the device bank wraps the same way) */
                seed = seeds0[i] + (seed_period > 0 ? (int64_t)(n_ep % seed_period) : (int64_t)n_ep) * seed_stride;
                if (orc_reset(&e, cfg, seed) != 0) break;
            }
            orc_obs(&e, o, zo);
        }
        if (reward_sum) reward_sum[i] = rs;
        if (episodes) episodes[i] = n_ep;
        if (last_return) last_return[i] = l_ret;
        if (last_len) last_len[i] = l_len;
        if (final_obs8) memcpy(final_obs8 + (size_t)i * 8, o, sizeof(o));
        if (final_zone_obs) memcpy(final_zone_obs + (size_t)i * Z * F, zo, sizeof(float) * Z * F);
    }
    return total;
}

int64_t orc_rollout(const orc_config *cfg, int n_env, int n_steps, int policy,
                    const int64_t *seeds0, int64_t seed_stride, uint64_t policy_seed,
                    uint64_t env_index0, int n_threads,
                    double *reward_sum, int32_t *episodes, double *last_return,
                    int32_t *last_len, float *final_obs8, float *final_zone_obs)
{
    return orc_rollout_wrapped(cfg, n_env, n_steps, policy, seeds0, seed_stride, 0, policy_seed, env_index0, n_threads,
                               reward_sum, episodes, last_return, last_len, final_obs8, final_zone_obs);
}

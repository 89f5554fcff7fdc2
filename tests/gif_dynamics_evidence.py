"""TEST INFRASTRUCTURE (not collected by pytest; uses the oracle, like everything under tests/).  What the reference's own PointTSP animation says about the robot's time constant.

Input: tests/golden/gif_pointtsp_track.json (robot track + zone layout of one whole episode rendered by the real
MuJoCo stack, extracted by tests/golden/make_gif_track.py).  CPU only, oracle constants only.  Two questions:

 (1) *Lower bound.*  How many env steps does a robot with time constant tau = m / b NEED to follow the track?  The
     planar dynamics of point.xml are linear in the world frame (m v' = g F u - b v with |u| <= 1 along
     the heading), so "is there a control sequence that passes the 57 measured positions at frame times k K" is a
     convex problem: minimise the tracking residual over u[n] in the unit disc (heading rate unconstrained -- that
     only helps the heavy robot), K env steps per frame, solved by accelerated projected gradient; each frame's time
     may slip by half a frame (the GIF's sampling is uneven).  The smallest K with residual <= 5 cm (2-3 pixels)
     times 56 frames is a lower bound on the episode's length.
 (2) *Upper bound.*  How long does the oracle's robot take on the same map?  The zones and the start are put into
     the oracle (fixed locations), and a pursuit controller drives through the zones in the animation's order.

The metric scale comes from constants that ARE in the tree: zone radius 0.2 (ZoneEnvBase.py:51) against the measured
disc size, cross-checked by the placement rule (no zone beyond 3 - 0.55, no two zones closer than 1.1).
Run:  python tests/gif_dynamics_evidence.py
"""
import json
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H_STEP = 0.02            # env step: 10 substeps of 0.002 s


def load_track():
    with open(os.path.join(ROOT, "tests", "golden", "gif_pointtsp_track.json")) as fh:
        d = json.load(fh)
    zones = np.array(d["zones_xy"])
    scale = 0.4 / float(np.mean(d["zone_diameter"]))
    visited = np.array(d["visited_count"])
    last = int(np.nonzero(np.diff(visited) < 0)[0][0])          # the frame before the reset
    robot = np.array(d["robot_xy"])[: last + 1]
    moving = int(np.nonzero(np.linalg.norm(np.diff(robot, axis=0), axis=1) > 1e-6)[0][-1]) + 2
    return {"zones": zones * scale, "robot": robot[:moving] * scale, "visited": visited[:moving], "scale": scale,
            "raw_zones": zones}


def scale_window(raw_zones):
    """Bounds on the metric scale from the placement rule of Engine.draw_placement with the zone envs' settings
    (ZoneEnvBase.py:50,52): |coordinate| <= 3 - 0.55, pairwise distance >= 2 * 0.55."""
    dmin = min(np.linalg.norm(a - b) for i, a in enumerate(raw_zones) for b in raw_zones[:i])
    return 1.1 / dmin, 2.45 / np.abs(raw_zones).max()


def constants(density):
    """point.xml as SURVEY.md A.3 recalls it, with the geom density as the free parameter."""
    m = density * (4.0 / 3.0 * math.pi * 0.1 ** 3 + 0.1 ** 3)
    b, gear, force = 0.01, 0.3, 0.05
    return m / b, gear * force / b                     # tau [s], terminal speed [m/s]


def track_fit(track, tau, v_term, steps_per_frame, rounds=3, iters=1200):
    """Smallest tracking residual (rms, max, in metres) any |u| <= 1 control reaches, and the share of saturated steps."""
    p = track - track[0]
    n_frames, k = len(p), steps_per_frame
    n = n_frames * k
    a = math.exp(-H_STEP / tau)
    j = np.arange(n + 1)
    resp = np.cumsum(np.where(j >= 1, (1 - a) * a ** (np.maximum(j, 1) - 1) * v_term, 0.0)) * H_STEP

    def rows(ts):
        g = np.zeros((len(ts), n))
        for i, s in enumerate(ts):
            g[i, :s] = resp[s - np.arange(s)]
        return g

    ts = np.arange(n_frames) * k
    u = np.zeros((n, 2))
    for rd in range(rounds):
        g = rows(ts)
        lip = np.linalg.norm(g, 2) ** 2
        y, t, p0 = u.copy(), 1.0, np.zeros(2)
        for _ in range(iters):
            un = y - g.T @ (g @ y + p0 - p) / lip
            un /= np.maximum(np.linalg.norm(un, axis=1, keepdims=True), 1.0)
            tn = (1 + math.sqrt(1 + 4 * t * t)) / 2
            y, u, t = un + (t - 1) / tn * (un - u), un, tn
            p0 = -(g @ u - p).mean(0)
        vel, pos = np.zeros((n + 1, 2)), np.zeros((n + 1, 2))
        for s in range(n):
            vel[s + 1] = a * vel[s] + (1 - a) * v_term * u[s]
            pos[s + 1] = pos[s] + H_STEP * vel[s + 1]
        pos += p0
        if rd == rounds - 1:
            break
        slip = [max(0, i * k - k // 2) + int(np.argmin(np.linalg.norm(
            pos[max(0, i * k - k // 2): min(n, i * k + k // 2) + 1] - p[i], axis=1))) for i in range(n_frames)]
        ts = np.maximum.accumulate(np.array(slip))
    err = np.linalg.norm(pos[ts] - p, axis=1)
    return float(np.sqrt((err ** 2).mean())), float(err.max()), float((np.linalg.norm(u[: ts[-1]], axis=1) > 0.999).mean())


def visiting_order(tr):
    """Zone indices in the order the animation colours them (the zone nearest to the robot when the count goes up)."""
    order, left = [], list(range(len(tr["zones"])))
    for f in range(1, len(tr["visited"])):
        for _ in range(int(tr["visited"][f] - tr["visited"][f - 1])):
            seg = tr["robot"][f - 1: f + 1]
            z = min(left, key=lambda i: min(np.linalg.norm(seg - tr["zones"][i], axis=1)))
            order.append(z)
            left.remove(z)
    return order + left                                   # the last zone ends the episode (never drawn yellow)


def oracle_replay(tr, density, order, max_steps=6000):
    """Env steps the oracle's robot needs for the animation's tour (pursuit controller: full throttle, turn towards the
    next zone, throttle reversed while the zone is behind)."""
    sys.path.insert(0, ROOT)
    from oracle import oracle as O
    z = len(tr["zones"])
    m = density * (4.0 / 3.0 * math.pi * 0.1 ** 3 + 0.1 ** 3)
    m_b = density * 0.1 ** 3
    cfg = O.default_config(O.TASK_TSP, z, num_steps=max_steps, zones_keepout=0.2, robot_keepout=0.1, mass=m,
                           com_x=0.1 * m_b / m,
                           inertia_zz=density * (0.4 * (4.0 / 3.0 * math.pi * 1e-3) * 0.01 + 1e-3 * (0.005 / 3) + 1e-3 * 0.01))
    cfg.n_zones_locations, cfg.n_robot_locations = z, 1
    cfg.robot_location[0], cfg.robot_location[1] = tr["robot"][0]
    for i in range(z):
        cfg.zones_locations[i][0], cfg.zones_locations[i][1] = tr["zones"][i]
    # initial heading: towards the first zone (unknown in the picture; the choice moves the count by a few steps)
    first = tr["zones"][order[0]] - tr["robot"][0]
    cfg.robot_rot_fixed, cfg.robot_rot = 1, math.atan2(first[1], first[0])
    env = O.OracleEnv(cfg)
    env.reset(1)
    tau = m / 0.01
    for step in range(max_steps):
        e = env.e
        nxt = next((i for i in order if not e.visited[i]), None)
        if nxt is None:
            return step
        pos, vel = np.array(e.xpos[:]), np.array(e.xvelp[:])
        target = np.array(e.zone_xy[nxt][:])
        heading = 2.0 * math.atan2(e.xquat3, e.xquat0)
        want = target - pos - 0.7 * tau * vel                  # lead: cancel the velocity that will coast past
        ang = math.atan2(want[1], want[0]) - heading
        ang = (ang + math.pi) % (2 * math.pi) - math.pi
        fwd = 1.0 if abs(ang) < math.pi / 2 else -1.0
        if fwd < 0:
            ang = (ang + 2 * math.pi) % (2 * math.pi) - math.pi
        _, done, _ = env.step((fwd, max(-1.0, min(1.0, 4.0 * ang))))
        if done:
            return step + 1
    return max_steps


def main():
    tr = load_track()
    lo, hi = scale_window(tr["raw_zones"])
    print(f"metric scale from the zone discs: {tr['scale']:.3f} (floor half width {3.5 * tr['scale']:.2f} m); "
          f"placement rule allows [{lo:.3f}, {hi:.3f}]")
    disp = np.linalg.norm(np.diff(tr["robot"], axis=0), axis=1)
    plateau = float(np.median(np.sort(disp)[-12:]))
    frames = len(tr["robot"]) - 1
    print(f"{frames} frames, path {disp.sum():.1f} m, plateau displacement {plateau:.3f} m per frame -> at the terminal "
          f"speed 1.5 m/s one frame = {plateau / 1.5 / H_STEP:.1f} env steps, the episode = {frames * plateau / 1.5 / H_STEP:.0f} steps")
    order = visiting_order(tr)
    for density in (1.0, 5.0):
        tau, vt = constants(density)
        need = None
        print(f"density {density:g}: tau = {tau:.2f} s")
        for k in (13, 14, 15, 16, 18, 20, 24, 28, 32, 36):
            rms, worst, sat = track_fit(tr["robot"], tau, vt, k)
            if need is None and rms <= 0.05:
                need = k
            print(f"   {k:2d} steps per frame ({frames * k:4d} steps): residual rms {rms:.3f} m, max {worst:.3f} m, "
                  f"saturated {sat:.2f}", flush=True)
        print(f"   -> needs >= {need} steps per frame = {frames * need if need else '> 2016'} env steps for 5 cm rms; "
              f"oracle replay (pursuit controller): {oracle_replay(tr, density, order)} steps")


if __name__ == "__main__":
    main()

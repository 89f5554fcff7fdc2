"""The reference's own PointTSP animation (gifs/pointtsp.gif, a rendering by the real MuJoCo stack) against the oracle's
model constants: DESIGN.md section 0.2.  The track was extracted once by tests/golden/make_gif_track.py (data only); the
fit is tests/gif_dynamics_evidence.py.  This is evidence for ONE constant (the geom density of point.xml, i.e. the
time constant m / b), not a parity pin: the oracle stays "parity unpinned" for the dynamics half."""
import importlib.util
import os

import numpy as np

from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _mod():
    spec = importlib.util.spec_from_file_location("gif_dynamics_evidence",
                                                  os.path.join(ROOT, "tests", "gif_dynamics_evidence.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_metric_scale_of_the_animation_is_consistent_with_the_placement_rule():
    """Three in-tree constants fix the picture's scale independently: the zone radius (ZoneEnvBase.py:51), and the
    placement rule's bounds (ZoneEnvBase.py:50,52: no zone beyond 3 - 0.55, no two closer than 1.1)."""
    g = _mod()
    tr = g.load_track()
    lo, hi = g.scale_window(tr["raw_zones"])
    assert lo < hi and lo - 0.01 <= tr["scale"] <= hi + 0.01
    assert len(tr["zones"]) == 15 and tr["visited"][-1] == 14 and len(tr["robot"]) == 57
    # the homography behind the numbers: interior tile boundaries land on whole tiles
    import json
    with open(os.path.join(ROOT, "tests", "golden", "gif_pointtsp_track.json")) as fh:
        d = json.load(fh)
    row = np.array(d["tile_check_row"])
    assert len(row) == 19 and np.abs(row - np.arange(1, 20)).max() < 0.08


def test_animation_track_needs_the_oracles_time_constant_not_the_five_times_heavier_robot():
    """With the time base the plateau speed sets (terminal speed 1.5 m/s = the oracle's g F / b and the normaliser of
    ZoneEnvBase.py:223), a robot with the oracle's m / b follows the measured track to about 5 cm rms; the density-5
    robot (the alternative SURVEY.md A.3 could not exclude) misses it by half a metre whatever it does."""
    g = _mod()
    tr = g.load_track()
    cfg = O.default_config(O.TASK_TSP, 15)
    tau, v_term = cfg.mass / cfg.damping[0], cfg.gear * cfg.forcerange / cfg.damping[0]
    assert abs(tau - g.constants(1.0)[0]) < 1e-12 and abs(v_term - 1.5) < 1e-12
    disp = np.linalg.norm(np.diff(tr["robot"], axis=0), axis=1)
    k = int(round(float(np.median(np.sort(disp)[-12:])) / v_term / g.H_STEP))
    assert k == 16                                             # one frame = 16 env steps, the episode ~ 890 steps
    light = g.track_fit(tr["robot"], tau, v_term, k, rounds=2, iters=600)
    heavy = g.track_fit(tr["robot"], 5.0 * tau, v_term, k, rounds=2, iters=600)
    assert light[0] < 0.065 and heavy[0] > 0.40 and heavy[1] > 1.0, (light, heavy)
    # and with the time base left free the heavy robot still needs 1.5 x the light one's steps for the same residual
    assert g.track_fit(tr["robot"], 5.0 * tau, v_term, 24, rounds=2, iters=600)[0] > light[0]


def test_solver_ordered_agent_reaches_the_readmes_return_on_the_evaluation_maps(zenv_mod):
    """README.md:59-69: "Solver 25.30" on PointTSP = 15 zones in about 970 steps on the maps 1000000-1000099, which ARE
    pinned here (numpy goldens).  The oracle's robot, zones in the built-in route's order under a hand-written pursuit
    controller, lands within half a point of it (24.9 over all 100 maps); the density-5 robot cannot finish half of the
    maps inside the 2 000-step horizon (tests/readme_solver_return.py: 15.5 at its best setting)."""
    spec = importlib.util.spec_from_file_location("readme_solver_return",
                                                  os.path.join(ROOT, "tests", "readme_solver_return.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    seeds = range(1000000, 1000100, 10)
    ret, steps, finished = m.mean_return(O, zenv_mod, 1.0, seeds, 0.3)
    assert finished == len(seeds) and 24.4 <= ret <= 25.8 and 920 <= steps <= 1060, (ret, steps)
    heavy = m.mean_return(O, zenv_mod, 5.0, seeds, 0.2)[0]                  # its best setting
    assert heavy < 18.0, heavy


def test_colour_cycle_and_cooldown_of_the_colourmatch_animation():
    """gifs/colourmatch.gif (rendered by the reference's stack): every colour change of a zone follows the cycle the oracle
    restates from colour_match_env.py:106-120 (Blue -> Green -> Red -> Blue), and where the robot re-triggers the zone it is
    parked on, the two changes are a cooldown (max_cd = 150 steps) apart -- five frames of about 28 steps."""
    import json
    with open(os.path.join(ROOT, "tests", "golden", "gif_colourmatch_changes.json")) as fh:
        d = json.load(fh)
    seen = {(c["from"], c["to"]) for c in d["changes"]}
    assert len(d["changes"]) >= 12 and len(seen) == 3
    # the oracle: the robot parked on zone 0, zero action; colours read from the observation rows (RGB of ZoneEnvBase._rgb)
    cfg = O.default_config(O.TASK_COLOUR, 6, num_steps=2000, robot_keepout=0.02, zones_keepout=0.02)
    cfg.n_zones_locations, cfg.n_robot_locations = 1, 1
    cfg.robot_location[0], cfg.robot_location[1] = 1.0, 1.0
    cfg.zones_locations[0][0], cfg.zones_locations[0][1] = 1.1, 1.0          # 0.1 from the robot: inside the 0.2 radius
    env = O.OracleEnv(cfg)
    env.reset(7)

    def colour0():
        row = env.obs()[1][0]
        return "RGB"[int(np.argmax(row[2:5]))]
    cycle, steps, last = set(), [], colour0()
    for t in range(1, 460):
        env.step((0.0, 0.0))
        now = colour0()
        if now != last:
            cycle.add((last, now))
            steps.append(t)
            last = now
    assert cycle == {("B", "G"), ("G", "R"), ("R", "B")} and len(steps) == 4
    assert set(np.diff(steps)) == {150}                 # max_cd = 150, decremented before the test: eligible again 150 steps on
    assert seen == cycle
    # the animation's parked robot: zone (0.87, 2.73) changes at frames 6 and 11, the robot within 0.3 of it in between
    first = [c for c in d["changes"] if abs(c["zone_xy"][0] - 0.86) < 0.05 and abs(c["zone_xy"][1] - 2.73) < 0.05]
    assert [c["frame"] for c in first] == [6, 11]
    rob = np.array(d["robot_xy"][6:11])
    assert np.abs(rob - rob[0]).max() < 0.3
    disp = np.linalg.norm(np.diff(np.array(d["robot_xy"][:6]), axis=0), axis=1) * 0.854      # metres per frame on the way in
    steps_per_frame = float(np.max(disp)) / 1.5 / 0.02
    assert 120 <= 5 * steps_per_frame <= 190, steps_per_frame     # five frames = one cooldown, within a frame

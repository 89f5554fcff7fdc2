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


# ---------------------------------------------------------------------------------------------------------------------
# Round 4: what the renderings can still pin -- the robot's nose (heading) in both animations and the
# zone fade of the TimedTSP one.  Extracted once by tests/golden/make_gif_heading.py (data only).

def _heading(name):
    import json
    with open(os.path.join(ROOT, "tests", "golden", name)) as fh:
        return json.load(fh)


def _axis_diff(a, b):
    return (a - b + np.pi / 2) % np.pi - np.pi / 2


def test_camera_model_behind_the_heading_fit_reproduces_the_zone_discs():
    """The pinhole camera recovered from the floor (focal length from |r1| = |r2|, metric scale from the zone radius 0.2,
    ZoneEnvBase.py:51) projects a disc of that radius onto every zone blob: widths within 10 %, heights (the foreshortened
    direction, which depends on the recovered tilt) within 20 %."""
    for name, tol_w, tol_h in (("gif_pointtsp_heading.json", 0.08, 0.12), ("gif_timedtsp_heading.json", 0.17, 0.19)):
        d = _heading(name)
        c = np.array(d["zone_disc_check"], float)
        assert len(c) >= 13 and 400 < d["focal_px"] < 560 and 0.8 < d["metres_per_floor_unit"] < 0.9
        assert np.abs(c[:, 0] / c[:, 1] - 1).max() < tol_w and np.abs(c[:, 2] / c[:, 3] - 1).max() < tol_h, name


def test_robot_nose_follows_the_motion_and_what_the_heading_cannot_resolve():
    """The Point robot's silhouette (sphere r = 0.1 + the 'pointarrow' box, xmls/point.xml) fitted per frame.  What the
    fit supports: the nose AXIS lies along the direction of motion (the robot is driven along its body x axis -- the
    motor's gear '0.3 0 0 0 0 0' on the body-fixed site) in the large majority of frames, in both episodes.
    What it does not: the axis is good to about +-25 degrees (median deviation from the track's own direction), its SIGN is
    unreliable when the nose points away from the camera, and a frame is 0.32-0.5 s -- two orders of magnitude above the
    hinge's servo time and of the order of the 0.35 s a full-rate quarter turn takes: the turn-in dynamics
    and the 3 rad/s plateau cannot be read off these pictures, only that nothing contradicts them."""
    for name in ("gif_pointtsp_heading.json", "gif_timedtsp_heading.json"):
        rob = [r for r in _heading(name)["robot"] if r]
        assert len(rob) == 60 and np.median([r["iou"] for r in rob]) > 0.55
        xy = np.array([[r["x"], r["y"]] for r in rob])
        psi = np.array([r["heading"] for r in rob])
        vel = np.diff(xy, axis=0)
        speed = np.hypot(vel[:, 0], vel[:, 1])
        moving = speed > 0.15                                   # metres per frame
        dev = np.degrees(np.abs(_axis_diff(psi[:-1], np.arctan2(vel[:, 1], vel[:, 0]))))[moving]
        assert moving.sum() >= 50 and (dev < 35).mean() > 0.65 and 15 < np.median(dev) < 32, (name, np.median(dev))
    # the model-based position agrees with the blob-centroid track of round 3 (make_gif_track.py) to the robot's own size
    import json
    with open(os.path.join(ROOT, "tests", "golden", "gif_pointtsp_track.json")) as fh:
        tr = json.load(fh)
    d = _heading("gif_pointtsp_heading.json")
    fit = np.array([[r["x"], r["y"]] for r in d["robot"]])
    cen = np.array(tr["robot_xy"]) * d["metres_per_floor_unit"]
    off = fit[:56] - cen[:56]
    assert np.hypot(*(off - off.mean(0)).T).max() < 0.12 and np.hypot(*off.mean(0)) < 0.15


def test_timed_tsp_animation_fades_every_unvisited_zone_at_one_rate_and_visited_zones_stay_yellow():
    """TTSP_env.py:23-27,46-60 as RENDERED by the reference: an unvisited zone's colour is (1 - t, t, t) with
    t = (tmax - steps) / max_steps -- so in the animation every unvisited zone's red channel rises, and its green falls,
    LINEARLY and at the SAME rate for all zones (one clock, `steps`), whatever its own deadline; a visited zone turns
    Yellow (zone_times = 1, :26) and never fades again.  The common rate is the animation's time base: with the zone that
    is still unvisited in the last frame alive throughout, steps per frame < 2000 / 59, and the time base the robot's
    speed plateau gives (terminal speed 1.5 m/s, the normaliser of ZoneEnvBase.py:223) lies below that bound."""
    d = _heading("gif_timedtsp_heading.json")
    zones = d["zones"]
    assert len(zones) == 13                                     # two of the 15 start near t = 0.5: grey on a grey floor
    slopes_r, slopes_g, visit = [], [], []
    for z in zones:
        rgb = np.array(z["rgb"])
        r, g, b = rgb.T
        yellow = (g - b) > 60
        fv = int(np.argmax(yellow)) if yellow.any() else None
        visit.append(fv)
        if fv is not None:
            assert yellow[fv + 1:].all() and np.abs(r[fv + 2:] - g[fv + 2:]).max() < 6      # Yellow from its visit on
        ok = (~yellow) & (r > 150) & (r < 205) & (np.abs(g - b) < 8)                        # inside the display's linear range
        idx = np.nonzero(ok)[0]
        if fv is not None:
            idx = idx[idx < fv - 2]                                                         # (the red robot is on the zone)
        if len(idx) >= 9:
            pr, pg = np.polyfit(idx, r[idx], 1), np.polyfit(idx, g[idx], 1)
            assert np.abs(np.polyval(pr, idx) - r[idx]).max() < 5.0                          # linear in time (palette steps)
            slopes_r.append(pr[0])
            slopes_g.append(pg[0])
    slopes_r, slopes_g = np.array(slopes_r), np.array(slopes_g)
    assert len(slopes_r) >= 10 and (slopes_r > 0).all() and (slopes_g < 0).all()
    assert slopes_r.std() / slopes_r.mean() < 0.08                                           # ONE rate: 1.27 +- 0.07 per frame
    assert sum(v is None for v in visit) == 1 and sorted(v for v in visit if v is not None)[0] >= 3
    # time base: the speed plateau (90th percentile of the per-frame displacement) against the alive bound
    rob = [r for r in d["robot"] if r]
    xy = np.array([[r["x"], r["y"]] for r in rob])
    plateau = np.percentile(np.hypot(*np.diff(xy, axis=0).T), 90)
    steps_per_frame = plateau / (1.5 * 0.02)
    assert 15 < steps_per_frame < 2000 / 59 and 60 * steps_per_frame <= 2000 * 1.02

"""Generate the golden vectors of the RNG / layout half of reset().

The reference's env modules cannot be imported in the build image (gym, safety_gym and
mujoco_py are absent: ordinary ModuleNotFoundError), so these vectors are produced with the
one reference dependency that IS here: numpy's legacy ``RandomState`` (the reference pins
numpy==1.21.1 in requirements.txt:4; the legacy stream is frozen across versions).  The
sampling logic below restates, in plain Python on top of real ``RandomState`` draws,
  * TTSP_env.py:19-21        tmax_i = int(RandomState(seed).beta(3, 1.5) * max_steps)
  * colour_match_env.py:57-68 colour_i = RandomState(seed).choice([Blue, Green, Red])
  * [not vendored] safety_gym Engine.reset/build_layout/sample_layout/draw_placement/random_rot
    with the zone envs' settings (ZoneEnvBase.py:41-53,118-122): RandomState(seed + 1).
Run:  python tests/golden/make_golden.py      (writes tests/golden/reset_vectors.npz)
"""
import os

import numpy as np

EVAL_SEEDS = np.arange(1000000, 1000100)      # main/scripts/evaluate.py:47
EXTENT, ROBOT_KEEPOUT, MARGIN = 3.0, 0.4, 0.0


def placements_dict(num_zones, zones_keepout, robot_locations=(), zones_locations=()):
    """[not vendored] Engine.build_placements_dict / placements_dict_from_object: object -> (placements, keepout);
    an object with a fixed location gets the box (x - k, y - k, x + k, y + k), k = keepout + 1e-9."""
    out = {}
    for name, locs, keepout in ([("robot", robot_locations, ROBOT_KEEPOUT)] +
                                [(f"zone{i}", zones_locations[i:i + 1], zones_keepout) for i in range(num_zones)]):
        if len(locs):
            x, y = locs[0]
            k = keepout + 1e-9
            out[name] = ([(x - k, y - k, x + k, y + k)], keepout)
        else:
            out[name] = (None, keepout)
    return out


def sample_layout(seed, num_zones, zones_keepout, robot_locations=(), zones_locations=(), robot_rot=None):
    rs = np.random.RandomState(seed + 1)       # Engine.reset: _seed += 1
    names = ["robot"] + [f"zone{i}" for i in range(num_zones)]
    placements = placements_dict(num_zones, zones_keepout, robot_locations, zones_locations)
    keepouts = {n: placements[n][1] for n in names}
    restarts = 0
    for _ in range(10000):
        layout = {}
        ok = True
        for name in names:
            k = keepouts[name]
            # draw_placement: constrain_placement(extents or the single fixed box, keepout)
            box = placements[name][0]
            bx0, by0, bx1, by1 = (-EXTENT, -EXTENT, EXTENT, EXTENT) if box is None else box[0]
            xmin, ymin, xmax, ymax = bx0 + k, by0 + k, bx1 - k, by1 - k
            conflicted = True
            for _t in range(100):
                xy = np.array([rs.uniform(xmin, xmax), rs.uniform(ymin, ymax)])
                valid = True
                for other, oxy in layout.items():
                    dist = np.sqrt(np.sum(np.square(xy - oxy)))
                    if dist < keepouts[other] + MARGIN + k:
                        valid = False
                        break
                if valid:
                    conflicted = False
                    break
            if conflicted:
                ok = False
                break
            layout[name] = xy
        if ok:
            break
        restarts += 1
    else:
        raise RuntimeError("Failed to sample layout of objects")
    # build_world_config: robot_rot = random_rot() unless 'robot_rot' is configured
    rot = rs.uniform(0, 2 * np.pi) if robot_rot is None else float(robot_rot)
    zones = np.stack([layout[f"zone{i}"] for i in range(num_zones)])
    return np.r_[layout["robot"], rot], zones, restarts


def main():
    out = {"seeds": EVAL_SEEDS}
    for tag, Z, keepout in (("z15", 15, 0.55), ("z6", 6, 0.55), ("z5", 5, 0.55), ("z25k40", 25, 0.40)):
        robots, zones, restarts = [], [], []
        for s in EVAL_SEEDS:
            r, z, n = sample_layout(int(s), Z, keepout)
            robots.append(r); zones.append(z); restarts.append(n)
        out[f"robot_{tag}"] = np.array(robots)
        out[f"zones_{tag}"] = np.array(zones)
        out[f"restarts_{tag}"] = np.array(restarts, np.int32)
    # TSPHardEnv: config_zone_fixed_1 / _2 (main/envs/__init__.py:52-81)
    hard = {"hard1": dict(robot_locations=[(-0.9, -0.9)], robot_rot=-1,
                          zones_locations=[(-2.6, -1.6), (-0., -0.5), (1., 0.5), (1.8, 1.5), (2.6, 2.6)]),
            "hard2": dict(robot_locations=[(0.8, 0.8)], zones_locations=[(-2.6, -2.6), (-2, -1.6), (2, 1)])}
    for tag, kw in hard.items():
        robots, zones, restarts = [], [], []
        for s in EVAL_SEEDS:
            r, z, n = sample_layout(int(s), 15, 0.55, **kw)
            robots.append(r); zones.append(z); restarts.append(n)
        out[f"robot_{tag}"] = np.array(robots)
        out[f"zones_{tag}"] = np.array(zones)
        out[f"restarts_{tag}"] = np.array(restarts, np.int32)
    tmax15, tmax5, col6 = [], [], []
    for s in EVAL_SEEDS:
        rs = np.random.RandomState(int(s))
        tmax15.append([int(rs.beta(3, 1.5) * 2000) for _ in range(15)])
        rs = np.random.RandomState(int(s))
        tmax5.append([int(rs.beta(3, 1.5) * 1000) for _ in range(5)])
        rs = np.random.RandomState(int(s))
        col6.append([int(rs.choice(3)) for _ in range(6)])
    out["tmax_z15"] = np.array(tmax15, np.int32)
    out["tmax_z5"] = np.array(tmax5, np.int32)
    out["colours_z6"] = np.array(col6, np.int32)
    # FixedSeedsWrapper (wrappers.py:18-21): default_rng(rng_seed).integers(1, 101, size=1)
    fs = {}
    for rng_seed in (0, 1, 10000, 20000, 123456789):
        g = np.random.default_rng(seed=rng_seed)
        fs[rng_seed] = [int(g.integers(low=1, high=101, size=1)[0]) for _ in range(32)]
    out["fixed_seed_rng_seeds"] = np.array(list(fs), np.int64)
    out["fixed_seed_draws"] = np.array(list(fs.values()), np.int64)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reset_vectors.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()

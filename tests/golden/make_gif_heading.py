"""Extract the robot's HEADING (and a model-based position) per frame from the reference's README animations.

``gifs/pointtsp.gif`` and ``gifs/timedtsp.gif`` are renderings of whole episodes by the real MuJoCo 2.0 + safety-gym
stack -- the only outputs of the reference's physics that exist in this setup.  ``make_gif_track.py`` took the robot's
position (centroid of the red blob) from the first; this script fits the robot's SHAPE: the Point robot of
``xmls/point.xml`` is a sphere of radius 0.1 at the body origin plus the "pointarrow" box of half size 0.05 at
(0.1, 0, 0) in the body frame, both red, so the silhouette has a nose and the nose points where the
robot heads.  Data, not source; written once, by hand, in the build container:

    python tests/golden/make_gif_heading.py      (needs PIL + scipy and /root/reference; writes gif_*_heading.json)

Method.  (1) Camera: the floor's corner homography (make_gif_track.py) is, for a pinhole camera with square pixels and the
principal point at the image centre, K [r1 r2 t].  The camera has no roll and looks along the floor's y axis, so of the
two conditions on K only |r1| = |r2| is informative (r1 . r2 = 0 holds for every focal length): it gives the focal length
and with it the projection of points ABOVE the floor.  The metric scale comes from the zone discs (radius 0.2,
ZoneEnvBase.py:51); check: a disc of that radius on the floor projects onto the zone blobs' width and height
(``zone_disc_check``: projected vs measured pixels).  (2) Per frame: the silhouette of sphere + box at (x, y, heading) is
projected (convex hulls of sampled surface points / the box corners), rasterised and compared with the red mask;
(x, y, heading) maximise intersection-over-union -- exhaustive over 5-degree headings and a 5 x 5 grid of positions,
refined to 1 degree / 1 cm.  ``iou`` is kept per frame: the fit is weak where the nose hides behind the sphere.
(3) TimedTSP also: every zone's mean colour per frame (TTSP_env.py:46-60 renders rgba = (1 - t, t, t) for an unvisited
zone with t = (tmax - steps) / 2000, Yellow once visited), so the fade's slope is the animation's time base in env
steps per frame and the frame a zone turns yellow is its visit.
"""
import json
import os

import numpy as np
from PIL import Image, ImageDraw
from scipy import ndimage
from scipy.spatial import ConvexHull

HERE = os.path.dirname(os.path.abspath(__file__))
HALF = 3.5
ZONE_RADIUS = 0.2       # ZoneEnvBase.py:51
R_SPHERE, BOX_HALF, BOX_AT, Z_BODY = 0.1, 0.05, 0.1, 0.1     # point.xml


def _track_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_gif_track", os.path.join(HERE, "make_gif_track.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def camera_from_floor(hm_img_to_ground, scale, size):
    """3 x 4 projection of metric world points from the image -> ground homography (ground in floor units, metres =
    units * scale), assuming square pixels and the principal point at the image centre."""
    cx, cy = size[0] / 2.0, size[1] / 2.0
    g2i = np.linalg.inv(hm_img_to_ground) @ np.diag([1.0 / scale, 1.0 / scale, 1.0])     # metres -> pixels
    shift = np.array([[1, 0, -cx], [0, 1, -cy], [0, 0, 1.0]])
    h = shift @ g2i
    h1, h2 = h[:, 0], h[:, 1]
    f2_b = -((h1[0] ** 2 + h1[1] ** 2) - (h2[0] ** 2 + h2[1] ** 2)) / (h1[2] ** 2 - h2[2] ** 2)
    f = float(np.sqrt(f2_b))
    kinv = np.diag([1 / f, 1 / f, 1.0])
    m = kinv @ h
    lam = 1.0 / np.linalg.norm(m[:, 0])
    r1, r2, t = m[:, 0] * lam, m[:, 1] * lam, m[:, 2] * lam
    r3 = np.cross(r1, r2)
    if t[2] < 0:                                   # the floor is in front of the camera
        r1, r2, t, r3 = -r1, -r2, -t, np.cross(-r1, -r2)
    k = np.array([[f, 0, cx], [0, f, cy], [0, 0, 1.0]])
    p = k @ np.c_[r1, r2, r3, t]
    return p, f


def project(p, xyz):
    q = np.c_[np.atleast_2d(xyz), np.ones(len(np.atleast_2d(xyz)))] @ p.T
    return q[:, :2] / q[:, 2:]


_SPHERE = None


def sphere_points():
    global _SPHERE
    if _SPHERE is None:
        rs = np.random.RandomState(0)
        v = rs.normal(size=(600, 3))
        _SPHERE = v / np.linalg.norm(v, axis=1, keepdims=True) * R_SPHERE
    return _SPHERE


def silhouette(p, x, y, psi, x0, y0, w, h, ss=3):
    """Binary mask (h x w crop at (x0, y0)) of the projected sphere + nose box at heading psi, supersampled ss x."""
    img = Image.new("L", (w * ss, h * ss), 0)
    dr = ImageDraw.Draw(img)
    c = np.array([x, y, Z_BODY])
    for pts in (sphere_points() + c,
                np.array([[BOX_AT + sx * BOX_HALF, sy * BOX_HALF, sz * BOX_HALF] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)])
                @ np.array([[np.cos(psi), np.sin(psi), 0], [-np.sin(psi), np.cos(psi), 0], [0, 0, 1.0]]) + c):
        uv = project(p, pts)
        hull = uv[ConvexHull(uv).vertices]
        dr.polygon([((u - x0) * ss, (v - y0) * ss) for u, v in hull], fill=255)
    a = np.asarray(img.resize((w, h), Image.BOX)).astype(float) / 255.0
    return a


def fit_robot(p, red, guess_xy):
    """(x, y, psi, iou) of the best sphere + nose silhouette against the boolean red mask."""
    yy, xx = np.nonzero(red)
    if len(xx) < 20:
        return None
    x0, y0 = int(xx.min()) - 6, int(yy.min()) - 6
    w, h = int(xx.max()) - x0 + 7, int(yy.max()) - y0 + 7
    target = red[y0:y0 + h, x0:x0 + w].astype(float)

    def score(x, y, psi):
        m = silhouette(p, x, y, psi, x0, y0, w, h)
        inter = np.minimum(m, target).sum()
        return inter / (np.maximum(m, target).sum() + 1e-9)
    best = (-1.0, guess_xy[0], guess_xy[1], 0.0)
    for dx in np.linspace(-0.08, 0.08, 5):
        for dy in np.linspace(-0.08, 0.08, 5):
            for psi in np.radians(np.arange(0, 360, 10)):
                s = score(guess_xy[0] + dx, guess_xy[1] + dy, psi)
                if s > best[0]:
                    best = (s, guess_xy[0] + dx, guess_xy[1] + dy, psi)
    for step_xy, step_psi in ((0.02, np.radians(4)), (0.01, np.radians(2)), (0.005, np.radians(1))):
        improved = True
        while improved:
            improved = False
            s0, x, y, psi = best
            for dx, dy, dp in ((step_xy, 0, 0), (-step_xy, 0, 0), (0, step_xy, 0), (0, -step_xy, 0), (0, 0, step_psi), (0, 0, -step_psi)):
                s = score(x + dx, y + dy, psi + dp)
                if s > best[0] + 1e-6:
                    best, improved = (s, x + dx, y + dy, psi + dp), True
    s, x, y, psi = best
    # how much better than the best heading-free explanation: the same fit with the nose turned by 180 degrees
    return x, y, (psi + np.pi) % (2 * np.pi) - np.pi, s, s - score(x, y, psi + np.pi)


def red_mask(r, g, b):
    return (r > 110) & (g < 75) & (b < 75)


def floor_and_scale(track, im):
    r, g, b = track.frame(im, 0)
    grey, corners = track.floor_corners(r, g, b)
    hm = track.homography(corners, [(-HALF, HALF), (HALF, HALF), (HALF, -HALF), (-HALF, -HALF)])
    return hm, corners


def zone_discs(track, hm, r, g, b, min_px=60):
    """Coloured discs on the floor (not grey, not background, not the robot): centre and diameter in floor units."""
    grey = (abs(r - g) < 14) & (abs(g - b) < 14) & (r > 130)
    rows = np.nonzero(grey.any(1))[0]
    col = ((abs(r - g) > 12) | (abs(g - b) > 12)) & ~((r < 90) & (g < 90))
    col[:rows.min()] = False
    col[rows.max():] = False
    lab, n = ndimage.label(col)
    out = []
    for i in range(1, n + 1):
        m = lab == i
        if m.sum() < min_px or (r[m].mean() > 190 and g[m].mean() < 80):
            continue
        yy, xx = np.nonzero(m)
        c = track.to_ground(hm, [(xx.mean(), yy.mean())])[0]
        l_ = track.to_ground(hm, [(xx.min() - 0.5, yy[xx == xx.min()].mean())])[0]
        r_ = track.to_ground(hm, [(xx.max() + 0.5, yy[xx == xx.max()].mean())])[0]
        out.append({"xy": c, "diam": float(r_[0] - l_[0]), "mask": ndimage.binary_erosion(m, iterations=2)})
    return out


def run(gif, name, with_zones):
    track = _track_module()
    im = Image.open(gif)
    hm, corners = floor_and_scale(track, im)
    r, g, b = track.frame(im, 0)
    discs = zone_discs(track, hm, r, g, b)
    scale = 2 * ZONE_RADIUS / float(np.median([d["diam"] for d in discs]))          # metres per floor unit
    p, f = camera_from_floor(hm, scale, im.size)
    # check of the camera: a disc of radius 0.2 on the floor projects onto the zone blobs (centres by construction; sizes)
    disc_check = []
    ang = np.linspace(0, 2 * np.pi, 90)
    for d in discs:
        c = np.array(d["xy"]) * scale
        uv = project(p, np.c_[c[0] + ZONE_RADIUS * np.cos(ang), c[1] + ZONE_RADIUS * np.sin(ang), np.zeros(len(ang))])
        yy, xx = np.nonzero(d["mask"])            # eroded by 2 px on every side
        disc_check.append([round(float(uv[:, 0].max() - uv[:, 0].min()), 1), int(xx.max() - xx.min() + 5),
                           round(float(uv[:, 1].max() - uv[:, 1].min()), 1), int(yy.max() - yy.min() + 5)])
    frames = []
    for fidx in range(im.n_frames):
        r, g, b = track.frame(im, fidx)
        red = red_mask(r, g, b)
        yy, xx = np.nonzero(red)
        if len(xx) < 20:
            frames.append(None)
            continue
        lab, n = ndimage.label(red)
        if n > 1:                                   # keep the largest red component (a fully red zone is red too)
            sizes = ndimage.sum(red, lab, range(1, n + 1))
            red = lab == (1 + int(np.argmax(sizes)))
            yy, xx = np.nonzero(red)
        guess = track.to_ground(hm, [(xx.mean(), yy.mean())])[0] * scale
        # the blob's centroid is the sphere's centre seen 0.1 above the floor: slide the guess along the view direction
        fit = fit_robot(p, red, guess)
        frames.append(fit)
    out = {
        "source": f"reference {os.path.relpath(gif, '/root/reference')} ({im.size[0]} x {im.size[1]}, {im.n_frames} frames of "
                  f"{im.info.get('duration')} ms)",
        "metres_per_floor_unit": round(scale, 5), "focal_px": round(f, 1),
        "zone_disc_check": disc_check,
        "robot": [None if fr is None else {"x": round(fr[0], 4), "y": round(fr[1], 4), "heading": round(float(fr[2]), 4),
                                            "iou": round(float(fr[3]), 3), "iou_gain_over_reversed": round(float(fr[4]), 3)}
                  for fr in frames],
    }
    if with_zones:
        zs = []
        for d in discs:
            series = []
            for fidx in range(im.n_frames):
                r, g, b = track.frame(im, fidx)
                m = d["mask"]
                series.append([round(float(r[m].mean()), 1), round(float(g[m].mean()), 1), round(float(b[m].mean()), 1)])
            zs.append({"xy_m": [round(float(v) * scale, 4) for v in d["xy"]], "rgb": series})
        out["zones"] = zs
    path = os.path.join(HERE, name)
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1)
    ok = [fr for fr in frames if fr is not None]
    print(path, "frames fitted", len(ok), "median IoU %.2f" % np.median([fr[3] for fr in ok]), "focal", f, "scale", scale)


if __name__ == "__main__":
    run("/root/reference/gifs/pointtsp.gif", "gif_pointtsp_heading.json", False)
    run("/root/reference/gifs/timedtsp.gif", "gif_timedtsp_heading.json", True)

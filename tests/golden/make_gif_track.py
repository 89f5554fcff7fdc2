"""Extract the robot track and the zone layout from the reference's README animation of PointTSP.

``/root/reference/gifs/pointtsp.gif`` (README.md:3-9 of the reference) is a rendering of one whole PointTSP episode by
the real MuJoCo 2.0 + safety-gym stack: 15 zones visited in 56 frames, then the reset.  It is the only output of the
reference's physics that exists in this setup, so DESIGN.md section 0.2 uses it as evidence for the one model constant
the oracle cannot take from the tree (the geom density of point.xml, i.e. the robot's time constant m / b).  This
script turns the picture into numbers (data, not source): written once, by hand, in the build container --

    python tests/golden/make_gif_track.py        (needs PIL + scipy and /root/reference; writes gif_pointtsp_track.json)

Method: the floor is a square checker of 20 x 20 tiles seen by a fixed camera without roll; its four corners (lines
fitted to the grey/background boundary) give the image -> ground homography, checked against the 19 + 19 interior tile
boundaries of one image row and one column (they land on integers to +-0.05 tile).  Ground coordinates are written in
units in which the floor's half width is 3.5; the metric scale is left to the user of the file, who has the zone discs
(radius 0.2, ZoneEnvBase.py:51) to calibrate it: ``zone_diameter`` holds their measured size in the same units.
Robot = centroid of the red blob (its centre is 0.1 above the ground: a constant offset of about 0.1 away from the
camera, irrelevant for displacements); visited zones = yellow blobs.
"""
import json
import os

import numpy as np
from PIL import Image
from scipy import ndimage

GIF = "/root/reference/gifs/pointtsp.gif"
HALF = 3.5


def homography(src, dst):
    rows = []
    for (x, y), (u, v) in zip(src, dst):
        rows.append([x, y, 1, 0, 0, 0, -u * x, -u * y, -u])
        rows.append([0, 0, 0, x, y, 1, -v * x, -v * y, -v])
    return np.linalg.svd(np.array(rows, float))[2][-1].reshape(3, 3)


def to_ground(hm, pts):
    pts = np.atleast_2d(np.asarray(pts, float))
    q = np.c_[pts, np.ones(len(pts))] @ hm.T
    return q[:, :2] / q[:, 2:]


def frame(im, f):
    im.seek(f)
    a = np.asarray(im.convert("RGB")).astype(int)
    return a[..., 0], a[..., 1], a[..., 2]


def floor_corners(r, g, b):
    grey = (abs(r - g) < 14) & (abs(g - b) < 14) & (r > 130)
    rows = np.nonzero(grey.any(1))[0]
    top, bottom = rows.min() - 0.5, rows.max() + 0.5
    ys = np.arange(rows.min() + 5, rows.max() - 4)
    left = np.polyfit(ys, [np.nonzero(grey[y])[0].min() - 0.5 for y in ys], 1)
    right = np.polyfit(ys, [np.nonzero(grey[y])[0].max() + 0.5 for y in ys], 1)
    corners = [(np.polyval(left, top), top), (np.polyval(right, top), top),
               (np.polyval(right, bottom), bottom), (np.polyval(left, bottom), bottom)]
    return grey, corners


def tile_boundaries(values):
    bits = (values > np.median(values)).astype(int)
    return np.nonzero(np.diff(bits))[0] + 0.5


def main():
    im = Image.open(GIF)
    r, g, b = frame(im, 0)
    grey, corners = floor_corners(r, g, b)
    hm = homography(corners, [(-HALF, HALF), (HALF, HALF), (HALF, -HALF), (-HALF, -HALF)])
    # check: interior tile boundaries of one zone-free column and one row, in tiles from the floor's edge
    x0, y0 = 230, 300
    ys = np.nonzero(grey[:, x0])[0]
    col = to_ground(hm, [(x0, ys.min() + t + 0.5) for t in tile_boundaries(r[ys.min():ys.max() + 1, x0])])
    xs = np.nonzero(grey[y0])[0]
    row = to_ground(hm, [(xs.min() + t + 0.5, y0) for t in tile_boundaries(r[y0, xs.min():xs.max() + 1])])
    tiles_col = (HALF - col[:, 1]) / (2 * HALF / 20)
    tiles_row = (row[:, 0] + HALF) / (2 * HALF / 20)
    # zones of frame 0 (all cyan)
    lab, n = ndimage.label((b > 180) & (g > 180) & (r < 170))
    zones, diam = [], []
    for i in range(1, n + 1):
        yy, xx = np.nonzero(lab == i)
        if len(xx) < 30:
            continue
        zones.append(to_ground(hm, [(xx.mean(), yy.mean())])[0])
        l_ = to_ground(hm, [(xx.min() - 0.5, yy[xx == xx.min()].mean())])[0]
        r_ = to_ground(hm, [(xx.max() + 0.5, yy[xx == xx.max()].mean())])[0]
        t_ = to_ground(hm, [(xx[yy == yy.min()].mean(), yy.min() - 0.5)])[0]
        b_ = to_ground(hm, [(xx[yy == yy.max()].mean(), yy.max() + 0.5)])[0]
        diam.append(0.5 * ((r_[0] - l_[0]) + (t_[1] - b_[1])))
    robot, visited = [], []
    for f in range(im.n_frames):
        r, g, b = frame(im, f)
        yy, xx = np.nonzero((r > 150) & (g < 60) & (b < 60))
        robot.append(to_ground(hm, [(xx.mean(), yy.mean())])[0])
        lab, n = ndimage.label((r > 180) & (g > 180) & (b < 120))
        visited.append(int(sum(1 for i in range(1, n + 1) if (lab == i).sum() > 30)))
    out = {
        "source": "reference gifs/pointtsp.gif (600 x 332, 60 frames of 100 ms)",
        "floor_half_units": HALF,
        "floor_corners_px": np.round(np.array(corners), 3).tolist(),
        "tile_check_column": np.round(tiles_col, 3).tolist(),
        "tile_check_row": np.round(tiles_row, 3).tolist(),
        "zones_xy": np.round(np.array(zones), 4).tolist(),
        "zone_diameter": np.round(np.array(diam), 4).tolist(),
        "robot_xy": np.round(np.array(robot), 4).tolist(),
        "visited_count": visited,
    }
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gif_pointtsp_track.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1)
    print(path, "zones", len(zones), "mean zone diameter", np.mean(diam), "frames", len(robot))
    print("tile check (column):", np.round(tiles_col, 2))
    print("tile check (row):   ", np.round(tiles_row, 2))


if __name__ == "__main__":
    main()

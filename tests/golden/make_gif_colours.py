"""Extract the zone colour changes from the reference's README animation of ColourMatch.

``/root/reference/gifs/colourmatch.gif`` shows three clips of ColourMatch-v0 episodes rendered by the real stack.  A zone
is a translucent disc in one of three colours; every visit outside its cooldown turns it to the next colour of the cycle
(colour_match_env.py:106-120).  This script follows each disc over the frames (same homography method as
make_gif_track.py) and writes every colour change it sees -- data, not source:

    python tests/golden/make_gif_colours.py        (needs PIL + scipy and /root/reference; writes gif_colourmatch_changes.json)

tests/test_gif_evidence.py holds the oracle's cycle (Blue -> Green -> Red -> Blue) and its cooldown against them.
"""
import json
import os

import numpy as np
from PIL import Image
from scipy import ndimage

from make_gif_track import HALF, floor_corners, frame, homography, to_ground

GIF = "/root/reference/gifs/colourmatch.gif"


def zones_of(im, hm, f):
    r, g, b = frame(im, f)
    sat = (np.maximum(np.maximum(r, g), b) - np.minimum(np.minimum(r, g), b)) > 40
    robot = (r > 150) & (g < 60) & (b < 60)
    lab, n = ndimage.label(sat & ~robot)
    out = []
    for i in range(1, n + 1):
        yy, xx = np.nonzero(lab == i)
        if not 60 <= len(xx) <= 5000:                      # specks, and the background
            continue
        c = to_ground(hm, [(xx.mean(), yy.mean())])[0]
        if max(abs(c[0]), abs(c[1])) > HALF - 0.1:
            continue
        col = [r[yy, xx].mean(), g[yy, xx].mean(), b[yy, xx].mean()]
        out.append((float(c[0]), float(c[1]), "RGB"[int(np.argmax(col))]))
    ys, xs = np.nonzero(robot)
    rob = to_ground(hm, [(xs.mean(), ys.mean())])[0] if len(xs) else np.array([np.nan, np.nan])
    return out, rob


def main():
    im = Image.open(GIF)
    grey, corners = floor_corners(*frame(im, 0))
    hm = homography(corners, [(-HALF, HALF), (HALF, HALF), (HALF, -HALF), (-HALF, -HALF)])
    changes, robot, clip, prev = [], [], 0, None
    for f in range(im.n_frames):
        zones, rob = zones_of(im, hm, f)
        robot.append([round(float(rob[0]), 3), round(float(rob[1]), 3)])
        if prev is not None:
            # the same layout as the frame before?  (a new clip / episode moves every disc)
            match = [min(prev, key=lambda p: (p[0] - z[0]) ** 2 + (p[1] - z[1]) ** 2) for z in zones]
            same = len(zones) == len(prev) and all(abs(m[0] - z[0]) + abs(m[1] - z[1]) < 0.25 for m, z in zip(match, zones))
            if not same:
                clip += 1
            else:
                for m, z in zip(match, zones):
                    if m[2] != z[2]:
                        changes.append({"frame": f, "clip": clip, "zone_xy": [round(z[0], 2), round(z[1], 2)],
                                        "from": m[2], "to": z[2]})
        prev = zones
    out = {"source": "reference gifs/colourmatch.gif (600 x 343, 66 frames of 100 ms)", "floor_half_units": HALF,
           "changes": changes, "robot_xy": robot}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gif_colourmatch_changes.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1)
    for c in changes:
        print(c)
    print(path, len(changes), "changes in", clip + 1, "clips")


if __name__ == "__main__":
    main()

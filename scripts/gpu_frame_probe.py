"""Draws one frame of the benchmark's orbit at 3840x2160 on the GPU and by the oracle (a band of rows, shaded from the
device's IBL maps) and lists every pixel that differs by more than 2/255 with what both sides hold around it.
Usage: python scripts/gpu_frame_probe.py FRAME Y0 Y1"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_utils as pu   # noqa: E402

frame, y0, y1 = (int(a) for a in sys.argv[1:4])
s = pu.SmallScene.build(width=3840, height=2160, hdr_size=(4096, 2048), sky_size=1024, pre_size=128, pre_mips=5, lut_size=512, irr_size=32)
r = s.renderer
for _ in range(frame):
    r.camera.update_camera_rotation(-10.0, 0.0)
rgba, vis = r.draw_frame(want_vis=True)
with s.oracle.with_maps(s.gpu_maps()):
    orgba, ovis, margin = s.render_oracle_with_margins(y0, y1)
d = np.abs(rgba[y0:y1].astype(np.int32) - orgba.astype(np.int32)).max(-1)
print("vis equal", bool((vis[y0:y1] == ovis).all()), "pixels beyond 2:", int((d > 2).sum()), "beyond 1:", int((d > 1).sum()))
for y, x in zip(*np.nonzero(d > 2)):
    print("pixel", x, y + y0, "gpu", rgba[y + y0, x], "oracle", orgba[y, x], "margin", margin[y, x], "id", vis[y + y0, x])
    ya, yb, xa, xb = max(0, y - 2), y + 3, max(0, x - 2), x + 3
    print(" ids\n", vis[y0 + ya:y0 + yb, xa:xb])
    print(" gpu r\n", rgba[y0 + ya:y0 + yb, xa:xb, 0], "\n oracle r\n", orgba[ya:yb, xa:xb, 0])
    print(" gpu g\n", rgba[y0 + ya:y0 + yb, xa:xb, 1], "\n oracle g\n", orgba[ya:yb, xa:xb, 1])
r.close()

"""Lists the pixels of an orbit sweep where the GPU frame differs from the oracle's by more than 2/255."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import parity_utils as pu
s = pu.SmallScene.build(width=480, height=270, sky_size=256, pre_size=128, pre_mips=5, lut_size=128, irr_size=16)
cam = s.pkg.Camera(); s.renderer.camera = cam
for frame in range(0, 360, 10):
    rgba, vis = s.render_gpu(); orgba, ovis = s.render_oracle()
    d = np.abs(rgba.astype(int) - orgba.astype(int)).max(-1)
    ys, xs = np.nonzero(d > 2)
    for y, x in zip(ys, xs):
        q = vis[y & ~1:(y & ~1) + 2, x & ~1:(x & ~1) + 2].reshape(-1)
        print(f"frame {frame} pixel ({x},{y}) gpu {rgba[y, x, :3]} oracle {orgba[y, x, :3]} id {vis[y, x]} quad ids {q}", flush=True)
    for _ in range(10):
        cam.update_camera_rotation(-10.0, 0.0)

"""Generates the committed golden fixtures under tests/golden/ from the CPU oracle.

The reference has no tests, fixtures or runnable implementation of this path, so these vectors pin
the oracle against itself over time (regressions) and give the GPU tests a second, committed
target.  Inputs: the DamagedHelmet asset under assets/ and synthetic_sky(2048, 1024, seed 355).
Run from the repository root:  python scripts/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_utils as pu  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
IBL_CFG = dict(hdr_size=(2048, 1024), sky_size=8, pre_size=16, pre_mips=3, lut_size=32, irr_size=4)
FRAME_CFG = dict(width=160, height=90, **IBL_CFG)
VIEWS = {"default": [], "orbit": [("rotate", -400.0, 0.0), ("rotate", 0.0, 150.0), ("dolly", 120.0, 0.0)]}


def apply_view(camera, moves):
    for kind, dx, dy in moves:
        if kind == "rotate":
            camera.update_camera_rotation(dx, dy)
        elif kind == "dolly":
            camera.update_camera_distance(dx, dy)


def main():
    os.makedirs(GOLD, exist_ok=True)
    s = pu.SmallScene.build(gpu=False, **FRAME_CFG)
    m = s.oracle.maps()
    np.savez_compressed(os.path.join(GOLD, "ibl_small.npz"), sky=m["sky"], irr=m["irr"], lut=m["lut"],
                        **{f"pre{i}": p for i, p in enumerate(m["pre"])})
    out = {}
    for name, moves in VIEWS.items():
        s.renderer.camera = s.pkg.Camera()
        apply_view(s.renderer.camera, moves)
        rgba, vis = s.render_oracle()
        out[f"{name}_rgba"] = rgba
        out[f"{name}_vis"] = vis
    np.savez_compressed(os.path.join(GOLD, "frame_160x90.npz"), **out)
    for f in sorted(os.listdir(GOLD)):
        print(f, os.path.getsize(os.path.join(GOLD, f)))


if __name__ == "__main__":
    main()

"""Prints how the covered pixels of 4K orbit frames split into uniform and mixed 2x2 quads (PbrFrameStats).
Usage: python scripts/gpu_quad_split.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_utils as pu
s = pu.SmallScene.build(width=3840, height=2160, hdr_size=(4096, 2048), sky_size=1024, pre_size=128, pre_mips=5, lut_size=512, irr_size=32)
r = s.renderer
for f in (0, 90, 180):
    rgba, vis, st = r.draw_frame(want_vis=True, want_stats=True)
    print("frame", f, "covered", st.covered_pixels, "uniform quads", st.uniform_quads, "mixed quads", st.mixed_quads, "setup tris", st.setup_triangles)
    for _ in range(90):
        r.camera.update_camera_rotation(-10.0, 0.0)
r.close()

"""Whole 4K frames of the orbit against the oracle: lists the pixels beyond 2/255 with their decision margins.
python scripts/gpu_whole_frame_diff.py [turns ...]  ->  gpurun_out/whole_frame_diff.json"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_utils as pu
stress = "stress" in sys.argv[1:]
W, H = (7680, 4320) if stress else (3840, 2160)
if stress:
    import bench
    pkg = pu._pkg()
    g = pkg.glTFLoader(); g.load(pkg.renderer.DEFAULT_MODEL)
    s = pu.SmallScene.build(width=W, height=H, instance_models=bench.stress_instances(pkg, g.get_geometries()[0].model))
else:
    s = pu.SmallScene.build(width=W, height=H, hdr_size=(4096, 2048), sky_size=1024, pre_size=128, pre_mips=5, lut_size=512, irr_size=32)
r = s.renderer
out = {}
for turns in [int(a) for a in sys.argv[1:] if a != "stress"] or [0]:
    r.camera = s.pkg.Camera()
    if stress:
        r.camera.update_camera_distance((5.0 - 30.0) * 100.0, 0.0)
    for _ in range(turns):
        r.camera.update_camera_rotation(-10.0, 0.0)
    u = r.frame_uniforms()
    rgba, vis = r.draw_frame(want_vis=True)
    ulps = float(os.environ.get("TANGENT_ULPS", "4"))
    orgba, ovis, margin, lo, hi = s.oracle.draw_with_bounds(u, W, H, 0, H, tangent_ulps=ulps)
    d = np.abs(rgba.astype(np.int32) - orgba.astype(np.int32)).max(-1)
    g = rgba.astype(np.int32)
    outside = ((g < lo.astype(np.int32) - 2) | (g > hi.astype(np.int32) + 2)).any(-1)
    print(turns, "beyond 2/255 of the nominal frame:", int((d > 2).sum()), "; outside the tangent interval:", int(outside.sum()), "; of those without a sampling decision nearby:",
          int((outside & (margin >= 2e-3)).sum()), "; pixels whose interval is wider than 2:", int(((hi.astype(np.int32) - lo) > 2).any(-1).sum()), flush=True)
    far = outside
    ys, xs = np.nonzero(far)
    rows = [{"x": int(x), "y": int(y), "d": int(d[y, x]), "margin": float(margin[y, x]), "vis": int(vis[y, x]), "ovis": int(ovis[y, x]), "gpu": rgba[y, x].tolist(), "oracle": orgba[y, x].tolist(), "lo": lo[y, x].tolist(), "hi": hi[y, x].tolist()} for y, x in zip(ys, xs)]
    rows.sort(key=lambda e: -e["margin"])
    out[str(turns)] = {"vis_equal": bool((vis == ovis).all()), "far": len(rows), "pixels": int(d.size), "within_1": float((d <= 1).mean()), "worst": rows[:40],
                       "margins_of_far": sorted(float(m) for m in margin[far])[-20:]}
    print(turns, "vis equal", out[str(turns)]["vis_equal"], "far", len(rows), "of", d.size, "largest margins", out[str(turns)]["margins_of_far"][-5:], flush=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "whole_frame_diff.json"), "w"), indent=1)
r.close()

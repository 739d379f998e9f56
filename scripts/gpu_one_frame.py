"""Renders a few frames (4K unless a size is given) (for profilers) and prints the per-stage event times of the last ones:
python scripts/gpu_one_frame.py [distance] [frames] [width] [height]
python scripts/gpu_one_frame.py stress [frames]      the 8K frame of 10 x 10 helmets"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
pkg = entry.load_package()
stress = len(sys.argv) > 1 and sys.argv[1] == "stress"
dist = 30.0 if stress else float(sys.argv[1]) if len(sys.argv) > 1 else 5.0
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 3
width = int(sys.argv[3]) if len(sys.argv) > 3 else 3840
height = int(sys.argv[4]) if len(sys.argv) > 4 else 2160
if stress:
    import bench
    g = pkg.glTFLoader(); g.load(pkg.renderer.DEFAULT_MODEL)
    r = pkg.Renderer(width=7680, height=4320, skybox_size=1024, instance_models=bench.stress_instances(pkg, g.get_geometries()[0].model))
else:
    r = pkg.Renderer(width=width, height=height, skybox_size=1024)
r.init()
r.camera.update_camera_distance((5.0 - dist) * 100.0, 0.0)
acc = {}
for i in range(frames):
    _, st = r.draw_frame(want_stats=True)
    if i >= frames // 2:
        for name, _ in st._fields_:
            if name.startswith("ms_"):
                acc.setdefault(name, []).append(getattr(st, name))
print({k: round(sum(v) / len(v) * 1000, 2) for k, v in acc.items()}, "us; covered", st.covered_pixels)

"""4K frame time for the reference's GUI-selectable render modes (sampler matrix, mipmaps, MSAA):
python scripts/gpu_sampler_modes.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as entry
pkg = entry.load_package()
r = pkg.Renderer(width=3840, height=2160, skybox_size=1024)
r.init()
color = torch.empty((2160, 3840), dtype=torch.int32, device="cuda:0")
modes = [("default: repeat, linear, mipmaps", {}, False), ("clamp-to-edge", {"addressModeU": "clamp-to-edge", "addressModeV": "clamp-to-edge"}, False),
         ("nearest mag + min", {"magFilter": "nearest", "minFilter": "nearest"}, False), ("mipmaps off", {"mipmaps": False}, False),
         ("default + MSAA", {}, True), ("nearest + MSAA", {"magFilter": "nearest", "minFilter": "nearest"}, True)]
base = dict(r.texture_options)
for name, opts, msaa in modes:
    r.texture_options.update(base); r.texture_options.update(opts); r.msaa_options["enable"] = msaa
    u = r.frame_uniforms()
    for _ in range(5):
        r.draw_frame_device(color.data_ptr(), uniforms=u)
    r.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        r.draw_frame_device(color.data_ptr(), uniforms=u)
    r.synchronize()
    print(f"{name:34s} sampler {u.sampler_index:2d} mipmaps {u.use_mipmaps}: {(time.perf_counter() - t0) / 100 * 1e6:7.1f} us per frame", flush=True)

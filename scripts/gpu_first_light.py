"""Diagnostic run on the GPU box: IBL maps and one frame, GPU vs oracle, with images saved."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_utils as pu  # noqa: E402

out_dir = os.path.join(ROOT, "gpurun_out")
os.makedirs(out_dir, exist_ok=True)
W, H = int(os.environ.get("W", 480)), int(os.environ.get("H", 270))
t = time.time()
s = pu.SmallScene.build(width=W, height=H, sky_size=256, pre_size=128, pre_mips=5, lut_size=128, irr_size=16)
print("gpu scene built", time.time() - t, flush=True)
lib = s.pkg._lib
r = s.renderer
m = s.oracle.maps()
print("oracle maps built", time.time() - t, flush=True)


def rel_err(a, b):
    a = a.astype(np.float64); b = b.astype(np.float64)
    d = np.abs(a - b) / np.maximum(np.abs(b), 1e-6)
    return d


sky = r.download_map(lib.MAP_SKYBOX)
print("sky exact frac", float((sky == m["sky"]).all(axis=-1).mean()))
irr = r.download_map(lib.MAP_IRRADIANCE)
e = rel_err(irr[..., :3], m["irr"][..., :3]); print("irr rel err max/mean", e.max(), e.mean())
for lvl in range(5):
    pre = r.download_map(lib.MAP_PREFILTER, lvl)
    e = rel_err(pre[..., :3], m["pre"][lvl][..., :3])
    print(f"pre L{lvl} rel err max {e.max():.3e} mean {e.mean():.3e} frac>1e-3 {float((e > 1e-3).mean()):.5f}")
lut = r.download_map(lib.MAP_BRDF_LUT).view(np.float16).astype(np.float32)
ref = m["lut"].view(np.float16).astype(np.float32)
nan_same = bool((np.isnan(lut) == np.isnan(ref)).all())
ok = ~np.isnan(ref)
print("lut nan pattern same", nan_same, "max abs diff", float(np.abs(lut[ok] - ref[ok]).max()), "bit-equal frac", float((lut[ok] == ref[ok]).mean()))

t = time.time()
rgba, vis, st = r.draw_frame(want_vis=True, want_stats=True)
print("gpu frame", time.time() - t, "stats ms", st.ms_vertex, st.ms_setup, st.ms_raster, st.ms_resolve, st.ms_total, "covered", st.covered_pixels,
      "setup", st.setup_triangles, "refs", st.binned_refs, flush=True)
t = time.time()
orgba, ovis = s.render_oracle()
print("oracle frame", time.time() - t)
print("vis equal frac", float((vis == ovis).mean()), "mismatch", int((vis != ovis).sum()))
d = np.abs(rgba.astype(np.int32) - orgba.astype(np.int32)).max(axis=-1)
print("rgba max diff", int(d.max()), "frac<=2", float((d <= 2).mean()), "frac==0", float((d == 0).mean()), "psnr", pu.psnr(rgba, orgba))
cov = ovis > 0
print("covered: frac<=2", float((d[cov] <= 2).mean()) if cov.any() else None, "sky: frac<=2", float((d[~cov] <= 2).mean()))
from PIL import Image  # noqa: E402
Image.fromarray(rgba).save(os.path.join(out_dir, "gpu_frame.png"))
Image.fromarray(orgba).save(os.path.join(out_dir, "oracle_frame.png"))
Image.fromarray((np.minimum(d, 25) * 10).astype(np.uint8)).save(os.path.join(out_dir, "diff_frame.png"))

import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_utils as pu, oracle_binding as ob
pkg = pu._pkg(); lib = pkg._lib
r = pkg.Renderer(width=64, height=64); r._open()
lib.check(r._ctx, r._lib.pbr_ibl_brdf_lut(r._ctx, 512))
lut = r.download_map(lib.MAP_BRDF_LUT).view(np.float16).astype(np.float32)
f32, f16 = ob.brdf_lut(512)
ref = f16.view(np.float16).astype(np.float32)
d = np.abs(lut - ref); d[np.isnan(d)] = 0
idx = np.argsort(d.reshape(-1))[::-1][:12]
for i in idx:
    y, x, c = np.unravel_index(i, d.shape)
    print(y, x, c, lut[y, x, c], ref[y, x, c], f32[y, x, c], d[y, x, c])
print("frac nonzero", (d > 0).mean(), "count >1ulp-ish", (d > 1e-3 * np.maximum(np.abs(ref), 1e-3)).sum())

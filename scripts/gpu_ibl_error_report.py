"""Prints, per IBL map, the share of texels whose relative error against the oracle exceeds 1e-3 and the
maximum error (reference sizes, synthetic sky)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_utils as pu
scene = pu.SmallScene.build(width=320, height=180, hdr_size=(4096, 2048), sky_size=64, pre_size=128, pre_mips=5, lut_size=64, irr_size=32)
lib, r, m = scene.pkg._lib, scene.renderer, scene.oracle.maps()
def rel(got, ref):
    ref = ref.astype(np.float64); got = got.astype(np.float64)
    return np.abs(got - ref) / np.maximum(np.abs(ref), 1e-3 * np.abs(ref).mean())
e = rel(r.download_map(lib.MAP_IRRADIANCE)[..., :3], m["irr"][..., :3]); print("irradiance", (e > 1e-3).mean(), e.max())
for lvl in range(5):
    e = rel(r.download_map(lib.MAP_PREFILTER, lvl)[..., :3], m["pre"][lvl][..., :3])
    print("prefilter level", lvl, "share>1e-3 %.6f share>1e-2 %.6f max %.3f" % ((e > 1e-3).mean(), (e > 1e-2).mean(), e.max()))

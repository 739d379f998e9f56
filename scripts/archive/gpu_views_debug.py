"""Debug aid: multi-view passes against single frames at one size; prints where they differ."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np
import parity_utils as pu
import test_gpu_views as tv
def parse(x):   # WxH or WxH:y0-y1
    size, _, band = x.partition(":")
    w, h = (int(v) for v in size.split("x"))
    y0, y1 = (int(v) for v in band.split("-")) if band else (0, h)
    return w, h, y0, y1
sizes = [parse(x) for x in sys.argv[1].split(",")]
order = [int(x) for x in sys.argv[2].split(",")]
scene = pu.SmallScene.build(width=sizes[0][0], height=sizes[0][1], sky_size=128, pre_size=32, pre_mips=4, lut_size=64, irr_size=8)
r = scene.renderer
for (w, h, y0, y1), n in [(sz, n) for sz in sizes for n in order]:
    r.width, r.height = w, h
    us = tv._orbit_uniforms(scene, n)
    want = [tv._single(scene, u, w, h, y0, y1) for u in us]
    for rep in range(2):
        got = tv._views(scene, us, w, h, y0, y1)
        for v in range(n):
            d = (got[v] != want[v]).any(axis=-1)
            if d.any():
                ys, xs = np.nonzero(d)
                print(f"{w}x{h} n={n} rep={rep} view={v}: {int(d.sum())} pixels differ, x {xs.min()}..{xs.max()} y {ys.min()}..{ys.max()}; tiles", sorted(set(zip((ys // 32).tolist(), (xs // 32).tolist())))[:12],
                      "got", got[v][ys[0], xs[0]], "want", want[v][ys[0], xs[0]], "untouched" if (got[v][d] == 0xa5).all() else "")
            else:
                print(f"{w}x{h} n={n} rep={rep} view={v}: equal")
r.close()

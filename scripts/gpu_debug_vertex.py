import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_utils as pu, oracle_binding as ob
s = pu.SmallScene.build(width=480, height=270, sky_size=16, pre_size=16, pre_mips=3, lut_size=16, irr_size=4)
r, o = s.renderer, s.oracle
u = s.uniforms()
n = o.verts.size // 8
clip, world, nrm = np.zeros((n, 4), np.float32), np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32)
s.pkg._lib.check(r._ctx, r._lib.pbr_debug_vertex_stage(r._ctx, C.byref(u), ob.ptr(clip), ob.ptr(world), ob.ptr(nrm)))
oclip, oworld, onrm = np.zeros_like(clip), np.zeros_like(world), np.zeros_like(nrm)
sc = o.c_scene(); ou = pu.copy_uniforms(u, ob.FrameUniforms)
o.lib.orc_vertex_transform(C.byref(sc), C.byref(ou), ob.ptr(oclip), ob.ptr(oworld), ob.ptr(onrm))
for name, a, b in (("clip", clip, oclip), ("world", world, oworld), ("nrm", nrm, onrm)):
    ne = a.view(np.uint32) != b.view(np.uint32)
    print(name, "mismatch", ne.sum(), "of", ne.size, "per column", ne.sum(0), "max abs", np.abs(a - b).max())
    idx = np.argwhere(ne)[:5]
    for i, j in idx:
        print("  ", i, j, a[i, j], b[i, j], hex(a.view(np.uint32)[i, j]), hex(b.view(np.uint32)[i, j]))
print("model", list(u.model)); print("proj", list(u.projection)); print("view", list(u.view))

"""Per-workgroup timeline of the set-up launch of one 4K frame (diagnostic build, see gpu_surface_trace.py)."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry

pkg = entry.load_package()
r = pkg.Renderer(width=3840, height=2160, skybox_size=1024)
r.init()
for _ in range(5):
    _, st = r.draw_frame(want_stats=True)
lib = pkg._lib.load()
raw = np.zeros((16384, 8), dtype=np.uint64)
fn = lib.pbr_debug_setup_trace
fn.argtypes = [C.c_void_p, C.c_size_t]
assert fn(raw.ctypes.data, raw.nbytes) == 0
t = raw.astype(np.int64)
used = t[:, 0] > 0
t = t[used]
t0 = t[:, 0].min()
geo = t[:, 3] > 0
us = lambda a: (a - t0) * 0.01
g, s = t[geo], t[~geo]
q = lambda a: {"min": float(a.min()), "median": float(np.median(a)), "p90": float(np.percentile(a, 90)), "max": float(a.max())}
out = {"ms_setup_event": float(st.ms_setup), "workgroups": int(used.sum()), "geometry_workgroups": int(geo.sum()),
       "geometry": {"start": q(us(g[:, 0])), "vertices_done": q(us(g[:, 1])[g[:, 1] > 0]), "records_issued": q(us(g[:, 2])[g[:, 2] > 0]), "binned": q(us(g[:, 3])), "end": q(us(g[:, 4])),
                    "positions_known": q(us(g[:, 5])[g[:, 5] > 0]), "references_stored": q(us(g[:, 6])[g[:, 6] > 0])},
       "sky": {"start": q(us(s[:, 0])), "end": q(us(s[:, 4])), "duration": q((s[:, 4] - s[:, 0]) * 0.01)}}
end_all = us(t[:, 4])
out["running_by_us"] = [(e, int(((us(t[:, 0]) <= e) & (end_all > e)).sum())) for e in range(0, int(end_all.max()) + 2)]
print(json.dumps(out))

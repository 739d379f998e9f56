"""Per-wavefront timeline of the surface kernel of one 4K frame (needs the diagnostic build:
scripts/build_variant.sh trace -DPBR_SURFACE_TRACE; PBR355_LIBRARY=build/trace/libpbr355.so python scripts/gpu_surface_trace.py)."""
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
raw = np.zeros((1024 * 16, 16), dtype=np.uint64)
fn = lib.pbr_debug_surface_trace
fn.argtypes = [C.c_void_p, C.c_size_t]
assert fn(raw.ctypes.data, raw.nbytes) == 0
np.save(os.path.join(ROOT, "gpurun_out", "surface_trace.npy"), raw)
used = raw[raw[:, 0] != 0]
start = used[:, 0].astype(np.int64)
info = used[:, 1]
block = (info >> np.uint64(32)).astype(np.int64)
own_mixed = ((info >> np.uint64(16)) & np.uint64(0xffff)).astype(np.int64)
own = (info & np.uint64(0xffff)).astype(np.int64)
ends = (used[:, 2:] >> np.uint64(1)).astype(np.int64)
kind = (used[:, 2:] & np.uint64(1)).astype(np.int64)
valid = used[:, 2:] != 0
t0 = start.min()
tick_us = 0.01   # 100 MHz
n_items = valid.sum(1)
last = np.where(valid, ends, 0).max(1)
last = np.where(n_items > 0, last, start)
dur = []
for w in range(len(used)):
    prev = start[w]
    for i in range(int(n_items[w])):
        dur.append((int(kind[w, i]), (ends[w, i] - prev) * tick_us))
        prev = ends[w, i]
dur = np.array(dur)
out = {
    "waves": int(len(used)), "ms_resolve_event": float(st.ms_resolve),
    "start_us": {"min": 0.0, "median": float(np.median(start - t0) * tick_us), "max": float((start - t0).max() * tick_us)},
    "finish_us": {"min": float((last - t0).min() * tick_us), "median": float(np.median(last - t0) * tick_us), "p90": float(np.percentile(last - t0, 90) * tick_us),
                  "max": float((last - t0).max() * tick_us)},
    "items_per_wave": {"min": int(n_items.min()), "mean": float(n_items.mean()), "max": int(n_items.max())},
    "items": int(n_items.sum()), "mixed_items": int((dur[:, 0] == 0).sum()), "uniform_items": int((dur[:, 0] == 1).sum()),
    "item_us_mixed": {"median": float(np.median(dur[dur[:, 0] == 0, 1])), "p90": float(np.percentile(dur[dur[:, 0] == 0, 1], 90))},
    "item_us_uniform": {"median": float(np.median(dur[dur[:, 0] == 1, 1])), "p90": float(np.percentile(dur[dur[:, 0] == 1, 1], 90))},
    "own_items_per_workgroup": {"min": int(own.min()), "max": int(own.max())}, "own_mixed_per_workgroup": {"min": int(own_mixed.min()), "max": int(own_mixed.max())},
}
# busy waves over time
edges = np.arange(0, (last - t0).max() + 100, 100)   # 1 us bins
busy = [(int(e * tick_us), int(((start - t0 <= e) & (last - t0 > e)).sum())) for e in edges]
out["busy_waves_by_us"] = busy
# per-workgroup finish spread
wg_finish = {}
for b, l in zip(block, last - t0):
    wg_finish[b] = max(wg_finish.get(b, 0), l)
f = np.array(sorted(wg_finish.values())) * tick_us
out["workgroup_finish_us"] = {"min": float(f.min()), "median": float(np.median(f)), "p90": float(np.percentile(f, 90)), "max": float(f.max())}
print(json.dumps(out))

"""One-GPU rehearsal of the sharded IBL prefilter (2048^2, 10 mips, 1024 samples): every shard of a 2-, 4- and
8-way partition is filtered alone and timed with HIP events on the stream it runs on; reports max / mean per
partition (the speed-up N GPUs can reach is N / that ratio) and writes a JSON summary.

usage: python scripts/gpu_prefilter_shards.py [out.json]"""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry   # noqa: E402

pkg = entry.load_package()
lib = pkg._lib.load()
sharding = __import__("wgpu_renderer_amd.sharding", fromlist=["x"])
torch.cuda.init()
stream = torch.cuda.Stream()
ctx = C.c_void_p()
pkg._lib.check(None, lib.pbr_create(0, C.byref(ctx)))
pkg._lib.check(ctx, lib.pbr_set_stream(ctx, C.c_void_p(stream.cuda_stream)))
sky = pkg.hdr.synthetic_sky()
data = np.ascontiguousarray(sky.data, dtype=np.float32)
pkg._lib.check(ctx, lib.pbr_upload_hdr(ctx, data.ctypes.data_as(C.c_void_p), sky.width, sky.height))
size, mips = 2048, 10


def timed(shard, count, repeats=3):
    best = None
    for _ in range(repeats):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        pkg._lib.check(ctx, lib.pbr_ibl_prefilter(ctx, size, mips, 1024, shard, count))
        b.record(stream)
        b.synchronize()
        ms = a.elapsed_time(b)
        best = ms if best is None else min(best, ms)
    return best


def timed_range(first, count, repeats=3):
    best = None
    for _ in range(repeats):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        pkg._lib.check(ctx, lib.pbr_ibl_prefilter_texels(ctx, size, mips, 1024, first, count))
        b.record(stream)
        b.synchronize()
        ms = a.elapsed_time(b)
        best = ms if best is None else min(best, ms)
    return best


timed(0, 1, 1)   # allocate + warm
report = {"config": f"{size}^2 cube, {mips} mips, 1024 samples; each shard alone on one MI355X, best of 3, HIP events",
          "kept_samples_per_level": sharding.prefilter_kept_samples(mips), "whole_ms": timed(0, 1), "partitions": {}}
print("whole map: %.3f ms" % report["whole_ms"], flush=True)
# cost of every face of every level on its own (the partition's cost model is checked against these)
starts = [0]
for m in range(mips):
    starts.append(starts[-1] + 6 * (size >> m) ** 2)
report["face_ms"] = []
for m in range(mips):
    n = (size >> m) ** 2
    row = [timed_range(starts[m] + f * n, n) for f in range(6)]
    report["face_ms"].append(row)
    print("level %d (%d kept samples): faces ms %s  us per kilotexel %s" % (m, report["kept_samples_per_level"][m], " ".join("%.3f" % t for t in row),
                                                                           " ".join("%.2f" % (t * 1e3 / (n / 1e3)) for t in row)), flush=True)
for n in (2, 4, 8):
    ms = [timed(k, n) for k in range(n)]
    cuts = sharding.prefilter_cuts(size, mips, n)
    mean = sum(ms) / n
    report["partitions"][str(n)] = {"shard_ms": ms, "cuts_texels": cuts, "max_over_mean": max(ms) / mean,
                                    "speedup_bound": report["whole_ms"] / max(ms)}
    print("N=%d shards ms %s max/mean %.3f speed-up bound %.2fx" % (n, " ".join("%.3f" % t for t in ms), max(ms) / mean, report["whole_ms"] / max(ms)), flush=True)
out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "prefilter_shards.json")
os.makedirs(os.path.dirname(out), exist_ok=True)
with open(out, "w") as f:
    json.dump(report, f, indent=1)
lib.pbr_destroy(ctx)

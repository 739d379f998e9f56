"""Times row bands of single faces of the large prefilter's levels 1 and 2 alone (HIP events, best of 3): how the cost of
a texel varies inside a face.  python scripts/gpu_prefilter_bands.py [bands]"""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
pkg = entry.load_package()
lib = pkg._lib.load()
torch.cuda.init()
stream = torch.cuda.Stream()
ctx = C.c_void_p()
pkg._lib.check(None, lib.pbr_create(0, C.byref(ctx)))
pkg._lib.check(ctx, lib.pbr_set_stream(ctx, C.c_void_p(stream.cuda_stream)))
sky = pkg.hdr.synthetic_sky()
data = np.ascontiguousarray(sky.data, dtype=np.float32)
pkg._lib.check(ctx, lib.pbr_upload_hdr(ctx, data.ctypes.data_as(C.c_void_p), sky.width, sky.height))
size, mips = 2048, 10
bands = int(sys.argv[1]) if len(sys.argv) > 1 else 8
pkg._lib.check(ctx, lib.pbr_ibl_prefilter(ctx, size, mips, 1024, 0, 1))
def timed(first, count):
    best = 1e9
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        pkg._lib.check(ctx, lib.pbr_ibl_prefilter_texels(ctx, size, mips, 1024, first, count))
        b.record(stream); b.synchronize()
        best = min(best, a.elapsed_time(b))
    return best
start = 0
for m in range(3):
    s = size >> m
    if m:
        for face in range(6):
            n = s * s // bands
            row = [timed(start + face * s * s + k * n, n) for k in range(bands)]
            print("level %d face %d: ns per texel by band of rows %s" % (m, face, " ".join("%.2f" % (t * 1e6 / n) for t in row)), flush=True)
    start += 6 * s * s

"""Runs the IBL prefilter at the large benchmark size (2048^2, 10 mips, 1024 samples) a few times, for profilers."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
pkg = entry.load_package()
lib = pkg._lib.load()
ctx = C.c_void_p()
pkg._lib.check(None, lib.pbr_create(0, C.byref(ctx)))
sky = pkg.hdr.synthetic_sky()
import numpy as np
data = np.ascontiguousarray(sky.data, dtype=np.float32)
pkg._lib.check(ctx, lib.pbr_upload_hdr(ctx, data.ctypes.data_as(C.c_void_p), sky.width, sky.height))
size, mips = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2048, 10)
for i in range(3):
    t0 = time.perf_counter()
    pkg._lib.check(ctx, lib.pbr_ibl_prefilter(ctx, size, mips, 1024, 0, 1))
    pkg._lib.check(ctx, lib.pbr_synchronize(ctx))
    print("prefilter %d^2 x%d: %.2f ms" % (size, mips, (time.perf_counter() - t0) * 1e3), flush=True)
lib.pbr_destroy(ctx)

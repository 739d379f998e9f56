"""SHA-256 of every level of the GGX prefilter map for the two benchmark configurations (128^2 x 5 and 2048^2 x 10
levels, synthetic 4096x2048 sky, seed 355), written as JSON: the record that later kernel work must reproduce bit
for bit.  PBR355_LIBRARY selects the build of the library that is hashed.

usage: python scripts/gpu_prefilter_hashes.py out.json"""
import ctypes as C
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry   # noqa: E402


def level_hashes(pkg, renderer, size, mips):
    lib = pkg._lib
    lib.check(renderer._ctx, renderer._lib.pbr_ibl_prefilter(renderer._ctx, size, mips, 1024, 0, 1))
    return [hashlib.sha256(np.ascontiguousarray(renderer.download_map(lib.MAP_PREFILTER, m)).tobytes()).hexdigest() for m in range(mips)]


def main(out):
    pkg = entry.load_package()
    if os.environ.get("PBR355_LIBRARY"):   # an older build lacks the newer exports: bind only what it has
        pkg._lib._bind_to_torch_runtime()
        probe = C.CDLL(pkg._lib.LIB_PATH)
        for name in [n for n in pkg._lib.EXPORTS if not hasattr(probe, n)]:
            del pkg._lib.EXPORTS[name]
    hdr = pkg.synthetic_sky(4096, 2048)
    data = np.ascontiguousarray(hdr.data, dtype=np.float32)
    r = pkg.Renderer(width=64, height=64, hdr=hdr)
    r._open()
    pkg._lib.check(r._ctx, r._lib.pbr_upload_hdr(r._ctx, data.ctypes.data_as(C.c_void_p), hdr.width, hdr.height))
    report = {"library": os.environ.get("PBR355_LIBRARY", "in-tree"), "hdr_sha256": hashlib.sha256(data.tobytes()).hexdigest(),
              "128x5": level_hashes(pkg, r, 128, 5), "2048x10": level_hashes(pkg, r, 2048, 10)}
    r.close()
    with open(out, "w") as f:
        json.dump(report, f, indent=1)
    print(json.dumps(report)[:200], flush=True)


if __name__ == "__main__":
    main(sys.argv[1])

"""Which HIP / HSA runtime libraries end up mapped when one process uses both torch and libpbr355.so, in either
import order, and whether both sides can use the GPU afterwards.  Each order runs in a fresh child process.

usage: python scripts/gpu_hip_runtime_probe.py            (parent: runs both orders)
       python scripts/gpu_hip_runtime_probe.py torch-first|lib-first   (child)"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def mapped():
    seen = []
    with open("/proc/self/maps") as f:
        for line in f:
            path = line.rsplit(" ", 1)[-1].strip()
            base = os.path.basename(path)
            if any(k in base for k in ("amdhip64", "hsa-runtime", "libhiprtc", "librccl", "libpbr355", "libamd_comgr", "libtorch_hip", "libc10_hip")):
                if path not in seen:
                    seen.append(path)
    return seen


def child(order):
    import __graft_entry__ as entry
    pkg = entry.load_package()

    def use_lib():
        lib = pkg._lib.load()
        ctx = C.c_void_p()
        rc = lib.pbr_create(0, C.byref(ctx))
        print(f"  pbr_create rc={rc} {lib.pbr_last_error(None) if rc else b''}", flush=True)
        return lib, ctx

    def use_torch():
        import torch
        ok = torch.cuda.is_available()
        print(f"  torch.cuda.is_available()={ok}", flush=True)
        if ok:
            x = torch.arange(8, device="cuda:0").sum().item()
            print(f"  torch kernel ok ({x})", flush=True)
        return torch

    if order == "torch-first":
        torch = use_torch(); lib, ctx = use_lib()
    else:
        lib, ctx = use_lib(); torch = use_torch()
    for p in mapped():
        print("  mapped:", p, flush=True)
    # a torch stream handed to the library: only meaningful when both sides talk to one runtime instance
    if ctx and torch.cuda.is_available():
        s = torch.cuda.Stream()
        rc = lib.pbr_set_stream(ctx, C.c_void_p(s.cuda_stream))
        rc2 = lib.pbr_ibl_brdf_lut(ctx, 64)
        rc3 = lib.pbr_synchronize(ctx)
        print(f"  torch stream in library: set_stream rc={rc} lut rc={rc2} sync rc={rc3}", flush=True)
    if ctx:
        lib.pbr_destroy(ctx)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1])
    else:
        for order in ("torch-first", "lib-first"):
            print(order + ":", flush=True)
            proc = subprocess.run([sys.executable, os.path.abspath(__file__), order], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
            print(proc.stdout + f"  exit code {proc.returncode}", flush=True)

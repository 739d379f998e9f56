"""ctypes binding of oracle/liboracle.so — the CPU restatement used as the parity checker.
Only tests, the smoke check and bench.py's cpu_baseline leg load this."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "liboracle.so")


class FrameUniforms(C.Structure):
    _fields_ = [
        ("model", C.c_float * 16), ("normal", C.c_float * 16), ("view", C.c_float * 16), ("projection", C.c_float * 16),
        ("inv_view_proj", C.c_float * 16), ("view_pos", C.c_float * 4), ("light_pos", (C.c_float * 4) * 4),
        ("light_color", (C.c_float * 4) * 4), ("emissive_factor", C.c_float * 4), ("sampler_index", C.c_uint32),
        ("use_mipmaps", C.c_uint32), ("reserved", C.c_uint32 * 2),
    ]


class Scene(C.Structure):
    _fields_ = [
        ("verts", C.c_void_p), ("n_verts", C.c_int32), ("indices", C.c_void_p), ("n_indices", C.c_int32), ("n_instances", C.c_int32),
        ("inst_model", C.c_void_p), ("inst_normal", C.c_void_p), ("tex", C.c_void_p * 5), ("tex_w", C.c_int32 * 5), ("tex_h", C.c_int32 * 5),
        ("tex_levels", C.c_int32 * 5), ("irradiance", C.c_void_p), ("irr_size", C.c_int32), ("prefilter", C.c_void_p), ("pre_size", C.c_int32),
        ("pre_levels", C.c_int32), ("lut_f16", C.c_void_p), ("lut_size", C.c_int32), ("skybox", C.c_void_p), ("sky_size", C.c_int32),
        ("indices32", C.c_void_p),
    ]


class Camera(C.Structure):
    _fields_ = [("eye", C.c_float * 3), ("target", C.c_float * 3), ("up", C.c_float * 3), ("right", C.c_float * 3), ("distance", C.c_float)]


_lib = None
P = C.c_void_p
I64 = C.c_int64


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        subprocess.run(["make"], cwd=ORACLE_DIR, check=True)
    lib = C.CDLL(LIB_PATH)
    sig = {
        "orc_mat4_perspective": (None, [C.c_float, C.c_float, C.c_float, C.c_float, P]),
        "orc_mat4_look_at": (None, [P, P, P, P]),
        "orc_mat4_mul": (None, [P, P, P]),
        "orc_mat4_inverse": (C.c_int, [P, P]),
        "orc_mat4_transpose": (None, [P, P]),
        "orc_mat4_from_quat": (None, [P, P]),
        "orc_mat4_scale": (None, [P, P, P]),
        "orc_mat4_translate": (None, [P, P, P]),
        "orc_node_matrix": (None, [P, P, P, P, P]),
        "orc_camera_init": (None, [C.POINTER(Camera)]),
        "orc_camera_rotate": (None, [C.POINTER(Camera), C.c_float, C.c_float]),
        "orc_camera_dolly": (None, [C.POINTER(Camera), C.c_float, C.c_float]),
        "orc_camera_track": (None, [C.POINTER(Camera), C.c_float, C.c_float]),
        "orc_camera_view": (None, [C.POINTER(Camera), P]),
        "orc_hdr_dims": (C.c_int, [P, I64, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
        "orc_hdr_decode": (C.c_int, [P, I64, P]),
        "orc_mip_level_count": (C.c_int, [C.c_int, C.c_int]),
        "orc_chain_texels": (I64, [C.c_int, C.c_int, C.c_int]),
        "orc_chain_offset": (I64, [C.c_int, C.c_int, C.c_int]),
        "orc_build_mips_rgba32f": (None, [P, C.c_int, C.c_int, C.c_int]),
        "orc_build_mips_rgba8": (None, [P, C.c_int, C.c_int, C.c_int, C.c_int]),
        "orc_equirect_to_cube": (None, [P, C.c_int, C.c_int, C.c_int, P, I64, I64]),
        "orc_irradiance": (None, [P, C.c_int, C.c_int, C.c_int, P, I64, I64]),
        "orc_prefilter": (None, [P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, P, I64, I64]),
        "orc_equirect_to_cube_bounds": (None, [P, C.c_int, C.c_int, C.c_int, C.c_float, P, P, I64, I64]),
        "orc_irradiance_bounds": (None, [P, C.c_int, C.c_int, C.c_int, C.c_float, P, P, I64, I64]),
        "orc_prefilter_bounds": (None, [P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, P, P, I64, I64]),
        "orc_draw_frame_margins": (None, [C.POINTER(Scene), C.POINTER(FrameUniforms), C.c_int, C.c_int, C.c_int, C.c_int, P, P, P]),
        "orc_draw_frame_bounds": (None, [C.POINTER(Scene), C.POINTER(FrameUniforms), C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, P, P, P, P, P]),
        "orc_brdf_lut": (None, [C.c_int, P, P, C.c_int, C.c_int]),
        "orc_radical_inverse": (C.c_float, [C.c_uint32]),
        "orc_irradiance_trip_counts": (None, [C.POINTER(C.c_int), C.POINTER(C.c_int)]),
        "orc_f32_to_f16": (C.c_uint16, [C.c_float]),
        "orc_f16_to_f32": (C.c_float, [C.c_uint16]),
        "orc_vertex_transform": (None, [C.POINTER(Scene), C.POINTER(FrameUniforms), P, P, P]),
        "orc_raster": (None, [C.POINTER(Scene), C.POINTER(FrameUniforms), C.c_int, C.c_int, C.c_int, C.c_int, P, P]),
        "orc_shade": (None, [C.POINTER(Scene), C.POINTER(FrameUniforms), C.c_int, C.c_int, C.c_int, C.c_int, P, P]),
        "orc_draw_frame": (None, [C.POINTER(Scene), C.POINTER(FrameUniforms), C.c_int, C.c_int, C.c_int, C.c_int, P, P]),
        "orc_sample_cube": (None, [P, C.c_int, C.c_int, P, C.c_float, P]),
        "orc_sample_2d_rgba8": (None, [P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_float, C.c_float, P, P, P]),
        "orc_num_threads": (C.c_int, []),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


# ---- numpy-level helpers over the raw calls ----
def mip_chain_rgba8(img, srgb, levels=None):
    """img [h, w, 4] uint8 -> flat chain (uint8) with all levels; returns (chain, levels)."""
    lib = load()
    h, w = img.shape[:2]
    levels = levels or lib.orc_mip_level_count(w, h)
    total = lib.orc_chain_texels(w, h, levels)
    chain = np.zeros(total * 4, dtype=np.uint8)
    chain[:w * h * 4] = img.reshape(-1)
    lib.orc_build_mips_rgba8(ptr(chain), w, h, levels, 1 if srgb else 0)
    return chain, levels


def mip_chain_rgba32f(data, w, h, levels=None):
    lib = load()
    levels = levels or lib.orc_mip_level_count(w, h)
    total = lib.orc_chain_texels(w, h, levels)
    chain = np.zeros(total * 4, dtype=np.float32)
    chain[:w * h * 4] = np.asarray(data, dtype=np.float32).reshape(-1)
    lib.orc_build_mips_rgba32f(ptr(chain), w, h, levels)
    return chain, levels


def chain_level(chain, w, h, level, channels=4):
    lib = load()
    off = lib.orc_chain_offset(w, h, level)
    lw, lh = max(1, w >> level), max(1, h >> level)
    return chain[off * channels:(off + lw * lh) * channels].reshape(lh, lw, channels)


def equirect_to_cube(hdr, w, h, size, first=0, count=-1):
    lib = load()
    out = np.zeros((6, size, size, 4), dtype=np.float32)
    lib.orc_equirect_to_cube(ptr(hdr), w, h, size, ptr(out), first, count)
    return out


def irradiance(hdr, w, h, size, first=0, count=-1):
    lib = load()
    out = np.zeros((6, size, size, 4), dtype=np.float32)
    lib.orc_irradiance(ptr(hdr), w, h, size, ptr(out), first, count)
    return out


def prefilter_level(chain, w, h, levels, size, roughness, first=0, count=-1):
    lib = load()
    out = np.zeros((6, size, size, 4), dtype=np.float32)
    lib.orc_prefilter(ptr(chain), w, h, levels, size, float(roughness), ptr(out), first, count)
    return out


BOUNDARY_SLACK = 5e-7   # radians: how far a lookup direction / the inverse trigonometric functions may differ between implementations


def equirect_to_cube_bounds(hdr, w, h, size, ang=BOUNDARY_SLACK, first=0, count=-1):
    lib = load()
    lo, hi = np.zeros((6, size, size, 4), dtype=np.float32), np.zeros((6, size, size, 4), dtype=np.float32)
    lib.orc_equirect_to_cube_bounds(ptr(hdr), w, h, size, ang, ptr(lo), ptr(hi), first, count)
    return lo, hi


def irradiance_bounds(hdr, w, h, size, ang=BOUNDARY_SLACK, first=0, count=-1):
    lib = load()
    lo, hi = np.zeros((6, size, size, 4), dtype=np.float32), np.zeros((6, size, size, 4), dtype=np.float32)
    lib.orc_irradiance_bounds(ptr(hdr), w, h, size, ang, ptr(lo), ptr(hi), first, count)
    return lo, hi


def prefilter_level_bounds(chain, w, h, levels, size, roughness, ang=BOUNDARY_SLACK, first=0, count=-1):
    lib = load()
    lo, hi = np.zeros((6, size, size, 4), dtype=np.float32), np.zeros((6, size, size, 4), dtype=np.float32)
    lib.orc_prefilter_bounds(ptr(chain), w, h, levels, size, float(roughness), ang, ptr(lo), ptr(hi), first, count)
    return lo, hi


def brdf_lut(size, row0=0, rows=-1):
    lib = load()
    n = size - row0 if rows < 0 else rows
    f32 = np.zeros((n, size, 4), dtype=np.float32)
    f16 = np.zeros((n, size, 4), dtype=np.uint16)
    lib.orc_brdf_lut(size, ptr(f32), ptr(f16), row0, rows)
    return f32, f16

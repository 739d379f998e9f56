"""ctypes binding of libpbr355.so (include/pbr355.h).  There is no fallback: if the shared library
is missing the import fails, and if no gfx950 device is present pbr_create reports it."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PBR355_LIBRARY") or os.path.join(_HERE, "libpbr355.so")   # the variable selects another build of the same library

PBR_OK, PBR_E_ARG, PBR_E_STATE, PBR_E_DEVICE, PBR_E_UNSUPPORTED = 0, -1, -2, -3, -4
TEX_ALBEDO, TEX_METALLIC_ROUGHNESS, TEX_NORMAL, TEX_EMISSIVE, TEX_OCCLUSION = range(5)
MAP_SKYBOX, MAP_IRRADIANCE, MAP_PREFILTER, MAP_BRDF_LUT, MAP_HDR = range(5)


class FrameUniforms(C.Structure):
    _fields_ = [
        ("model", C.c_float * 16), ("normal", C.c_float * 16), ("view", C.c_float * 16), ("projection", C.c_float * 16),
        ("inv_view_proj", C.c_float * 16), ("view_pos", C.c_float * 4), ("light_pos", (C.c_float * 4) * 4),
        ("light_color", (C.c_float * 4) * 4), ("emissive_factor", C.c_float * 4), ("sampler_index", C.c_uint32),
        ("use_mipmaps", C.c_uint32), ("reserved", C.c_uint32 * 2),
    ]


class FrameStats(C.Structure):
    _fields_ = [
        ("ms_vertex", C.c_float), ("ms_setup", C.c_float), ("ms_raster", C.c_float), ("ms_resolve", C.c_float), ("ms_total", C.c_float),
        ("covered_pixels", C.c_uint32), ("binned_refs", C.c_uint32), ("setup_triangles", C.c_uint32), ("uniform_quads", C.c_uint32), ("mixed_quads", C.c_uint32),
    ]


class PbrError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(message)
        self.code = code


EXPORTS = {
    "pbr_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "pbr_destroy": (C.c_int, [C.c_void_p]),
    "pbr_fork": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "pbr_last_error": (C.c_char_p, [C.c_void_p]),
    "pbr_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pbr_synchronize": (C.c_int, [C.c_void_p]),
    "pbr_upload_mesh": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "pbr_upload_mesh_u32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "pbr_upload_texture": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "pbr_upload_hdr": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]),
    "pbr_ibl_equirect_to_cube": (C.c_int, [C.c_void_p, C.c_int32]),
    "pbr_ibl_irradiance": (C.c_int, [C.c_void_p, C.c_int32]),
    "pbr_ibl_prefilter": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "pbr_ibl_prefilter_texels": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_uint64, C.c_uint64]),
    "pbr_ibl_brdf_lut": (C.c_int, [C.c_void_p, C.c_int32]),
    "pbr_ibl_level_info": (C.c_int, [C.c_void_p, C.c_int, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_size_t)]),
    "pbr_ibl_download": (C.c_int, [C.c_void_p, C.c_int, C.c_int32, C.c_void_p, C.c_size_t]),
    "pbr_ibl_device_ptr": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "pbr_ibl_prefilter_shard_range": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "pbr_ibl_prefilter_partition": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_uint64)]),
    "pbr_ibl_prefilter_kept_samples": (C.c_int, [C.c_int32, C.c_int32, C.POINTER(C.c_int32)]),
    "pbr_draw_frame": (C.c_int, [C.c_void_p, C.POINTER(FrameUniforms), C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(FrameStats)]),
    "pbr_draw_frame_device": (C.c_int, [C.c_void_p, C.POINTER(FrameUniforms), C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(FrameStats)]),
    "pbr_draw_views_device": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(FrameUniforms), C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    "pbr_release_view_buffers": (C.c_int, [C.c_void_p]),
    "pbr_draw_frame_static": (C.c_int, [C.c_void_p, C.POINTER(FrameUniforms), C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "pbr_debug_vertex_stage": (C.c_int, [C.c_void_p, C.POINTER(FrameUniforms), C.c_void_p, C.c_void_p, C.c_void_p]),
    "pbr_debug_texture_level": (C.c_int, [C.c_void_p, C.c_int, C.c_int32, C.c_void_p, C.c_size_t, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "pbr_debug_set_capacity": (C.c_int, [C.c_void_p, C.c_size_t, C.c_size_t]),
    "pbr_debug_capture_depth": (C.c_int, [C.c_void_p, C.c_int32]),
    "pbr_debug_depth": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "pbr_profile_begin": (C.c_int, [C.c_void_p, C.c_int32]),
    "pbr_profile_end": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "pbr_profile_begin_kernels": (C.c_int, [C.c_void_p, C.c_int32]),
    "pbr_profile_end_kernels": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_float), C.POINTER(C.c_float)]),
}

_lib = None


def _hip_runtimes_mapped():
    """Paths of the HIP runtime libraries mapped into this process."""
    seen = []
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                path = line.rsplit(" ", 1)[-1].strip()
                if os.path.basename(path).startswith("libamdhip64.so") and path not in seen:
                    seen.append(path)
    except OSError:
        pass
    return seen


def _bind_to_torch_runtime():
    """One process must talk to the GPU through ONE HIP runtime.  libpbr355.so needs `libamdhip64.so.7`; a PyTorch
    ROCm wheel ships its own runtime as `torch/lib/libamdhip64.so`, whose SONAME is also `libamdhip64.so.7`.  The
    dynamic loader matches a needed library against the loaded ones by name: with torch loaded first, libpbr355.so is
    bound to torch's copy (one runtime); with libpbr355.so loaded first, /opt/rocm's copy is mapped, torch's request
    for `libamdhip64.so` matches neither its name nor its SONAME, and a second runtime (with a second HSA runtime
    under it) is mapped - the one that initialises second then finds no usable GPU ("No HIP GPUs are available"),
    and a hipStream_t of one is meaningless to the other.  So when a torch with its own runtime is installed, that
    runtime is mapped first, globally, whether or not torch has been imported yet; both orders then end with it."""
    import importlib.util
    if any("torch" in p for p in _hip_runtimes_mapped()):
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    bundled = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.isfile(bundled):
        C.CDLL(bundled, mode=C.RTLD_GLOBAL)


def load():
    """Load libpbr355.so and declare every export of include/pbr355.h."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        _bind_to_torch_runtime()
        lib = C.CDLL(LIB_PATH)
        runtimes = _hip_runtimes_mapped()
        if len(runtimes) > 1:
            raise ImportError("two HIP runtimes are mapped into this process (" + ", ".join(runtimes) + "): libpbr355.so and torch would each "
                              "open the GPU through their own; load libpbr355 through this module before anything maps /opt/rocm's runtime")
        for name, (restype, argtypes) in EXPORTS.items():
            fn = getattr(lib, name)
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = lib
    return _lib


def check(ctx, rc):
    if rc != PBR_OK:
        msg = load().pbr_last_error(ctx)
        raise PbrError(rc, (msg or b"").decode("utf-8", "replace") or f"pbr355 error {rc}")

"""Headless `Renderer` with the surface of the reference's class (src/renderer.ts:29-2052):
`Renderer().render()` initialises everything the reference's `init()` does and draws a frame.
WebGPU objects become calls into libpbr355 (include/pbr355.h); the browser plumbing (canvas,
adapter, dat.gui) is gone and its default option values are plain attributes."""
import ctypes as C
import os

import numpy as np

from . import _lib, mathx
from .camera import Camera
from .gltf import DEFAULT_TEXEL, glTFLoader
from .hdr import HDR, HDRLoader, synthetic_sky
from .images import load_image_bitmap

_REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_MODEL = os.path.join(_REPO, "assets", "DamagedHelmet", "DamagedHelmet.gltf")   # renderer.ts:283


def _fptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Renderer:
    def __init__(self, width=1280, height=720, device=0, model_path=DEFAULT_MODEL, hdr=None,
                 skybox_size=1440, irradiance_size=32, prefilter_size=128, prefilter_mips=5, lut_size=512,
                 instance_models=None, reference_limits=True):
        self.camera = Camera()                                    # renderer.ts:77
        self.width, self.height = int(width), int(height)         # canvas size
        self.device_ordinal = device
        self.model_path = model_path
        self.hdr_source = hdr                                      # HDR, path to a .hdr, or None for the synthetic sky
        self.skybox_size, self.irradiance_size = skybox_size, irradiance_size      # renderer.ts:983, :764
        self.prefilter_size, self.prefilter_mips = prefilter_size, prefilter_mips  # renderer.ts:773, :1545
        self.lut_size = lut_size                                                   # renderer.ts:779
        self.instance_models = instance_models
        self.reference_limits = reference_limits                  # False: glTF read per the specification (see gltf.py)
        # dat.gui defaults, renderer.ts:1762-1768, :1793-1795
        self.texture_options = {"mipmaps": True, "addressModeU": "repeat", "addressModeV": "repeat",
                                "magFilter": "linear", "minFilter": "linear"}
        self.msaa_options = {"enable": False}
        self.geometries = None
        self.hdrs = None
        self._lib = None
        self._ctx = C.c_void_p()
        self._ready = False
        self._forks = []        # renderers forked from this one (they share its scene on the device)
        self._parent = None

    # ---- lifetime -------------------------------------------------------------------------
    def _open(self):
        if self._lib is None:
            self._lib = _lib.load()
            rc = self._lib.pbr_create(self.device_ordinal, C.byref(self._ctx))
            if rc != _lib.PBR_OK:
                self._ctx = C.c_void_p()
                raise _lib.PbrError(rc, (self._lib.pbr_last_error(None) or b"").decode())

    def fork(self):
        """A renderer that draws this renderer's scene from the same device copy of it (pbr_fork): its own
        camera, stream and working buffers, for keeping several frames in flight.  Close it before this one."""
        if not self._ready:
            self.init()
        other = Renderer.__new__(Renderer)
        other.__dict__.update(self.__dict__)
        other.camera = Camera()
        other.texture_options = dict(self.texture_options)
        other.msaa_options = dict(self.msaa_options)
        other._ctx = C.c_void_p()
        other._forks = []
        other._parent = self
        self._check(self._lib.pbr_fork(self._ctx, C.byref(other._ctx)))
        self._forks.append(other)
        return other

    def close(self):
        for f in list(getattr(self, "_forks", [])):
            f.close()
        if self._lib is not None and self._ctx:
            self._lib.pbr_destroy(self._ctx)
            self._ctx = C.c_void_p()
            parent = getattr(self, "_parent", None)
            if parent is not None and self in parent._forks:
                parent._forks.remove(self)
        self._ready = False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        _lib.check(self._ctx, rc)

    # ---- init(), renderer.ts:107-136 --------------------------------------------------------
    def init(self):
        self._open()
        self.load_model()
        self.load_hdr()
        self.create_vertex_buffer()
        self.create_texture()
        self.create_skybox()
        self._ready = True

    def load_model(self):
        loader = glTFLoader(reference_limits=self.reference_limits)
        loader.load(self.model_path)                               # renderer.ts:281-286
        self.geometries = loader.get_geometries()

    def load_hdr(self):
        src = self.hdr_source
        if src is None:
            hdr = synthetic_sky()
        elif isinstance(src, HDR):
            hdr = src
        else:
            loader = HDRLoader()                                   # renderer.ts:288-294
            loader.load(src)
            hdr = loader.hdr
        self.hdrs = [hdr]

    def create_vertex_buffer(self):
        g = self.geometries[0]                                     # renderer.ts:506, :525: only geometry 0 is drawn
        verts = np.ascontiguousarray(g.vertices, dtype=np.float32)
        # the reference's Uint16Array (glTF.ts:222-226); the general loader hands over uint32 for larger meshes
        wide = np.asarray(g.indices).dtype == np.uint32
        idx = np.ascontiguousarray(g.indices, dtype=np.uint32 if wide else np.uint16)
        upload = self._lib.pbr_upload_mesh_u32 if wide else self._lib.pbr_upload_mesh
        if self.instance_models is None:
            self._check(upload(self._ctx, _fptr(verts), verts.size // 8, _fptr(idx), idx.size, 1, None, None))
        else:
            models = np.ascontiguousarray(self.instance_models, dtype=np.float32).reshape(-1, 16)
            normals = np.stack([mathx.transpose(mathx.inverse(m)) for m in models]).astype(np.float32)
            self._check(upload(self._ctx, _fptr(verts), verts.size // 8, _fptr(idx), idx.size, len(models), _fptr(models), _fptr(normals)))

    def create_texture(self):
        """renderer.ts:604-860: material textures (with their mip chains), irradiance and prefilter
        cubes, BRDF LUT."""
        t = self.geometries[0].textures
        slots = [(_lib.TEX_ALBEDO, t["baseColorURI"], 1), (_lib.TEX_METALLIC_ROUGHNESS, t["metallicRoughnessURI"], 0),
                 (_lib.TEX_NORMAL, t["normalURI"], 0), (_lib.TEX_EMISSIVE, t["emissiveURI"], 1), (_lib.TEX_OCCLUSION, t["occlusionURI"], 0)]
        for slot, uri, srgb in slots:
            if uri is None:      # only a loader without the reference's limits leaves a texture out: neutral 1x1 stand-in
                names = ("baseColorURI", "metallicRoughnessURI", "normalURI", "emissiveURI", "occlusionURI")
                img = np.array([[DEFAULT_TEXEL[names[slot]]]], dtype=np.uint8)
            else:
                img = load_image_bitmap(uri)
            self._check(self._lib.pbr_upload_texture(self._ctx, slot, _fptr(img), img.shape[1], img.shape[0], srgb, 1))
        hdr = self.hdrs[0]
        data = np.ascontiguousarray(hdr.data, dtype=np.float32)
        self._check(self._lib.pbr_upload_hdr(self._ctx, _fptr(data), hdr.width, hdr.height))
        self._check(self._lib.pbr_ibl_irradiance(self._ctx, self.irradiance_size))
        self._check(self._lib.pbr_ibl_prefilter(self._ctx, self.prefilter_size, self.prefilter_mips, 1024, 0, 1))
        self._check(self._lib.pbr_ibl_brdf_lut(self._ctx, self.lut_size))

    def create_skybox(self):
        self._check(self._lib.pbr_ibl_equirect_to_cube(self._ctx, self.skybox_size))   # renderer.ts:978-985
        self._check(self._lib.pbr_synchronize(self._ctx))

    # ---- per-frame uniforms, renderer.ts:1828-1932 ---------------------------------------------
    def frame_uniforms(self):
        u = _lib.FrameUniforms()
        g = self.geometries[0]
        model = np.asarray(g.model, dtype=np.float32)
        normal = mathx.transpose(mathx.inverse(model))                              # renderer.ts:1837
        view = self.camera.view                                                      # renderer.ts:1841
        aspect = self.width / self.height
        projection = mathx.perspective(mathx.deg_to_rad(45.0), aspect, 1.0, 100.0)   # renderer.ts:1846-1851
        inv_vp = mathx.inverse(mathx.mul(projection, view))                          # renderer.ts:1867
        for name, m in (("model", model), ("normal", normal), ("view", view), ("projection", projection), ("inv_view_proj", inv_vp)):
            getattr(u, name)[:] = [float(x) for x in m]
        eye = self.camera.position
        u.view_pos[:] = [float(eye[0]), float(eye[1]), float(eye[2]), 0.0]
        lights = [(10.0, 10.0, 10.0), (-10.0, 10.0, 10.0), (10.0, -10.0, 10.0), (-10.0, -10.0, 10.0)]   # renderer.ts:1891-1894
        for i, lp in enumerate(lights):
            u.light_pos[i][:] = [lp[0], lp[1], lp[2], 0.0]
            u.light_color[i][:] = [300.0, 300.0, 300.0, 0.0]                        # renderer.ts:1909-1912
        e = g.textures["emissive"]
        u.emissive_factor[:] = [float(e[0]), float(e[1]), float(e[2]), 0.0]
        o = self.texture_options
        u.sampler_index = ((1 if o["addressModeU"] == "repeat" else 0) + (2 if o["addressModeV"] == "repeat" else 0) +
                           (4 if o["magFilter"] == "linear" else 0) + (8 if o["minFilter"] == "linear" else 0) +
                           (16 if self.msaa_options["enable"] else 0))              # renderer.ts:1927-1932
        u.use_mipmaps = 1 if o["mipmaps"] else 0                                    # renderer.ts:1977
        return u

    # ---- drawFrame, renderer.ts:1820-2051 ------------------------------------------------------
    def draw_frame(self, y0=0, y1=None, want_vis=False, want_stats=False):
        if not self._ready:
            self.init()
        y1 = self.height if y1 is None else y1
        u = self.frame_uniforms()
        rows = y1 - y0
        rgba = np.empty((rows, self.width, 4), dtype=np.uint8)
        # one triangle id per pixel, or one per sample with multisampling on
        vis_shape = (rows, self.width, 4) if u.sampler_index & 16 else (rows, self.width)
        vis = np.empty(vis_shape, dtype=np.uint32) if want_vis else None
        stats = _lib.FrameStats() if want_stats else None
        self._check(self._lib.pbr_draw_frame(self._ctx, C.byref(u), self.width, self.height, y0, y1, _fptr(rgba),
                                             _fptr(vis) if want_vis else None, C.byref(stats) if want_stats else None))
        out = [rgba]
        if want_vis:
            out.append(vis)
        if want_stats:
            out.append(stats)
        return out[0] if len(out) == 1 else tuple(out)

    def draw_frame_device(self, d_rgba8, d_vis=None, y0=0, y1=None, uniforms=None, stats=None):
        """Same pass with device-resident outputs (integer device addresses, e.g. tensor.data_ptr())."""
        if not self._ready:
            self.init()
        y1 = self.height if y1 is None else y1
        u = uniforms if uniforms is not None else self.frame_uniforms()
        self._check(self._lib.pbr_draw_frame_device(self._ctx, C.byref(u), self.width, self.height, y0, y1, C.c_void_p(d_rgba8),
                                                    C.c_void_p(d_vis) if d_vis else None, C.byref(stats) if stats is not None else None))

    def draw_views_device(self, uniforms, d_rgba8, y0=0, y1=None):
        """Several frames of the scene in one pass: view v is drawn with uniforms[v] into device address d_rgba8[v]
        (one to four views); every target gets the bytes draw_frame_device would have given it."""
        if not self._ready:
            self.init()
        y1 = self.height if y1 is None else y1
        n = len(uniforms)
        if n != len(d_rgba8):
            raise ValueError("one colour target per view")
        us = (_lib.FrameUniforms * n)(*uniforms)
        outs = (C.c_void_p * n)(*[C.c_void_p(int(a)) for a in d_rgba8])
        self._check(self._lib.pbr_draw_views_device(self._ctx, n, us, self.width, self.height, y0, y1, outs))

    def release_view_buffers(self):
        """Frees the working buffers draw_views_device keeps for the second and further views of its passes."""
        self._check(self._lib.pbr_release_view_buffers(self._ctx))

    def render(self):
        """renderer.ts:102-105: init then run; headless `run` is one frame, returned as RGBA8."""
        if not self._ready:
            self.init()
        return self.draw_frame()

    # ---- helpers --------------------------------------------------------------------------
    def synchronize(self):
        self._check(self._lib.pbr_synchronize(self._ctx))

    def set_stream(self, hip_stream):
        self._check(self._lib.pbr_set_stream(self._ctx, C.c_void_p(hip_stream) if hip_stream else None))

    def download_map(self, which, level=0):
        w, h, layers, nbytes = C.c_int32(), C.c_int32(), C.c_int32(), C.c_size_t()
        self._check(self._lib.pbr_ibl_level_info(self._ctx, which, level, C.byref(w), C.byref(h), C.byref(layers), C.byref(nbytes)))
        if which == _lib.MAP_BRDF_LUT:
            out = np.empty((h.value, w.value, 4), dtype=np.uint16)
        elif layers.value == 6:
            out = np.empty((6, h.value, w.value, 4), dtype=np.float32)
        else:
            out = np.empty((h.value, w.value, 4), dtype=np.float32)
        self._check(self._lib.pbr_ibl_download(self._ctx, which, level, _fptr(out), out.nbytes))
        return out

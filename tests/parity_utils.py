"""Shared scaffolding for the parity tests: builds one scene twice — through the product's
C ABI (GPU) and through the CPU oracle — from the same inputs."""
import contextlib
import ctypes as C
import os

import numpy as np

import oracle_binding as ob

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _pkg():
    import __graft_entry__ as entry
    return entry.load_package()


def copy_uniforms(src, dst_type):
    dst = dst_type()
    C.memmove(C.byref(dst), C.byref(src), C.sizeof(dst_type))
    return dst


def vis_shape(uniforms, rows, W):
    """One triangle id per pixel, or four (one per sample) when bit 16 of the sampler index selects
    the multisampled pipeline (renderer.ts:1927-1932)."""
    return (rows, W, 4) if uniforms.sampler_index & 16 else (rows, W)


class OracleScene:
    """Everything the oracle's frame functions need, computed by the oracle itself."""

    def __init__(self, geometry, images, hdr, sky_size, irr_size, pre_size, pre_mips, lut_size, instance_models=None, instance_normals=None):
        lib = ob.load()
        self.lib = lib
        self.verts = np.ascontiguousarray(geometry.vertices, dtype=np.float32)
        wide = np.asarray(geometry.indices).dtype == np.uint32
        self.indices = np.ascontiguousarray(geometry.indices, dtype=np.uint32 if wide else np.uint16)
        srgb = [True, False, False, True, False]
        self.chains = []
        self.levels = []
        for img, s in zip(images, srgb):
            chain, levels = ob.mip_chain_rgba8(img, s)
            self.chains.append(chain)
            self.levels.append(levels)
        self.images = images
        self.hdr = hdr
        self.hdr_data = np.ascontiguousarray(hdr.data, dtype=np.float32)
        self.hdr_chain, self.hdr_levels = ob.mip_chain_rgba32f(self.hdr_data, hdr.width, hdr.height)
        self.sky_size, self.irr_size, self.pre_size, self.pre_mips, self.lut_size = sky_size, irr_size, pre_size, pre_mips, lut_size
        self.inst_model = None if instance_models is None else np.ascontiguousarray(instance_models, dtype=np.float32).reshape(-1, 16)
        self.inst_normal = None if instance_normals is None else np.ascontiguousarray(instance_normals, dtype=np.float32).reshape(-1, 16)
        self._maps = None      # maps to shade from when set (use_maps); the oracle's own otherwise
        self._own = None

    def use_maps(self, maps):
        """From now on shade from the given IBL maps (same keys and shapes as own_maps()), e.g. the ones the device
        generated: the maps' own parity is a separate test (every texel within 1e-3 of the interval the oracle's
        samples can reach), and a frame shaded by both sides from the same maps differs only by what the frame path
        does differently - not by a map texel whose sample next to the sun fell on the neighbouring source pixel."""
        cube = lambda n: (6, n, n, 4)
        assert maps["sky"].shape == cube(self.sky_size) and maps["irr"].shape == cube(self.irr_size) and maps["lut"].shape == (self.lut_size, self.lut_size, 4)
        assert [p.shape for p in maps["pre"]] == [cube(max(1, self.pre_size >> m)) for m in range(self.pre_mips)]
        assert maps["sky"].dtype == maps["irr"].dtype == np.float32 and maps["lut"].dtype == np.uint16 and all(p.dtype == np.float32 for p in maps["pre"])
        self._maps = maps

    def maps(self):
        return self._maps if self._maps is not None else self.own_maps()

    def own_maps(self):
        if self._own is None:
            h = self.hdr
            sky = ob.equirect_to_cube(self.hdr_data, h.width, h.height, self.sky_size)
            irr = ob.irradiance(self.hdr_data, h.width, h.height, self.irr_size)
            pre_levels = []
            for m in range(self.pre_mips):
                rough = np.float32(m / (self.pre_mips - 1)) if self.pre_mips > 1 else np.float32(0.0)
                pre_levels.append(ob.prefilter_level(self.hdr_chain, h.width, h.height, self.hdr_levels, max(1, self.pre_size >> m), rough))
            _, lut16 = ob.brdf_lut(self.lut_size)
            self._own = {"sky": sky, "irr": irr, "pre": pre_levels, "lut": lut16}
        return self._own

    @contextlib.contextmanager
    def with_maps(self, maps):
        """Shade from the given IBL maps (same keys and shapes as maps()) instead of the oracle's own: with the maps
        another implementation generated, the frames differ only by what the frame path itself does differently."""
        before = self._maps
        self.use_maps(maps)
        try:
            yield self
        finally:
            self._maps = before

    def c_scene(self):
        m = self.maps()
        sc = ob.Scene()
        sc.verts = ob.ptr(self.verts); sc.n_verts = self.verts.size // 8
        if self.indices.dtype == np.uint32:
            sc.indices32 = ob.ptr(self.indices)
        else:
            sc.indices = ob.ptr(self.indices)
        sc.n_indices = self.indices.size
        if self.inst_model is not None:
            sc.n_instances = len(self.inst_model)
            sc.inst_model = ob.ptr(self.inst_model); sc.inst_normal = ob.ptr(self.inst_normal)
        else:
            sc.n_instances = 1
        for i in range(5):
            sc.tex[i] = ob.ptr(self.chains[i]).value
            sc.tex_w[i] = self.images[i].shape[1]; sc.tex_h[i] = self.images[i].shape[0]; sc.tex_levels[i] = self.levels[i]
        self._pre_flat = np.ascontiguousarray(np.concatenate([p.reshape(-1) for p in m["pre"]]))
        sc.irradiance = ob.ptr(m["irr"]); sc.irr_size = self.irr_size
        sc.prefilter = ob.ptr(self._pre_flat); sc.pre_size = self.pre_size; sc.pre_levels = self.pre_mips
        sc.lut_f16 = ob.ptr(m["lut"]); sc.lut_size = self.lut_size
        sc.skybox = ob.ptr(m["sky"]); sc.sky_size = self.sky_size
        return sc

    def draw(self, uniforms, W, H, y0=0, y1=None):
        y1 = H if y1 is None else y1
        sc = self.c_scene()
        u = copy_uniforms(uniforms, ob.FrameUniforms)
        rgba = np.zeros((y1 - y0, W, 4), dtype=np.uint8)
        vis = np.zeros(vis_shape(uniforms, y1 - y0, W), dtype=np.uint32)
        self.lib.orc_draw_frame(C.byref(sc), C.byref(u), W, H, y0, y1, ob.ptr(rgba), ob.ptr(vis))
        return rgba, vis

    def draw_with_margins(self, uniforms, W, H, y0=0, y1=None):
        """The frame plus, per pixel, how close any discrete sampling decision (mip level, min / mag switch, nearest
        texel) came to its switching point (oracle/frame.cpp, note_margin)."""
        y1 = H if y1 is None else y1
        sc = self.c_scene()
        u = copy_uniforms(uniforms, ob.FrameUniforms)
        rgba = np.zeros((y1 - y0, W, 4), dtype=np.uint8)
        vis = np.zeros(vis_shape(uniforms, y1 - y0, W), dtype=np.uint32)
        margin = np.zeros((y1 - y0, W), dtype=np.float32)
        self.lib.orc_draw_frame_margins(C.byref(sc), C.byref(u), W, H, y0, y1, ob.ptr(rgba), ob.ptr(vis), ob.ptr(margin))
        return rgba, vis, margin

    def draw_with_bounds(self, uniforms, W, H, y0=0, y1=None, tangent_ulps=8.0):
        """draw_with_margins plus, per pixel and channel, the interval of 8-bit values the pixel takes when the tangent
        the shader builds from screen-space differences is turned by its own uncertainty (oracle/frame.cpp,
        t_tangent_variant): returns rgba, vis, margin, lo, hi."""
        y1 = H if y1 is None else y1
        sc = self.c_scene()
        u = copy_uniforms(uniforms, ob.FrameUniforms)
        rgba, lo, hi = (np.zeros((y1 - y0, W, 4), dtype=np.uint8) for _ in range(3))
        vis = np.zeros((y1 - y0, W), dtype=np.uint32)
        margin = np.zeros((y1 - y0, W), dtype=np.float32)
        self.lib.orc_draw_frame_bounds(C.byref(sc), C.byref(u), W, H, y0, y1, float(tangent_ulps), ob.ptr(rgba), ob.ptr(vis), ob.ptr(margin), ob.ptr(lo), ob.ptr(hi))
        return rgba, vis, margin, lo, hi

    def map_bounds(self):
        """Reachable intervals of the three nearest-pixel cube generators (oracle/ibl.cpp, *_bounds)."""
        if getattr(self, "_bounds", None) is None:
            h = self.hdr
            pre = []
            for m in range(self.pre_mips):
                rough = np.float32(m / (self.pre_mips - 1)) if self.pre_mips > 1 else np.float32(0.0)
                pre.append(ob.prefilter_level_bounds(self.hdr_chain, h.width, h.height, self.hdr_levels, max(1, self.pre_size >> m), rough))
            self._bounds = {"sky": ob.equirect_to_cube_bounds(self.hdr_data, h.width, h.height, self.sky_size),
                            "irr": ob.irradiance_bounds(self.hdr_data, h.width, h.height, self.irr_size), "pre": pre}
        return self._bounds

    def raster(self, uniforms, W, H, y0=0, y1=None):
        y1 = H if y1 is None else y1
        sc = self.c_scene()
        u = copy_uniforms(uniforms, ob.FrameUniforms)
        vis = np.zeros(vis_shape(uniforms, y1 - y0, W), dtype=np.uint32)
        depth = np.zeros(vis.shape, dtype=np.float32)
        self.lib.orc_raster(C.byref(sc), C.byref(u), W, H, y0, y1, ob.ptr(vis), ob.ptr(depth))
        return vis, depth


class SmallScene:
    """DamagedHelmet with a synthetic sky at sizes the oracle finishes in seconds."""

    def __init__(self):
        self.renderer = None
        self.oracle = None

    @classmethod
    def build(cls, width=320, height=180, hdr_size=(2048, 1024), sky_size=64, pre_size=32, pre_mips=4, lut_size=64, irr_size=8,
              gpu=True, instance_models=None, seed=355, model_path=None, reference_limits=True):
        pkg = _pkg()
        self = cls()
        self.pkg = pkg
        hdr = pkg.synthetic_sky(hdr_size[0], hdr_size[1], seed)
        self.hdr = hdr
        r = pkg.Renderer(width=width, height=height, hdr=hdr, skybox_size=sky_size, irradiance_size=irr_size, prefilter_size=pre_size,
                         prefilter_mips=pre_mips, lut_size=lut_size, instance_models=instance_models, reference_limits=reference_limits,
                         **({"model_path": model_path} if model_path else {}))
        r.load_model()
        self.renderer = r
        g = r.geometries[0]
        t = g.textures
        keys = ("baseColorURI", "metallicRoughnessURI", "normalURI", "emissiveURI", "occlusionURI")
        images = [pkg.load_image_bitmap(t[k]) if t[k] is not None else np.array([[pkg.gltf.DEFAULT_TEXEL[k]]], dtype=np.uint8) for k in keys]
        normals = None
        if instance_models is not None:
            normals = np.stack([pkg.mathx.transpose(pkg.mathx.inverse(m)) for m in np.asarray(instance_models, dtype=np.float32).reshape(-1, 16)])
        self.oracle = OracleScene(g, images, hdr, sky_size, irr_size, pre_size, pre_mips, lut_size, instance_models, normals)
        self.width, self.height = width, height
        if gpu:
            r.init()
            # frames are shaded by both sides from the maps the device generated (see OracleScene.use_maps); tests of
            # the maps themselves ask for oracle.own_maps()
            self.oracle.use_maps(self.gpu_maps())
        return self

    def uniforms(self):
        return self.renderer.frame_uniforms()

    def render_gpu(self, y0=0, y1=None):
        rgba, vis = self.renderer.draw_frame(y0=y0, y1=y1, want_vis=True)
        return rgba, vis

    def render_oracle(self, y0=0, y1=None):
        return self.oracle.draw(self.uniforms(), self.width, self.height, y0, y1)

    def render_oracle_with_margins(self, y0=0, y1=None):
        return self.oracle.draw_with_margins(self.uniforms(), self.width, self.height, y0, y1)

    def gpu_maps(self):
        """The IBL maps as the device holds them, in the layout of OracleScene.maps()."""
        lib, r = self.pkg._lib, self.renderer
        return {"sky": r.download_map(lib.MAP_SKYBOX), "irr": r.download_map(lib.MAP_IRRADIANCE),
                "pre": [r.download_map(lib.MAP_PREFILTER, lvl) for lvl in range(self.oracle.pre_mips)],
                "lut": r.download_map(lib.MAP_BRDF_LUT)}


def psnr(a, b):
    d = a.astype(np.float64) - b.astype(np.float64)
    mse = float((d * d).mean())
    return 99.0 if mse == 0 else 10.0 * np.log10(255.0 * 255.0 / mse)

"""pbr_draw_frame_static, the entry point that allocates nothing: called through the C ABI and compared, bit for
bit, with pbr_draw_frame_device on two calls in a row, the second on a smaller target; then its limits and its
precondition.  Collected last so that a failure here hides no other test."""
import ctypes as C

import numpy as np
import pytest

import parity_utils as pu

pytestmark = pytest.mark.gpu


def _draw(scene, entry, width, height, y0=0, y1=None):
    import torch
    r = scene.renderer
    r.width, r.height = width, height
    y1 = height if y1 is None else y1
    u = r.frame_uniforms()
    out = torch.full((y1 - y0, width, 4), 0x5a, dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()   # the fill runs on torch's stream, the frame on the library's
    rc = entry(r._ctx, C.byref(u), width, height, y0, y1, C.c_void_p(out.data_ptr()))
    scene.pkg._lib.check(r._ctx, rc)
    r.synchronize()
    return out.cpu().numpy()


def test_static_frame_equals_device_frame():
    scene = pu.SmallScene.build(width=640, height=360, sky_size=128, pre_size=32, pre_mips=4, lut_size=64, irr_size=8)
    r, lib = scene.renderer, scene.pkg._lib
    try:
        def device_entry(ctx, u, w, h, y0, y1, out):
            return r._lib.pbr_draw_frame_device(ctx, u, w, h, y0, y1, out, None, None)
        static_entry = r._lib.pbr_draw_frame_static
        u = r.frame_uniforms()
        import torch
        probe = torch.zeros((360, 640, 4), dtype=torch.uint8, device="cuda:0")
        torch.cuda.synchronize()   # the fill runs on torch's stream, the frames on the library's
        # before any ordinary frame the lookup copies of the IBL maps do not exist, and this entry point may not build them
        assert static_entry(r._ctx, C.byref(u), 640, 360, 0, 360, C.c_void_p(probe.data_ptr())) == lib.PBR_E_STATE
        assert b"pbr_draw_frame_device" in r._lib.pbr_last_error(r._ctx)

        free_before_first_frame = torch.cuda.mem_get_info()[0]
        want_big = _draw(scene, device_entry, 640, 360)
        # (the ordinary entry point's first frame allocates its working buffers and the lookup copies: the measurement
        # below does see allocations of this kind)
        assert torch.cuda.mem_get_info()[0] < free_before_first_frame
        got_big = _draw(scene, static_entry, 640, 360)                 # first call
        assert (got_big == want_big).all() and (got_big[..., 3] == 255).all()
        r.camera.update_camera_rotation(-300.0, 120.0)                 # another view, and a smaller target: fewer tiles, shorter lists
        want_small = _draw(scene, device_entry, 250, 130)
        got_small = _draw(scene, static_entry, 250, 130)               # second call finds the first one's contents in the array
        assert (got_small == want_small).all()
        assert (got_small != 0x5a).any(axis=-1).all()                  # every pixel written
        # a row band, and the large target once more after the small one
        band = _draw(scene, static_entry, 250, 130, 32, 96)
        assert (band == want_small[32:96]).all()
        r.camera = scene.pkg.Camera()
        assert (_draw(scene, static_entry, 640, 360) == want_big).all()
        # the property the entry point is named for: the device's free memory is the same before and after its calls
        # (targets allocated beforehand, both target sizes and a band, nothing else running on the device)
        targets = {(w, h): torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda:0") for w, h in ((640, 360), (250, 130))}
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info()[0]
        for (w, h, y0, y1) in ((640, 360, 0, 360), (250, 130, 0, 130), (250, 130, 32, 96), (640, 360, 0, 360)):
            r.width, r.height = w, h
            uu = r.frame_uniforms()
            lib.check(r._ctx, static_entry(r._ctx, C.byref(uu), w, h, y0, y1, C.c_void_p(targets[(w, h)].data_ptr())))
        r.synchronize()
        assert torch.cuda.mem_get_info()[0] == free0
        assert (targets[(640, 360)].cpu().numpy() == want_big).all()
        # the ordinary entry point still works afterwards and agrees
        assert (_draw(scene, device_entry, 640, 360) == want_big).all()

        # limits: 4K is 8160 tiles; the multisampled pipeline is not offered
        big = torch.zeros((2160, 3840, 4), dtype=torch.uint8, device="cuda:0")
        torch.cuda.synchronize()   # the fill runs on torch's stream, the frames on the library's
        r.width, r.height = 3840, 2160
        u = r.frame_uniforms()
        assert static_entry(r._ctx, C.byref(u), 3840, 2160, 0, 2160, C.c_void_p(big.data_ptr())) == lib.PBR_E_ARG
        assert b"tiles" in r._lib.pbr_last_error(r._ctx)
        # ... but a band of it that fits the array is fine and equals the same band of the ordinary frame
        assert static_entry(r._ctx, C.byref(u), 3840, 2160, 1024, 1504, C.c_void_p(big.data_ptr())) == lib.PBR_OK
        r.synchronize()
        ref = torch.zeros((480, 3840, 4), dtype=torch.uint8, device="cuda:0")
        torch.cuda.synchronize()   # the fill runs on torch's stream, the frames on the library's
        lib.check(r._ctx, r._lib.pbr_draw_frame_device(r._ctx, C.byref(u), 3840, 2160, 1024, 1504, C.c_void_p(ref.data_ptr()), None, None))
        r.synchronize()
        assert torch.equal(big[:480], ref)
        u.sampler_index |= 16
        assert static_entry(r._ctx, C.byref(u), 640, 360, 0, 360, C.c_void_p(probe.data_ptr())) == lib.PBR_E_UNSUPPORTED
        assert static_entry(r._ctx, C.byref(u), 640, 360, 0, 360, None) == lib.PBR_E_ARG
    finally:
        r.close()


def test_static_frame_rejects_meshes_beyond_its_capacity():
    """Three helmet instances are 46 356 triangles: more than the fixed working memory holds."""
    pkg = pu._pkg()
    g = pkg.glTFLoader(); g.load(pkg.renderer.DEFAULT_MODEL)
    base = np.asarray(g.get_geometries()[0].model, dtype=np.float32)
    models = np.stack([pkg.mathx.mul(pkg.mathx.translate(pkg.mathx.identity(), [(i - 1) * 2.5, 0.0, 0.0]), base) for i in range(3)])
    scene = pu.SmallScene.build(width=320, height=180, instance_models=models)
    r, lib = scene.renderer, scene.pkg._lib
    try:
        import torch
        out = torch.zeros((180, 320, 4), dtype=torch.uint8, device="cuda:0")
        torch.cuda.synchronize()   # the fill runs on torch's stream, the frames on the library's
        u = r.frame_uniforms()
        lib.check(r._ctx, r._lib.pbr_draw_frame_device(r._ctx, C.byref(u), 320, 180, 0, 180, C.c_void_p(out.data_ptr()), None, None))
        assert r._lib.pbr_draw_frame_static(r._ctx, C.byref(u), 320, 180, 0, 180, C.c_void_p(out.data_ptr())) == lib.PBR_E_ARG
        assert b"triangles" in r._lib.pbr_last_error(r._ctx)
    finally:
        r.close()

"""pbr_draw_views_device, several frames of one scene in one pass: through the C ABI, every view's target must hold
exactly the bytes pbr_draw_frame_device gives the same frame, whatever the number of views, the target size, the row
band, and whatever the context drew before (single frames between passes, passes of other sizes)."""
import ctypes as C

import numpy as np
import pytest

import parity_utils as pu

pytestmark = pytest.mark.gpu


def _orbit_uniforms(scene, n, step=-137.0):
    """n uniform blocks of successive camera positions (the renderer's camera moves on)."""
    r = scene.renderer
    out = []
    for _ in range(n):
        out.append(r.frame_uniforms())
        r.camera.update_camera_rotation(step, 23.0)
    return out


def _single(scene, u, width, height, y0, y1):
    import torch
    r = scene.renderer
    out = torch.full((y1 - y0, width, 4), 0x5a, dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()   # the fill runs on torch's stream, the frame on the library's
    scene.pkg._lib.check(r._ctx, r._lib.pbr_draw_frame_device(r._ctx, C.byref(u), width, height, y0, y1, C.c_void_p(out.data_ptr()), None, None))
    r.synchronize()
    return out.cpu().numpy()


def _views(scene, us, width, height, y0, y1):
    import torch
    r = scene.renderer
    n = len(us)
    outs = [torch.full((y1 - y0, width, 4), 0xa5, dtype=torch.uint8, device="cuda:0") for _ in range(n)]
    torch.cuda.synchronize()   # the fills run on torch's stream, the pass on the library's
    arr = (scene.pkg._lib.FrameUniforms * n)(*us)
    ptrs = (C.c_void_p * n)(*[C.c_void_p(o.data_ptr()) for o in outs])
    scene.pkg._lib.check(r._ctx, r._lib.pbr_draw_views_device(r._ctx, n, arr, width, height, y0, y1, ptrs))
    r.synchronize()
    return [o.cpu().numpy() for o in outs]


def test_views_equal_single_frames():
    scene = pu.SmallScene.build(width=640, height=360, sky_size=128, pre_size=32, pre_mips=4, lut_size=64, irr_size=8)
    r = scene.renderer
    try:
        for (w, h, y0, y1) in ((640, 360, 0, 360), (250, 130, 0, 130), (640, 360, 64, 200), (1280, 720, 0, 720), (33, 31, 0, 31)):
            r.width, r.height = w, h
            for n in (4, 2, 3, 1):
                us = _orbit_uniforms(scene, n)
                want = [_single(scene, u, w, h, y0, y1) for u in us]
                got = _views(scene, us, w, h, y0, y1)
                for v in range(n):
                    assert (got[v] == want[v]).all(), (w, h, y0, y1, n, v, int((got[v] != want[v]).any(axis=-1).sum()))
                    assert (got[v][..., 3] == 255).all()                 # every pixel written
                # and once more straight after: the pass leaves its counters and tile bins ready for the next one
                again = _views(scene, us, w, h, y0, y1)
                for v in range(n):
                    assert (again[v] == want[v]).all(), (w, h, n, v, "second pass")
        # the views of one pass differ from each other (the cameras do), so a mix-up of targets could not pass above
        r.width, r.height = 640, 360
        us = _orbit_uniforms(scene, 4)
        got = _views(scene, us, 640, 360, 0, 360)
        assert all((got[0] != got[v]).any() for v in range(1, 4))
    finally:
        r.close()


def test_views_fall_back_for_other_pipelines():
    """Multisampled views and other samplers are drawn one after the other, with the same bytes as single frames."""
    scene = pu.SmallScene.build(width=320, height=180, sky_size=64, pre_size=32, pre_mips=4, lut_size=64, irr_size=8)
    r = scene.renderer
    try:
        us = _orbit_uniforms(scene, 3)
        for index in (15 | 16, 5):
            for u in us:
                u.sampler_index = index
            want = [_single(scene, u, 320, 180, 0, 180) for u in us]
            got = _views(scene, us, 320, 180, 0, 180)
            for v in range(3):
                assert (got[v] == want[v]).all(), (index, v)
    finally:
        r.close()


def test_views_argument_checks():
    import torch
    scene = pu.SmallScene.build(width=320, height=180, sky_size=64, pre_size=32, pre_mips=4, lut_size=64, irr_size=8)
    r, lib = scene.renderer, scene.pkg._lib
    try:
        us = _orbit_uniforms(scene, 5)
        outs = [torch.zeros((180, 320, 4), dtype=torch.uint8, device="cuda:0") for _ in range(5)]
        torch.cuda.synchronize()
        arr = (lib.FrameUniforms * 5)(*us)
        ptrs = (C.c_void_p * 5)(*[C.c_void_p(o.data_ptr()) for o in outs])
        entry = r._lib.pbr_draw_views_device
        assert entry(r._ctx, 5, arr, 320, 180, 0, 180, ptrs) == lib.PBR_E_ARG          # at most four views
        assert entry(r._ctx, 0, arr, 320, 180, 0, 180, ptrs) == lib.PBR_E_ARG
        assert entry(r._ctx, 2, arr, 320, 180, 0, 180, None) == lib.PBR_E_ARG
        same = (C.c_void_p * 2)(C.c_void_p(outs[0].data_ptr()), C.c_void_p(outs[0].data_ptr()))
        assert entry(r._ctx, 2, arr, 320, 180, 0, 180, same) == lib.PBR_E_ARG          # two views, one target
        assert entry(r._ctx, 2, arr, 320, 180, 1, 180, ptrs) == lib.PBR_E_ARG          # odd first row
        assert entry(r._ctx, 2, arr, 320, 180, 0, 180, ptrs) == lib.PBR_OK
        r.synchronize()
    finally:
        r.close()


def test_views_from_a_fork_and_at_4k():
    """A forked context draws passes of the parent's scene; a 4K pass equals its single frames (sampled rows)."""
    import torch
    scene = pu.SmallScene.build(width=3840, height=2160, sky_size=128, pre_size=32, pre_mips=4, lut_size=64, irr_size=8)
    r, lib = scene.renderer, scene.pkg._lib
    try:
        us = _orbit_uniforms(scene, 4, step=-10.0)
        want = [_single(scene, u, 3840, 2160, 0, 2160) for u in us]
        got = _views(scene, us, 3840, 2160, 0, 2160)
        for v in range(4):
            assert (got[v] == want[v]).all(), v
        fork = r.fork()
        try:
            outs = [torch.zeros((2160, 3840, 4), dtype=torch.uint8, device="cuda:0") for _ in range(4)]
            torch.cuda.synchronize()
            fork.draw_views_device(us, [o.data_ptr() for o in outs])
            fork.synchronize()
            for v in range(4):
                assert (outs[v].cpu().numpy() == want[v]).all(), ("fork", v)
        finally:
            fork.close()
    finally:
        r.close()


def test_views_report_and_repair_working_buffer_overflow():
    """A camera inside a 3 x 3 grid of helmets with every view's set-up buffer and tile-reference list forced far too
    small: the queued pass is reported incomplete by the next synchronize (whichever view overflowed), the buffers are
    enlarged, and a pass drawn after that equals its single frames."""
    import torch
    pkg = pu._pkg()
    lib = pkg._lib
    g = pkg.glTFLoader(); g.load(pkg.renderer.DEFAULT_MODEL)
    base = g.get_geometries()[0].model
    models = np.stack([pkg.mathx.mul(pkg.mathx.translate(pkg.mathx.identity(), [(i - 1) * 2.5, (j - 1) * 2.5, 0.0]), base) for j in range(3) for i in range(3)])
    scene = pu.SmallScene.build(width=480, height=270, instance_models=models)
    r = scene.renderer
    try:
        r.camera.update_camera_distance(430.0, 0.0)          # inside the centre helmet
        r.camera.update_camera_rotation(-300.0, 150.0)
        us = _orbit_uniforms(scene, 3, step=-40.0)
        want = [_single(scene, u, 480, 270, 0, 270) for u in us]
        got = _views(scene, us, 480, 270, 0, 270)            # creates the views' buffers, with the library's own sizing
        for v in range(3):
            assert (got[v] == want[v]).all(), v
        lib.check(r._ctx, r._lib.pbr_debug_set_capacity(r._ctx, 1500, 64))
        reported = 0
        for attempt in range(8):                              # one buffer of one view is enlarged per report
            try:
                got = _views(scene, us, 480, 270, 0, 270)
                break
            except lib.PbrError as e:
                assert "draw it again" in str(e)
                reported += 1
        else:
            raise AssertionError("the pass never completed")
        assert reported >= 2                                  # at least view 0 and one further view overflowed
        for v in range(3):
            assert (got[v] == want[v]).all(), ("after the repair", v)
    finally:
        r.close()

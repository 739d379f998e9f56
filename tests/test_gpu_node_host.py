"""The JavaScript host end to end on the GPU: `node index.js` (N-API addon -> C ABI) must produce
the same frame, byte for byte, as the Python host driving the same library with the same inputs."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

from conftest import ROOT, pkg  # noqa: F401

HOST = os.path.join(ROOT, "wgpu-renderer_amd", "host")

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(shutil.which("node") is None or not os.path.isfile(os.path.join(HOST, "pbr355_node.node")),
                                 reason="node or the built addon is not available")]


@pytest.mark.parametrize("msaa", [False, True])
def test_node_host_frame_equals_python_host_frame(tmp_path, pkg, msaa):
    W, H = 320, 180
    sky = pkg.hdr.synthetic_sky(512, 256, seed=5)
    hdr_path = str(tmp_path / "sky.hdr")
    pkg.hdr.write_hdr_rle(hdr_path, sky.data.reshape(256, 512, 4)[..., :3])
    sizes = dict(skybox_size=128, irradiance_size=16, prefilter_size=32, prefilter_mips=4, lut_size=64)

    out, vis = str(tmp_path / "frame.rgba"), str(tmp_path / "frame.vis")
    cmd = ["node", os.path.join(HOST, "index.js"), "--width", str(W), "--height", str(H), "--hdr", hdr_path, "--out", out, "--vis", vis,
           "--skybox", "128", "--irradiance", "16", "--prefilter", "32", "--prefilter-mips", "4", "--lut", "64", "--orbit", "7", "--msaa", "1" if msaa else "0"]
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=240)
    assert proc.returncode == 0, proc.stderr
    info = json.loads(proc.stdout.strip().splitlines()[-1])
    assert (info["width"], info["height"]) == (W, H)
    js_rgba = np.fromfile(out, dtype=np.uint8).reshape(H, W, 4)
    js_vis = np.fromfile(vis, dtype=np.uint32).reshape((H, W, 4) if msaa else (H, W))

    r = pkg.Renderer(width=W, height=H, hdr=hdr_path, **sizes)
    r.init()
    for _ in range(7):
        r.camera.update_camera_rotation(-10.0, 0.0)
    r.msaa_options["enable"] = msaa
    py_rgba, py_vis, stats = r.draw_frame(want_vis=True, want_stats=True)
    r.close()
    assert np.array_equal(js_vis, py_vis)
    assert np.array_equal(js_rgba, py_rgba)
    assert info["stats"]["coveredPixels"] == stats.covered_pixels > 1000
    assert info["stats"]["setupTriangles"] == stats.setup_triangles

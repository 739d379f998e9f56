"""Host-side mirror of the reference's loaders, camera and option handling (no GPU needed)."""
import ctypes as C
import json
import os
import shutil

import numpy as np
import pytest

import oracle_binding as ob

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HELMET = os.path.join(ROOT, "assets", "DamagedHelmet", "DamagedHelmet.gltf")


def test_gltf_loader_geometry(pkg):
    loader = pkg.glTFLoader()
    loader.load(HELMET)
    geos = loader.get_geometries()
    assert len(geos) == 1
    g = geos[0]
    assert g.vertices.size == 14556 * 8 and g.indices.size == 46356 and g.indices.dtype == np.uint16
    assert g.array_stride == 32
    assert (g.position.format, g.position.offset) == ("float32x3", 0)        # glTF.ts:139-143
    assert (g.tex_coord.format, g.tex_coord.offset) == ("float32x2", 12)     # glTF.ts:153-157
    assert (g.normal.format, g.normal.offset) == ("float32x3", 20)           # glTF.ts:167-171
    raw = open(HELMET.replace(".gltf", ".bin"), "rb").read()
    pos = np.frombuffer(raw, np.float32, 14556 * 3, 92712).reshape(-1, 3)
    nrm = np.frombuffer(raw, np.float32, 14556 * 3, 267384).reshape(-1, 3)
    uv = np.frombuffer(raw, np.float32, 14556 * 2, 442056).reshape(-1, 2)
    v = g.vertices.reshape(-1, 8)
    assert (v[:, 0:3] == pos).all() and (v[:, 3:5] == uv).all() and (v[:, 5:8] == nrm).all()
    assert g.textures["emissive"] == [1.0, 1.0, 1.0]
    assert g.textures["baseColorURI"].endswith("Default_albedo.jpg") and g.textures["occlusionURI"].endswith("Default_AO.jpg")
    # model = I * rotation(+90 degrees about X), glTF.ts:112-130
    lib = ob.load()
    q = np.array([0.7071068286895752, 0.0, -0.0, 0.7071068286895752], np.float32)
    m = np.zeros(16, np.float32)
    lib.orc_node_matrix(None, None, ob.ptr(q), None, ob.ptr(m))
    np.testing.assert_allclose(g.model, m, atol=1e-7)


def test_gltf_loader_errors(pkg, tmp_path):
    loader = pkg.glTFLoader()
    with pytest.raises(RuntimeError, match="Failed to fetch .gltf file"):
        loader.load(str(tmp_path / "missing.gltf"))
    src = json.load(open(HELMET))
    bad = dict(src); bad["asset"] = {"version": "1.0"}
    p = tmp_path / "v1.gltf"; p.write_text(json.dumps(bad))
    with pytest.raises(RuntimeError, match="Unsupported glTF version: 1.0"):
        loader.load(str(p))
    p2 = tmp_path / "short.gltf"; p2.write_text(json.dumps(src))
    (tmp_path / "DamagedHelmet.bin").write_bytes(b"\0" * 100)
    with pytest.raises(RuntimeError, match="Wrong buffer byte length."):
        loader.load(str(p2))
    os.remove(tmp_path / "DamagedHelmet.bin")
    with pytest.raises(RuntimeError, match="Failed to fetch .bin file"):
        loader.load(str(p2))


def test_node_matrix_order_is_scale_translate_rotate(pkg, tmp_path):
    """glTF.ts:116-129 builds I*S*T*R (not T*R*S); a translated, scaled node shows the difference."""
    src = json.load(open(HELMET))
    src["nodes"][0].update({"scale": [2.0, 3.0, 4.0], "translation": [1.0, 0.0, -2.0]})
    (tmp_path / "n.gltf").write_text(json.dumps(src))
    shutil.copy(HELMET.replace(".gltf", ".bin"), tmp_path / "DamagedHelmet.bin")
    loader = pkg.glTFLoader(); loader.load(str(tmp_path / "n.gltf"))
    m = loader.get_geometries()[0].model.reshape(4, 4).T
    np.testing.assert_allclose(m[:3, 3], [2.0, 0.0, -8.0], atol=1e-6)       # translation is scaled
    lib = ob.load()
    s, t, q = np.array([2, 3, 4], np.float32), np.array([1, 0, -2], np.float32), np.array(src["nodes"][0]["rotation"], np.float32)
    o = np.zeros(16, np.float32)
    lib.orc_node_matrix(ob.ptr(s), ob.ptr(t), ob.ptr(q), None, ob.ptr(o))
    np.testing.assert_allclose(loader.get_geometries()[0].model, o, atol=1e-6)


def test_mathx_matches_oracle_matrices(pkg):
    lib = ob.load()
    mx = pkg.mathx
    rng = np.random.default_rng(7)
    a, b = rng.normal(size=16).astype(np.float32), rng.normal(size=16).astype(np.float32)
    o = np.zeros(16, np.float32)
    lib.orc_mat4_mul(ob.ptr(a), ob.ptr(b), ob.ptr(o)); np.testing.assert_allclose(mx.mul(a, b), o, rtol=1e-6, atol=1e-6)
    lib.orc_mat4_inverse(ob.ptr(a), ob.ptr(o)); np.testing.assert_allclose(mx.inverse(a), o, rtol=2e-5, atol=1e-5)
    lib.orc_mat4_transpose(ob.ptr(a), ob.ptr(o)); np.testing.assert_array_equal(mx.transpose(a), o)
    lib.orc_mat4_perspective(0.7, 1.6, 1.0, 100.0, ob.ptr(o)); np.testing.assert_array_equal(mx.perspective(0.7, 1.6, 1.0, 100.0), o)
    eye, tgt, up = rng.normal(size=3).astype(np.float32), rng.normal(size=3).astype(np.float32), np.array([0, 1, 0], np.float32)
    lib.orc_mat4_look_at(ob.ptr(eye), ob.ptr(tgt), ob.ptr(up), ob.ptr(o)); np.testing.assert_array_equal(mx.look_at(eye, tgt, up), o)
    v = rng.normal(size=3).astype(np.float32)
    lib.orc_mat4_scale(ob.ptr(a), ob.ptr(v), ob.ptr(o)); np.testing.assert_array_equal(mx.scale(a, v), o)
    lib.orc_mat4_translate(ob.ptr(a), ob.ptr(v), ob.ptr(o)); np.testing.assert_array_equal(mx.translate(a, v), o)


def test_camera_matches_oracle_camera(pkg):
    lib = ob.load()
    cam = pkg.Camera()
    oc = ob.Camera(); lib.orc_camera_init(C.byref(oc))
    assert list(cam.position) == [0, 0, 5] and list(oc.eye) == [0, 0, 5]     # camera.ts:38-42
    rng = np.random.default_rng(11)
    for i in range(40):
        dx, dy = float(rng.integers(-30, 30)), float(rng.integers(-30, 30))
        kind = i % 4
        if kind in (0, 1):
            cam.update_camera_rotation(dx, dy); lib.orc_camera_rotate(C.byref(oc), dx, dy)
        elif kind == 2:
            cam.update_camera_distance(dx, dy); lib.orc_camera_dolly(C.byref(oc), dx, dy)
        else:
            cam.update_camera_target(dx, dy); lib.orc_camera_track(C.byref(oc), dx, dy)
        np.testing.assert_allclose(cam.eye, list(oc.eye), atol=2e-5)
        np.testing.assert_allclose(cam.up, list(oc.up), atol=2e-5)
        v = np.zeros(16, np.float32); lib.orc_camera_view(C.byref(oc), ob.ptr(v))
        np.testing.assert_allclose(cam.view, v, atol=5e-5)


def test_orbit_is_360_one_degree_steps(pkg):
    """updateCameraRotation(-10, 0) tumbles by +1 degree about the up axis (camera.ts:166-169)."""
    cam = pkg.Camera()
    cam.update_camera_rotation(-10.0, 0.0)
    np.testing.assert_allclose(np.degrees(np.arctan2(cam.eye[0], cam.eye[2])), 1.0, atol=1e-4)
    for _ in range(359):
        cam.update_camera_rotation(-10.0, 0.0)
    np.testing.assert_allclose(cam.eye, [0, 0, 5], atol=2e-3)
    np.testing.assert_allclose(np.linalg.norm(cam.eye), 5.0, atol=1e-3)


def test_hdr_loader_roundtrip_and_errors(pkg):
    lib = ob.load()
    rng = np.random.default_rng(3)
    rgb = np.exp(rng.normal(0, 2.0, size=(12, 40, 3)))
    data = pkg.write_hdr_rle(None, rgb)
    loader = pkg.HDRLoader(); loader.load_bytes(data)
    hdr = loader.hdr
    assert (hdr.width, hdr.height) == (40, 12) and hdr.data.dtype == np.float32 and hdr.data.size == 40 * 12 * 4
    got = hdr.data.reshape(12, 40, 4)
    assert (got[..., 3] == 1.0).all()
    np.testing.assert_allclose(got[..., :3].max(-1), rgb.max(-1), rtol=1 / 128)   # 8-bit shared-exponent mantissas
    # the oracle's decoder follows hdr.ts independently
    buf = np.frombuffer(data, np.uint8)
    w, h = C.c_int(), C.c_int()
    assert lib.orc_hdr_dims(ob.ptr(buf), len(data), C.byref(w), C.byref(h)) == 0 and (w.value, h.value) == (40, 12)
    out = np.zeros(40 * 12 * 4, np.float32)
    assert lib.orc_hdr_decode(ob.ptr(buf), len(data), ob.ptr(out)) == 0
    np.testing.assert_array_equal(out, hdr.data)
    # run-length runs (count > 128) decode too
    w2 = 16
    line = bytes([2, 2, 0, w2]) + b"".join(bytes([128 + w2, v]) for v in (10, 20, 30, 129))
    blob = b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 1 +X 16\n" + line
    loader.load_bytes(blob)
    np.testing.assert_allclose(loader.hdr.data.reshape(16, 4)[0], [10 * 2 / 255, 20 * 2 / 255, 30 * 2 / 255, 1.0], rtol=1e-6)
    with pytest.raises(RuntimeError, match="HDRLoader: Unsupported .hdr format."):
        loader.load_bytes(b"#?RADIANCE\nFORMAT=32-bit_float\n\n-Y 1 +X 16\n")
    with pytest.raises(RuntimeError, match="HDRLoader: No resolution found."):
        loader.load_bytes(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y x +X 16\n")
    with pytest.raises(RuntimeError, match="HDRLoader: Only support RLE format."):
        loader.load_bytes(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 1 +X 16\n" + bytes(64))
    with pytest.raises(RuntimeError, match="Failed to fetch .hdr file"):
        loader.load("/nonexistent/x.hdr")


def test_synthetic_sky_is_deterministic(pkg):
    a = pkg.synthetic_sky(256, 128)
    b = pkg.synthetic_sky(256, 128)
    assert (a.data == b.data).all() and a.data.dtype == np.float32
    img = a.data.reshape(128, 256, 4)
    assert (img[..., 3] == 1).all() and img[..., :3].min() > 0
    assert img[..., :3].max() > 1000          # the sun lobe
    assert img[:8, :, 2].mean() > 3 * img[-8:, :, 2].mean()   # brighter at the top rows


def test_frame_uniforms_defaults(pkg):
    r = pkg.Renderer(width=1280, height=720)
    r.load_model()
    u = r.frame_uniforms()
    assert u.sampler_index == 15 and u.use_mipmaps == 1                      # renderer.ts:1762-1768, :1927-1932
    assert list(u.view_pos)[:3] == [0, 0, 5]
    assert [list(l)[:3] for l in u.light_pos] == [[10, 10, 10], [-10, 10, 10], [10, -10, 10], [-10, -10, 10]]
    assert all(list(c)[:3] == [300, 300, 300] for c in u.light_color)
    np.testing.assert_allclose(u.projection[0], 2.41421356 / (1280 / 720), rtol=1e-6)
    m = np.array(list(u.model), np.float32).reshape(4, 4)
    n = np.array(list(u.normal), np.float32).reshape(4, 4)
    np.testing.assert_allclose(n, m, atol=1e-6)                              # a pure rotation: transpose(inverse(M)) = M
    pv = pkg.mathx.mul(np.array(list(u.projection), np.float32), np.array(list(u.view), np.float32))
    np.testing.assert_allclose(pkg.mathx.mul(pv, np.array(list(u.inv_view_proj), np.float32)).reshape(4, 4), np.eye(4), atol=1e-4)
    r.texture_options.update(addressModeU="clamp-to-edge", magFilter="nearest", mipmaps=False)
    u = r.frame_uniforms()
    assert u.sampler_index == 2 + 8 and u.use_mipmaps == 0


def test_sharding_partitions(pkg):
    sh = pkg._lib and __import__("wgpu_renderer_amd.sharding", fromlist=["x"])
    for world in (1, 2, 3, 4, 8):
        frames = sorted(f for r in range(world) for f in sh.orbit_frames(r, world, 360))
        assert frames == list(range(360))
        total = sh.prefilter_total_texels(2048, 10)
        assert total == sum(6 * (2048 >> m) ** 2 for m in range(10)) == 33554400 + 0 * world
        ranges = [sh.prefilter_shard_texels(2048, 10, r, world) for r in range(world)]
        # level 0 (roughness 0, one sample per texel) is filtered by every rank of several and belongs to no shard
        assert ranges[0][0] == sh.prefilter_replicated_texels(2048, 10, world) == (0 if world == 1 else 6 * 2048 ** 2) and ranges[-1][1] == total
        assert all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))
        for height in (4320, 2160, 721, 2):
            rows = [sh.strip_rows(height, r, world) for r in range(world)]
            assert rows[0][0] == 0 and rows[-1][1] == height
            assert all(rows[i][1] == rows[i + 1][0] for i in range(world - 1))
            assert all(y0 % 2 == 0 for y0, _ in rows)


def kept_samples_f64(roughness, n=1024):
    """Samples of prefiltermap.wgsl:131-168 with N.L > 0 when N = V, evaluated in float64 from the WGSL alone."""
    i = np.arange(n, dtype=np.uint32)
    bits = i.copy()
    bits = (bits << 16) | (bits >> 16)
    bits = ((bits & 0x55555555) << 1) | ((bits & 0xAAAAAAAA) >> 1)
    bits = ((bits & 0x33333333) << 2) | ((bits & 0xCCCCCCCC) >> 2)
    bits = ((bits & 0x0F0F0F0F) << 4) | ((bits & 0xF0F0F0F0) >> 4)
    bits = ((bits & 0x00FF00FF) << 8) | ((bits & 0xFF00FF00) >> 8)
    xi_y = bits.astype(np.float64) * 2.3283064365386963e-10
    a = float(roughness) ** 2
    cos_theta = np.sqrt((1.0 - xi_y) / (1.0 + (a * a - 1.0) * xi_y))
    # H = (sin t cos p, sin t sin p, cos t), V = N = (0, 0, 1): L.z = 2 (V.H) H.z - 1 = 2 cos^2 t - 1
    return int((2.0 * cos_theta * cos_theta - 1.0 > 0.0).sum())


def test_prefilter_shards_carry_equal_cost(pkg):
    """The shards of the large prefilter (2048^2, 10 levels) must be balanced in work, not in texels: level 0 is three
    quarters of the texels and, at roughness 0, one sample each.  Cost model: kept samples per texel (counted here
    in float64 independently of the library) times the per-sample factors measured on MI355X for the +-Y faces (their
    middle rows, around the poles, apart) and the rougher levels, for the lane split of the small levels and for the floor of a launch
    (profiles/r03_prefilter_shards.json)."""
    sh = __import__("wgpu_renderer_amd.sharding", fromlist=["x"])
    size, mips = 2048, 10
    kept = [1 if m == 0 else kept_samples_f64(np.float32(m / (mips - 1))) for m in range(mips)]
    lib_kept = sh.prefilter_kept_samples(mips)
    assert lib_kept[0] == 1 and all(abs(a - b) <= 2 for a, b in zip(kept, lib_kept)), (kept, lib_kept)

    def texel_cost(m, face, middle):   # the library's cost model restated: picoseconds per texel
        r = m / (mips - 1)
        level_texels = 6 * (size >> m) ** 2
        pole = face in (2, 3)
        scatter = ((2.71 if middle else 1.03) if pole else 1.0) if r <= 0.17 else ((4.70 if middle else 1.25) if pole else 1.22) if r <= 0.28 else (1.41 if pole else 1.33)
        shape = 1.0 if level_texels >= 1 << 18 else 1.85 if level_texels >= 1 << 14 else 1.60
        return (kept[m] + 8) * 1.22 * scatter * shape + 8.5e6 / level_texels

    # (first texel, texel count, cost per texel, samples per texel) of the three row parts of every face of every level:
    # the rows in front of the face's middle quarter, the middle quarter (on the +-Y faces: around the pole), the rest
    segments = []
    first = 0
    for m in range(mips):
        s = size >> m
        rows = (s * 3 // 8, s * 5 // 8 - s * 3 // 8, s - s * 5 // 8)
        for face in range(6):
            for part in range(3):
                segments.append((first, rows[part] * s, texel_cost(m, face, part == 1), kept[m]))
                first += rows[part] * s
    total_texels = first

    def load(b, e, column):
        return sum(max(0, min(e, f + n) - max(b, f)) * seg[column] for seg in segments for f, n in [seg[:2]])

    level0 = 6 * size * size
    for world in (2, 4, 8, 3, 7):
        cuts = sh.prefilter_cuts(size, mips, world)
        assert cuts[0] == level0 and cuts[-1] == total_texels and all(a < b for a, b in zip(cuts, cuts[1:]))
        costs = [load(b, e, 2) for b, e in zip(cuts, cuts[1:])]
        assert max(costs) / (sum(costs) / world) <= 1.02, (world, costs)
        samples = [load(b, e, 3) for b, e in zip(cuts, cuts[1:])]          # without the measured factors: still no idle shard
        assert max(samples) / (sum(samples) / world) <= 1.6, (world, samples)
        assert [sh.prefilter_shard_texels(size, mips, r, world) for r in range(world)] == list(zip(cuts, cuts[1:]))
    # the reference's own size: five levels of 128^2
    for world in (2, 4, 8):
        cuts = sh.prefilter_cuts(128, 5, world)
        assert cuts[0] == 6 * 128 * 128 and cuts[-1] == sh.prefilter_total_texels(128, 5) and all(a <= b for a, b in zip(cuts, cuts[1:]))
    assert sh.prefilter_cuts(128, 5, 1) == [0, sh.prefilter_total_texels(128, 5)]
    assert sh.prefilter_cuts(16, 1, 4) == [1536] * 5      # a single level (roughness 0): every rank filters it, nothing is sent


def test_prefilter_pieces_tile_every_shard(pkg):
    """The parts a rank filters and sends one after the other are consecutive, cover its range exactly, and are the same
    list on every rank (each rank computes it from the cuts alone)."""
    sh = __import__("wgpu_renderer_amd.sharding", fromlist=["x"])
    for size, mips, world, pieces in ((2048, 10, 8, 2), (128, 5, 3, 2), (128, 5, 2, 3), (16, 1, 4, 2), (128, 5, 2, 1)):
        cuts = sh.prefilter_cuts(size, mips, world)
        parts = sh.prefilter_pieces(cuts, pieces)
        assert len(parts) == pieces and all(len(p) == world for p in parts)
        for k in range(world):
            assert parts[0][k][0] == cuts[k] and parts[-1][k][1] == cuts[k + 1]
            for j in range(pieces):
                b, e = parts[j][k]
                assert b <= e and (j == 0 or parts[j - 1][k][1] == b)


def test_prefilter_exchange_carries_only_what_is_worth_sending(pkg):
    """Equal cost is not equal bytes: level 0 is 403 MB of the 2048^2 x 10 map's 537 MB and 0.3 ms of its 15 ms on one
    GPU, so it must not cross a link.  What a rank contributes to the exchange, for both ways of exchanging: at most
    25 MB at eight ranks (and in proportion at fewer), at most 1.3 times the mean range, the chunks of the
    equal-chunk all-gather included; everything a rank does not receive it filters itself."""
    sh = __import__("wgpu_renderer_amd.sharding", fromlist=["x"])
    size, mips = 2048, 10
    total = sh.prefilter_total_texels(size, mips)
    for world in (2, 4, 8):
        cuts = sh.prefilter_cuts(size, mips, world)
        own = [16 * (e - b) for b, e in zip(cuts, cuts[1:])]
        assert sum(own) == 16 * (total - 6 * size * size) == 134_217_216
        for how in ("p2p", "collective"):
            sent = sh.prefilter_exchange_bytes(size, mips, world, how)
            assert len(sent) == world and all(s >= o for s, o in zip(sent, own))
            assert max(sent) <= 25e6 * 8 / world, (world, how, sent)
        # the default exchange sends the ranges as they are: on average exactly the mean; the equal-chunk all-gather
        # pads every rank's range to the largest, which equal cost puts up to a third above the mean
        assert sh.prefilter_exchange_bytes(size, mips, world) == sh.prefilter_exchange_bytes(size, mips, world, "p2p") == own
        assert sum(own) / world * 1.0 <= max(own) <= 1.4 * sum(own) / world
    assert sh.prefilter_exchange_bytes(size, mips, 1) == [16 * total]


def test_gltf_loader_without_reference_limits(tmp_path, pkg):
    """accessor.byteOffset, bufferView.byteStride, 32-bit indices and a texture-less material: read per
    the glTF 2.0 specification when asked to, refused or misread by the reference-faithful default."""
    import gltf_samples
    path, expected_vertices, expected_indices = gltf_samples.write_interleaved_quad(str(tmp_path))
    loader = pkg.glTFLoader(reference_limits=False)
    loader.load(path)
    g = loader.get_geometries()[0]
    assert np.array_equal(g.vertices, expected_vertices)
    assert np.array_equal(g.indices, expected_indices) and g.indices.dtype == np.uint16
    assert all(g.textures[k] is None for k in ("baseColorURI", "metallicRoughnessURI", "normalURI", "emissiveURI", "occlusionURI"))
    assert g.textures["emissive"] == [0.0, 0.0, 0.0]
    # node matrix = I * S * T * R (glTF.ts:112-130): translation then the 45 degree turn about y
    m = np.asarray(g.model, dtype=np.float64).reshape(4, 4).T
    assert np.allclose(m[:3, 3], [0.25, 0, 0]) and np.allclose(m[0, 0], np.cos(np.pi / 4), atol=1e-6) and np.allclose(m[0, 2], np.sin(np.pi / 4), atol=1e-6)
    # the same helmet either way
    a, b = pkg.glTFLoader(), pkg.glTFLoader(reference_limits=False)
    a.load(pkg.renderer.DEFAULT_MODEL); b.load(pkg.renderer.DEFAULT_MODEL)
    ga, gb = a.get_geometries()[0], b.get_geometries()[0]
    assert np.array_equal(ga.vertices, gb.vertices) and np.array_equal(ga.indices, gb.indices) and ga.textures == gb.textures
    # the reference's loader has no material guards (glTF.ts:228-252): the default mirrors its failure
    with pytest.raises((KeyError, RuntimeError, ValueError)):
        pkg.glTFLoader().load(path)


def test_gltf_loader_joins_primitives_with_wide_indices(tmp_path, pkg):
    """Beyond 65 536 vertices the general loader hands over 32-bit indices; all triangle primitives of the mesh
    become one geometry, later primitives' indices shifted past the earlier vertices, other modes skipped."""
    import gltf_samples
    path, verts, indices = gltf_samples.write_large_two_primitive_mesh(str(tmp_path))
    loader = pkg.glTFLoader(reference_limits=False)
    loader.load(path)
    g = loader.get_geometries()[0]
    assert g.indices.dtype == np.uint32 and np.array_equal(g.indices, indices)
    assert np.array_equal(g.vertices.reshape(-1, 8), verts) and verts.shape[0] == 272 * 272 + 4 > 65536
    assert int(g.indices[-6:].min()) == 272 * 272
    # a small mesh of several primitives stays at 16 bits
    os.makedirs(tmp_path / "small")
    small_path, sverts, sindices = gltf_samples.write_large_two_primitive_mesh(str(tmp_path / "small"), n=16)
    small = pkg.glTFLoader(reference_limits=False)
    small.load(small_path)
    gs = small.get_geometries()[0]
    assert gs.indices.dtype == np.uint16 and np.array_equal(gs.indices, sindices) and np.array_equal(gs.vertices.reshape(-1, 8), sverts)

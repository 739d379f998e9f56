"""The JavaScript host (wgpu-renderer_amd/host) against the Python host: loaders, camera, uniform
block and JPEG decode must agree bit for bit; the N-API addon must load and fail loudly without a GPU."""
import ctypes as C
import glob
import hashlib
import io
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

from conftest import ROOT, pkg  # noqa: F401

HOST = os.path.join(ROOT, "wgpu-renderer_amd", "host")
MODEL = os.path.join(ROOT, "assets", "DamagedHelmet", "DamagedHelmet.gltf")
ALBEDO = os.path.join(ROOT, "assets", "DamagedHelmet", "Default_albedo.jpg")

needs_node = pytest.mark.skipif(shutil.which("node") is None or not os.path.isfile(os.path.join(HOST, "pbr355_node.node")),
                                reason="node or the built addon is not available")


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _hex(a):
    return np.ascontiguousarray(a).tobytes().hex()


@pytest.fixture(scope="module")
def small_hdr(tmp_path_factory, pkg):
    sky = pkg.hdr.synthetic_sky(64, 32, seed=3)
    path = str(tmp_path_factory.mktemp("hdr") / "sky.hdr")
    pkg.hdr.write_hdr_rle(path, sky.data.reshape(32, 64, 4)[..., :3])
    return path


@pytest.fixture(scope="module")
def older_hdrs(tmp_path_factory, pkg):
    """The same image in the two pre-RLE scanline encodings (with a run of equal pixels in it)."""
    sky = pkg.hdr.synthetic_sky(64, 32, seed=3).data.reshape(32, 64, 4)[..., :3].copy()
    sky[5, 10:40] = sky[5, 10]
    sky[6, :] = sky[6, 0]
    d = tmp_path_factory.mktemp("hdr_old")
    paths = (str(d / "flat.hdr"), str(d / "old_rle.hdr"), str(d / "new_rle.hdr"))
    pkg.hdr.write_hdr_flat(paths[0], sky)
    pkg.hdr.write_hdr_flat(paths[1], sky, old_rle=True)
    pkg.hdr.write_hdr_rle(paths[2], sky)
    return paths


@pytest.fixture(scope="module")
def general_gltf(tmp_path_factory):
    import gltf_samples
    return gltf_samples.write_interleaved_quad(str(tmp_path_factory.mktemp("gltf")))[0]


@pytest.fixture(scope="module")
def large_gltf(tmp_path_factory):
    import gltf_samples
    return gltf_samples.write_large_two_primitive_mesh(str(tmp_path_factory.mktemp("large")))[0]


@pytest.fixture(scope="module")
def js_dump(small_hdr, older_hdrs, general_gltf, large_gltf):
    proc = subprocess.run(["node", os.path.join(HOST, "test", "dump_host.js"), MODEL, small_hdr, ALBEDO, older_hdrs[0], older_hdrs[1], general_gltf, large_gltf], stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, text=True, timeout=120)
    assert proc.returncode == 0, proc.stderr
    return json.loads(proc.stdout)


@needs_node
def test_gltf_loader_matches(js_dump, pkg):
    loader = pkg.glTFLoader()
    loader.load(MODEL)
    g = loader.get_geometries()[0]
    j = js_dump["geometry"]
    assert j["vertices"] == _sha(g.vertices.astype(np.float32))
    assert j["indices"] == _sha(g.indices.astype(np.uint16))
    assert j["model"] == _hex(np.asarray(g.model, dtype=np.float32))
    assert j["arrayStride"] == g.array_stride == 32
    assert (j["position"], j["texCoord"], j["normal"]) == ({"format": "float32x3", "offset": 0}, {"format": "float32x2", "offset": 12},
                                                          {"format": "float32x3", "offset": 20})
    for key in ("baseColorURI", "metallicRoughnessURI", "normalURI", "emissiveURI", "occlusionURI"):
        assert j["textures"][key] == g.textures[key]
    assert j["textures"]["emissive"] == list(g.textures["emissive"])


@needs_node
def test_general_gltf_loader_matches(js_dump, general_gltf, pkg):
    loader = pkg.glTFLoader(reference_limits=False)
    loader.load(general_gltf)
    g = loader.get_geometries()[0]
    j = js_dump["generalGeometry"]
    assert j["vertices"] == _sha(g.vertices.astype(np.float32)) and j["indices"] == _sha(g.indices.astype(np.uint16))
    assert j["model"] == _hex(np.asarray(g.model, dtype=np.float32))
    assert all(j["textures"][k] is None for k in ("baseColorURI", "metallicRoughnessURI", "normalURI", "emissiveURI", "occlusionURI"))
    assert j["textures"]["emissive"] == [0, 0, 0]


@needs_node
def test_large_multi_primitive_gltf_matches(js_dump, large_gltf, pkg):
    loader = pkg.glTFLoader(reference_limits=False)
    loader.load(large_gltf)
    g = loader.get_geometries()[0]
    j = js_dump["largeGeometry"]
    assert j["indexBytes"] == 4 and g.indices.dtype == np.uint32
    assert j["vertices"] == _sha(g.vertices.astype(np.float32)) and j["indices"] == _sha(g.indices)
    assert j["model"] == _hex(np.asarray(g.model, dtype=np.float32))


@needs_node
def test_hdr_loader_matches(js_dump, small_hdr, pkg):
    loader = pkg.HDRLoader()
    loader.load(small_hdr)
    assert (js_dump["hdr"]["width"], js_dump["hdr"]["height"]) == (64, 32)
    assert js_dump["hdr"]["data"] == _sha(loader.hdr.data.astype(np.float32))


def test_hdr_older_encodings_python(older_hdrs, pkg):
    """Flat and old-RLE scanlines decode to the same pixels as the new-RLE file of the same image; the
    default loader keeps the reference's refusal."""
    ref = pkg.HDRLoader()
    ref.load(older_hdrs[2])
    assert os.path.getsize(older_hdrs[1]) < os.path.getsize(older_hdrs[0])     # the runs were really encoded
    for path in older_hdrs[:2]:
        loader = pkg.HDRLoader(reference_limits=False)
        loader.load(path)
        assert (loader.width, loader.height) == (64, 32)
        assert np.array_equal(loader.hdr.data, ref.hdr.data)
        with pytest.raises(RuntimeError, match="Only support RLE format"):     # hdr.ts:81
            pkg.HDRLoader().load(path)


ZERO_EXPONENT_HDR = (b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 1 +X 8\n" +
                     bytes([2, 2, 0, 8, 0x88, 200, 0x88, 100, 0x88, 50, 0x84, 0, 0x84, 129]))   # 4 pixels (200,100,50,e=0), 4 with e=129


def test_hdr_zero_exponent_python(pkg):
    """hdr.ts:105-116 has no special case for an exponent byte of 0 and returns mantissa * 2^-128 / 255; the Radiance
    format defines such a pixel as black.  The default loader keeps the reference's value, the opt-in one the format's."""
    ref = pkg.HDRLoader(); ref.load_bytes(ZERO_EXPONENT_HDR)
    fmt = pkg.HDRLoader(reference_limits=False); fmt.load_bytes(ZERO_EXPONENT_HDR)
    a, b = ref.hdr.data.reshape(8, 4), fmt.hdr.data.reshape(8, 4)
    tiny = np.float32(2.0 ** -128 / 255.0)
    want = (np.array([200, 100, 50, 0], np.float64) * 2.0 ** -128 / 255.0 + np.array([0, 0, 0, 1])).astype(np.float32)
    assert np.array_equal(a[0], want) and 0 < a[0, 2] < 1e-36 and tiny > 0
    assert np.array_equal(b[0], [0.0, 0.0, 0.0, 1.0]) and np.array_equal(b[:4], np.tile(b[0], (4, 1)))
    assert np.array_equal(a[4:], b[4:]) and np.allclose(a[4], [200 * 2 / 255, 100 * 2 / 255, 50 * 2 / 255, 1.0])


@needs_node
def test_hdr_zero_exponent_js(js_dump, pkg):
    ref = pkg.HDRLoader(); ref.load_bytes(ZERO_EXPONENT_HDR)
    fmt = pkg.HDRLoader(reference_limits=False); fmt.load_bytes(ZERO_EXPONENT_HDR)
    for got, want in zip(js_dump["hdrZero"], (ref, fmt)):
        w = want.hdr.data.reshape(8, 4)
        assert np.array_equal(np.array(got, np.float32), np.concatenate([w[0], w[4]]).astype(np.float32))


@needs_node
def test_hdr_older_encodings_js(js_dump, older_hdrs, pkg):
    ref = pkg.HDRLoader()
    ref.load(older_hdrs[2])
    for got in js_dump["hdrOlder"]:
        assert (got["width"], got["height"]) == (64, 32)
        assert got["data"] == _sha(ref.hdr.data.astype(np.float32))


@needs_node
def test_jpeg_decode_matches(js_dump, pkg):
    img = pkg.load_image_bitmap(ALBEDO)
    assert (js_dump["image"]["width"], js_dump["image"]["height"]) == (img.shape[1], img.shape[0])
    assert js_dump["image"]["data"] == _sha(img)


@needs_node
def test_camera_and_uniforms_match(js_dump, pkg):
    r = pkg.Renderer(width=1280, height=720)
    r.load_model()
    got = [bytes(r.frame_uniforms()).hex()]
    for _ in range(3):
        r.camera.update_camera_rotation(-10, 4)
    got.append(bytes(r.frame_uniforms()).hex())
    r.camera.update_camera_distance(30, -5)
    r.camera.update_camera_target(12, -7)
    r.texture_options.update(minFilter="nearest", addressModeU="clamp-to-edge", mipmaps=False)
    got.append(bytes(r.frame_uniforms()).hex())
    assert js_dump["uniforms"] == got
    cam = js_dump["camera"]
    assert cam["eye"] == _hex(r.camera.eye) and cam["up"] == _hex(r.camera.up) and cam["right"] == _hex(r.camera.right)
    assert cam["target"] == _hex(r.camera.target)
    assert cam["distance"] == pytest.approx(r.camera.distance, abs=0)
    assert js_dump["mips"] == [12, 1, 13, 9]      # utils.ts:20-24


@needs_node
def test_error_messages(js_dump):
    e = js_dump["errors"]
    assert e["gltfMissing"] == "Failed to fetch .gltf file: /nonexistent/model.gltf"      # glTF.ts:80
    assert e["hdrMissing"] == "Failed to fetch .hdr file: /nonexistent/sky.hdr"           # hdr.ts:31
    assert e["hdrFormat"] == "HDRLoader: Unsupported .hdr format."                         # hdr.ts:58
    assert e["hdrFlat"] == "HDRLoader: Only support RLE format."                           # hdr.ts:81
    # no software fallback: without a usable device the renderer throws
    assert e["noDevice"] and ("device" in e["noDevice"].lower())


def _jpeg_lib():
    path = os.path.join(HOST, "libpbrhost.so")
    if not os.path.isfile(path):
        pytest.skip("libpbrhost.so is not built")
    lib = C.CDLL(path)
    lib.pbr_host_jpeg_decode.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                         C.c_char_p, C.c_size_t]
    lib.pbr_host_jpeg_decode.restype = C.c_int
    lib.pbr_host_free.argtypes = [C.c_void_p]
    return lib


def _decode(lib, data):
    out, w, h, err = C.POINTER(C.c_uint8)(), C.c_int32(), C.c_int32(), C.create_string_buffer(200)
    rc = lib.pbr_host_jpeg_decode(data, len(data), C.byref(out), C.byref(w), C.byref(h), err, 200)
    if rc:
        raise RuntimeError(err.value.decode())
    a = np.ctypeslib.as_array(out, shape=(h.value, w.value, 4)).copy()
    lib.pbr_host_free(out)
    return a


def test_jpeg_decoder_helmet_textures_bit_exact():
    """The five material textures (four baseline 4:2:0, one progressive 4:4:4) decode to the same
    bytes as Pillow's libjpeg-turbo, the decoder family browsers use for createImageBitmap."""
    from PIL import Image
    lib = _jpeg_lib()
    files = sorted(glob.glob(os.path.join(ROOT, "assets", "DamagedHelmet", "*.jpg")))
    assert len(files) == 5
    for f in files:
        with open(f, "rb") as fh:
            data = fh.read()
        ref = np.asarray(Image.open(f).convert("RGBA"))
        assert np.array_equal(_decode(lib, data), ref), f


def test_jpeg_decoder_variants_bit_exact():
    """Sub-sampling modes, progressive / baseline, restart intervals, grayscale and ragged sizes."""
    from PIL import Image, ImageFilter
    lib = _jpeg_lib()
    rng = np.random.default_rng(11)
    cases = 0
    for (w, h) in [(1, 1), (7, 5), (16, 16), (17, 33), (100, 75), (255, 3), (3, 200)]:
        noise = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        soft = np.asarray(Image.fromarray(noise).filter(ImageFilter.GaussianBlur(2)))
        for img in (noise, soft):
            for mode in ("RGB", "L"):
                im = Image.fromarray(img).convert(mode)
                for sub in ([0, 1, 2] if mode == "RGB" else [None]):
                    for prog in (False, True):
                        for rst in (0, 3):
                            kw = {"quality": 85 if prog else 40, "progressive": prog}
                            if sub is not None:
                                kw["subsampling"] = sub
                            if rst:
                                kw["restart_marker_blocks"] = rst
                            buf = io.BytesIO()
                            im.save(buf, "JPEG", **kw)
                            data = buf.getvalue()
                            ref = np.asarray(Image.open(io.BytesIO(data)).convert("RGBA"))
                            assert np.array_equal(_decode(lib, data), ref), (w, h, mode, sub, prog, rst)
                            cases += 1
    assert cases == 7 * 2 * (3 + 1) * 2 * 2


def test_jpeg_decoder_rejects_garbage():
    lib = _jpeg_lib()
    with pytest.raises(RuntimeError, match="Not a JPEG"):
        _decode(lib, b"\x89PNG\r\n\x1a\n" + b"\0" * 64)
    with pytest.raises(RuntimeError):
        _decode(lib, b"\xff\xd8\xff\xd9")

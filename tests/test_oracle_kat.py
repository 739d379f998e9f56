"""Known-answer tests that pin the CPU oracle.  The reference ships no tests or golden vectors
(package.json:7), so every value here is derived by hand from the WGSL / TypeScript and computed
in the test independently of the oracle's code."""
import ctypes as C
import json
import math
import os

import numpy as np
import pytest

import oracle_binding as ob

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HELMET = os.path.join(ROOT, "assets", "DamagedHelmet")


def test_radical_inverse():
    # prefiltermap.wgsl:32-39
    lib = ob.load()
    assert lib.orc_radical_inverse(1) == 0.5
    assert lib.orc_radical_inverse(2) == 0.25
    assert lib.orc_radical_inverse(3) == 0.75
    assert lib.orc_radical_inverse(1023) == 1023.0 / 1024.0
    assert lib.orc_radical_inverse(0) == 0.0


def test_irradiance_trip_counts():
    # irradiancemap.wgsl:75-76 with f32 loop counters: 252 x 63 = 15876 taps
    lib = ob.load()
    a, b = C.c_int(), C.c_int()
    lib.orc_irradiance_trip_counts(C.byref(a), C.byref(b))
    assert (a.value, b.value) == (252, 63)
    phi, n = np.float32(0), 0
    while phi < np.float32(2.0) * np.float32(3.14159265359):
        phi = np.float32(phi + np.float32(0.025)); n += 1
    assert n == 252


def test_f16_conversion_matches_ieee():
    lib = ob.load()
    rng = np.random.default_rng(1)
    vals = np.concatenate([rng.normal(0, 1, 2000), rng.normal(0, 1e-6, 500), rng.normal(0, 7e4, 500),
                           [0.0, -0.0, 1.0, 65504.0, 65520.0, 1e-8, 6e-8, np.inf, -np.inf]]).astype(np.float32)
    for v in vals:
        want = np.float32(v).astype(np.float16).view(np.uint16)
        assert lib.orc_f32_to_f16(float(v)) == int(want), v
    assert math.isnan(lib.orc_f16_to_f32(lib.orc_f32_to_f16(float("nan"))))
    for h in [0, 1, 0x03ff, 0x0400, 0x3c00, 0x7bff, 0x8001, 0xfbff]:
        assert lib.orc_f16_to_f32(h) == float(np.uint16(h).view(np.float16))


def _coord_hdr(w, h):
    hdr = np.zeros((h, w, 4), dtype=np.float32)
    hdr[..., 0] = np.arange(w, dtype=np.float32)[None, :]
    hdr[..., 1] = np.arange(h, dtype=np.float32)[:, None]
    hdr[..., 3] = 1.0
    return hdr.reshape(-1)


FACES = [((1, 0, 0), (0, 1, 0), (0, 0, -1)), ((-1, 0, 0), (0, 1, 0), (0, 0, 1)), ((0, -1, 0), (0, 0, 1), (1, 0, 0)),
         ((0, 1, 0), (0, 0, -1), (1, 0, 0)), ((0, 0, 1), (0, 1, 0), (1, 0, 0)), ((0, 0, -1), (0, 1, 0), (-1, 0, 0))]   # equirectangular.wgsl:20-57


def test_equirect_face_table_and_pixel_choice():
    """equirectangular.wgsl:60-69: corner texel gid=(0,0) of each face looks along
    forward - right - up, and the source pixel is the truncated (atan2, asin) * invAtan + 0.5."""
    w, h, size = 64, 32, 4
    cube = ob.equirect_to_cube(_coord_hdr(w, h), w, h, size)
    for face, (fwd, up, right) in enumerate(FACES):
        for gy in range(size):
            for gx in range(size):
                cu = np.float32(gx) / np.float32(size) * np.float32(2) - np.float32(1)
                cv = np.float32(gy) / np.float32(size) * np.float32(2) - np.float32(1)
                d = np.array(fwd, np.float32) + np.array(right, np.float32) * cu + np.array(up, np.float32) * cv
                d = d / np.float32(math.sqrt(float(np.dot(d.astype(np.float64), d.astype(np.float64)))))
                u = np.float32(math.atan2(d[2], d[0])) * np.float32(0.1591) + np.float32(0.5)
                v = np.float32(math.asin(min(1.0, max(-1.0, float(d[1]))))) * np.float32(0.3183) + np.float32(0.5)
                # allow the neighbouring texel when the coordinate sits within rounding of a boundary
                px, py = float(u) * w, float(v) * h
                got = cube[face, gy, gx]
                assert abs(got[0] - math.floor(px)) <= (1 if abs(px - round(px)) < 1e-3 else 0), (face, gx, gy)
                assert abs(got[1] - math.floor(py)) <= (1 if abs(py - round(py)) < 1e-3 else 0), (face, gx, gy)
    # gid = (0, 0): direction forward - right - up, e.g. +X face looks along (1, -1, 1) / sqrt(3)
    d = np.array([1.0, -1.0, 1.0]) / math.sqrt(3)
    u = math.atan2(d[2], d[0]) * 0.1591 + 0.5
    v = math.asin(d[1]) * 0.3183 + 0.5
    assert tuple(cube[0, 0, 0, :2]) == (math.floor(u * w), math.floor(v * h))


def test_prefilter_roughness_zero_is_nearest_texel():
    # prefiltermap.wgsl:131-168 with roughness 0: H = N = L, weight 1, mip 0 -> the equirect texel itself
    pkg_hdr = np.random.default_rng(3).uniform(0.1, 4.0, size=(64 * 128 * 4)).astype(np.float32)
    w, h, size = 128, 64, 8
    chain, levels = ob.mip_chain_rgba32f(pkg_hdr, w, h)
    pre = ob.prefilter_level(chain, w, h, levels, size, 0.0)
    sky = ob.equirect_to_cube(pkg_hdr, w, h, size)
    np.testing.assert_allclose(pre[..., :3], sky[..., :3], rtol=3e-5)
    assert (pre[..., 3] == 1.0).all()


def test_prefilter_constant_environment_is_constant():
    w, h, size = 2048, 1024, 4
    hdr = np.tile(np.array([0.5, 1.5, 2.5, 1.0], np.float32), w * h)
    chain, levels = ob.mip_chain_rgba32f(hdr, w, h)
    assert levels == 12
    for rough in (0.25, 1.0):
        pre = ob.prefilter_level(chain, w, h, levels, size, rough)
        np.testing.assert_allclose(pre[..., :3], np.broadcast_to([0.5, 1.5, 2.5], pre[..., :3].shape), rtol=2e-5)


def test_prefilter_matches_a_float64_evaluation():
    """prefiltermap.wgsl:70-171 written again in float64 numpy (Hammersley points, GGX importance sampling with its
    tangent frame and the |N.z| >= 0.999 branch, pdf-driven source level with the 512^2 constant, two-level blend,
    N.L weighting) on a smooth environment, for texels of every face and three roughness values."""
    w, h, size = 256, 128, 4
    u, v = (np.arange(w) + 0.5) / w, (np.arange(h) + 0.5) / h
    phi, lat = (u - 0.5) / 0.1591, (v - 0.5) / 0.3183
    d = np.stack([np.cos(lat)[:, None] * np.cos(phi)[None, :], np.broadcast_to(np.sin(lat)[:, None], (h, w)), np.cos(lat)[:, None] * np.sin(phi)[None, :]], -1)
    hdr = np.ones((h, w, 4), np.float32)
    hdr[..., 0] = 1.5 + d @ np.array([0.5, 0.7, -0.4]); hdr[..., 1] = 2.0 + np.sin(3.0 * d[..., 0]) * d[..., 1]; hdr[..., 2] = 0.2 + d[..., 2] ** 2
    chain, levels = ob.mip_chain_rgba32f(hdr.reshape(-1), w, h)
    level = [ob.chain_level(chain, w, h, l).astype(np.float64) for l in range(levels)]
    PI = 3.14159265359
    faces = [((1, 0, 0), (0, 1, 0), (0, 0, -1)), ((-1, 0, 0), (0, 1, 0), (0, 0, 1)), ((0, -1, 0), (0, 0, 1), (1, 0, 0)),
             ((0, 1, 0), (0, 0, -1), (1, 0, 0)), ((0, 0, 1), (0, 1, 0), (1, 0, 0)), ((0, 0, -1), (0, 1, 0), (-1, 0, 0))]   # forward, up, right
    i = np.arange(1024, dtype=np.uint64)
    bits = np.zeros(1024, np.uint64)
    for b in range(32):
        bits |= ((i >> np.uint64(b)) & np.uint64(1)) << np.uint64(31 - b)              # bit reversal
    xi = np.stack([i / 1024.0, bits * 2.3283064365386963e-10], 1)

    def texel_of(L, l):
        l = min(max(l, 0), levels - 1)                                                  # a level outside the chain is clamped into it
        lw, lh = max(1, w >> l), max(1, h >> l)
        eu, ev = math.atan2(L[2], L[0]) * 0.1591 + 0.5, math.asin(min(max(L[1], -1.0), 1.0)) * 0.3183 + 0.5
        return level[l][min(int(ev * lh), lh - 1), min(int(eu * lw), lw - 1), :3]

    def prefilter(face, gx, gy, rough):
        fwd, up_, right = (np.array(a, float) for a in faces[face])
        N = fwd + right * (gx / size * 2 - 1) + up_ * (gy / size * 2 - 1)
        N /= np.linalg.norm(N)
        a = rough * rough
        up = np.array([0.0, 0.0, 1.0]) if abs(N[2]) < 0.999 else np.array([1.0, 0.0, 0.0])
        tangent = np.cross(up, N); tangent /= np.linalg.norm(tangent)
        bitangent = np.cross(N, tangent)
        color, total = np.zeros(3), 0.0
        for x0, x1 in xi:
            ph = 2.0 * PI * x0
            ct = math.sqrt((1.0 - x1) / (1.0 + (a * a - 1.0) * x1)); st = math.sqrt(max(1.0 - ct * ct, 0.0))
            H = tangent * (math.cos(ph) * st) + bitangent * (math.sin(ph) * st) + N * ct
            H /= np.linalg.norm(H)
            L = 2.0 * (N @ H) * H - N
            L /= np.linalg.norm(L)
            ndotl = max(float(N @ L), 0.0)
            if ndotl <= 0.0:
                continue
            ndoth = max(float(N @ H), 0.0)
            D = a * a / (PI * (ndoth * ndoth * (a * a - 1.0) + 1.0) ** 2)
            pdf = D * ndoth / (4.0 * ndoth) + 0.0001                                     # V = N: H.V = N.H
            sa_texel, sa_sample = 4.0 * PI / (6.0 * 512.0 * 512.0), 1.0 / (1024.0 * pdf + 0.0001)
            mip = 0.0 if rough == 0.0 else 0.5 * math.log2(sa_sample / sa_texel)
            m0 = math.floor(mip); f = mip - m0
            color += (texel_of(L, m0) * (1.0 - f) + texel_of(L, m0 + 1) * f) * ndotl
            total += ndotl
        return color / total

    for rough in (0.25, 0.6, 1.0):
        got = ob.prefilter_level(chain, w, h, levels, size, np.float32(rough))
        for face, gx, gy in ((0, 1, 3), (1, 0, 0), (2, 3, 1), (3, 2, 2), (4, 2, 2), (5, 1, 2)):
            want = prefilter(face, gx, gy, float(np.float32(rough)))
            np.testing.assert_allclose(got[face, gy, gx, :3], want, rtol=2e-3, err_msg=f"roughness {rough} face {face} texel ({gx},{gy})")


def test_irradiance_constant_environment():
    # irradiancemap.wgsl:75-88: PI * L * sum(cos t sin t) / N over the 252 x 63 grid
    w, h, size = 64, 32, 2
    hdr = np.tile(np.array([2.0, 1.0, 0.5, 1.0], np.float32), w * h)
    irr = ob.irradiance(hdr, w, h, size)
    s, theta = 0.0, np.float32(0)
    thetas = []
    while theta < np.float32(0.5) * np.float32(3.14159265359):
        thetas.append(float(theta)); theta = np.float32(theta + np.float32(0.025))
    s = sum(math.cos(t) * math.sin(t) for t in thetas) * 252
    want = 3.14159265359 * s / (252 * 63)
    np.testing.assert_allclose(irr[..., 0], 2.0 * want, rtol=1e-4)
    np.testing.assert_allclose(irr[..., 2], 0.5 * want, rtol=1e-4)


def test_irradiance_of_a_linear_environment():
    """A radiance field L(d) = 1 + a.d convolves to 1 + (2/3) a.N under the shader's normalisation (PI * sum / count
    with weights cos t sin t): checks the face table and cube coordinates that give N (irradiancemap.wgsl:21-67), the
    hemisphere centred on N, and the direction -> equirect pixel mapping (:82-83), for every face."""
    w, h, size = 1024, 512, 4
    a = np.array([0.3, -0.5, 0.4])
    u, v = (np.arange(w) + 0.5) / w, (np.arange(h) + 0.5) / h
    phi, lat = (u - 0.5) / 0.1591, (v - 0.5) / 0.3183                        # inverse of eqUV = (atan2(z, x), asin(y)) * invAtan + 0.5
    d = np.stack([np.cos(lat)[:, None] * np.cos(phi)[None, :], np.broadcast_to(np.sin(lat)[:, None], (h, w)), np.cos(lat)[:, None] * np.sin(phi)[None, :]], -1)
    hdr = np.ones((h, w, 4), np.float32)
    hdr[..., 0] = 1.0 + d @ a
    hdr[..., 1] = 1.0 - 0.5 * (d @ a)
    irr = ob.irradiance(hdr.reshape(-1), w, h, size)
    faces = [((1, 0, 0), (0, 1, 0), (0, 0, -1)), ((-1, 0, 0), (0, 1, 0), (0, 0, 1)), ((0, -1, 0), (0, 0, 1), (1, 0, 0)),
             ((0, 1, 0), (0, 0, -1), (1, 0, 0)), ((0, 0, 1), (0, 1, 0), (1, 0, 0)), ((0, 0, -1), (0, 1, 0), (-1, 0, 0))]   # forward, up, right
    for face, (fwd, up, right) in enumerate(faces):
        for gy in range(size):
            for gx in range(size):
                cx, cy = gx / size * 2 - 1, gy / size * 2 - 1             # no half-texel offset in the shader
                n = np.array(fwd, float) + np.array(right, float) * cx + np.array(up, float) * cy
                n /= np.linalg.norm(n)
                if abs(n[1]) == 1.0:
                    # N parallel to the shader's fixed up vector: cross(up, N) = 0, the tangent frame and every sample direction
                    # are NaN, and all taps read equirect pixel (0, 0) (the one implementation-defined texel of the map)
                    assert abs(irr[face, gy, gx, 0] - hdr[0, 0, 0]) < 0.02 and face in (2, 3) and (gx, gy) == (size // 2, size // 2)
                    continue
                assert abs(irr[face, gy, gx, 0] - (1 + 2 / 3 * (a @ n))) < 0.02, (face, gx, gy)
                assert abs(irr[face, gy, gx, 1] - (1 - 1 / 3 * (a @ n))) < 0.02, (face, gx, gy)
                assert abs(irr[face, gy, gx, 2] - 1.0) < 0.01


def test_brdf_lut_quirks():
    """brdf.wgsl:67-106: column 0 (NdotV = 0) is 0/0 = NaN; row y holds roughness (size-y)/(size-1);
    the last row/column entry is finite with A + B <= 1."""
    size = 64
    f32, f16 = ob.brdf_lut(size)
    assert np.isnan(f32[:, 0, 0]).all() and np.isnan(f32[:, 0, 1]).all()
    assert np.isfinite(f32[:, 1:, :2]).all()
    a, b = f32[size - 1, size - 1, 0], f32[size - 1, size - 1, 1]
    assert 0.9 < a + b <= 1.0 + 1e-6
    assert (f32[..., 2] == 0).all() and (f32[..., 3] == 1).all()
    # independent evaluation of one entry in f64 following the WGSL
    x, y = 40, 20
    ndv, rough = x / (size - 1), (size - y) / (size - 1)
    V = np.array([math.sqrt(1 - ndv * ndv), 0.0, ndv])
    A = B = 0.0
    for i in range(1024):
        xi_y = int(format(i, "032b")[::-1], 2) / 2 ** 32
        phi = 2 * 3.14159265359 * i / 1024
        a_ = rough * rough
        ct = math.sqrt((1 - xi_y) / (1 + (a_ * a_ - 1) * xi_y)); st = math.sqrt(1 - ct * ct)
        H = np.array([math.sin(phi) * st, -math.cos(phi) * st, ct])   # tangent (0,-1,0), bitangent (1,0,0) for N = +Z
        L = 2 * V.dot(H) * H - V; L /= np.linalg.norm(L)
        if L[2] > 0:
            k = rough * rough / 2
            G = (ndv / (ndv * (1 - k) + k)) * (L[2] / (L[2] * (1 - k) + k))
            gv = G * max(V.dot(H), 0) / (max(H[2], 0) * ndv)
            fc = (1 - max(V.dot(H), 0)) ** 5
            A += (1 - fc) * gv; B += fc * gv
    np.testing.assert_allclose(f32[y, x, :2], [A / 1024, B / 1024], rtol=2e-4)
    assert (f16.view(np.float16).astype(np.float32)[y, x, :2] == f32[y, x, :2].astype(np.float16).astype(np.float32)).all()


def test_mips_are_box_filters():
    # mipmap.wgsl:23-26: one bilinear tap at the destination texel centre = 2x2 mean
    rng = np.random.default_rng(5)
    w, h = 8, 4
    data = rng.uniform(0, 10, size=(h, w, 4)).astype(np.float32)
    chain, levels = ob.mip_chain_rgba32f(data, w, h)
    assert levels == 4      # utils.ts:20-24: 1 + floor(log2(8))
    l1 = ob.chain_level(chain, w, h, 1)
    want = data.reshape(h // 2, 2, w // 2, 2, 4).mean(axis=(1, 3))
    np.testing.assert_allclose(l1, want, rtol=1e-6)
    assert ob.chain_level(chain, w, h, 3).shape == (1, 1, 4)   # 8x4 -> 4x2 -> 2x1 -> 1x1
    img = rng.integers(0, 256, size=(4, 4, 4), dtype=np.uint8); img[..., 3] = 255
    c8, lv = ob.mip_chain_rgba8(img, False)
    m = ob.chain_level(c8, 4, 4, 1)
    want8 = np.rint(img.astype(np.float32).reshape(2, 2, 2, 2, 4).mean(axis=(1, 3)))
    assert np.abs(m.astype(np.int32) - want8.astype(np.int32)).max() <= 1
    cs, _ = ob.mip_chain_rgba8(img, True)   # sRGB: averaged in linear light, darker than the byte mean never brighter than max
    ms = ob.chain_level(cs, 4, 4, 1)
    assert (ms[..., :3] >= m[..., :3] - 1).all()   # linear-light average of an sRGB image is >= the encoded-value average


def _helmet():
    with open(os.path.join(HELMET, "DamagedHelmet.gltf")) as f:
        g = json.load(f)
    raw = open(os.path.join(HELMET, "DamagedHelmet.bin"), "rb").read()
    return g, raw


def test_helmet_asset_statistics():
    # DamagedHelmet.gltf:6,18: 46356 indices = 15452 triangles, 14556 vertices; 10 triangles have a zero UV determinant
    g, raw = _helmet()
    idx = np.frombuffer(raw, np.uint16, 46356, 0).reshape(-1, 3)
    pos = np.frombuffer(raw, np.float32, 14556 * 3, 92712).reshape(-1, 3)
    uv = np.frombuffer(raw, np.float32, 14556 * 2, 442056).reshape(-1, 2)
    assert idx.shape[0] == 15452 and idx.max() == 14555
    np.testing.assert_allclose(pos.max(0), g["accessors"][1]["max"], rtol=1e-6)
    np.testing.assert_allclose(pos.min(0), g["accessors"][1]["min"], rtol=1e-6)
    assert uv[:, 1].min() > 1.0 and uv[:, 1].max() < 2.0      # only "repeat" addressing renders this mesh
    e1, e2 = uv[idx[:, 1]] - uv[idx[:, 0]], uv[idx[:, 2]] - uv[idx[:, 0]]
    det = e1[:, 0] * e2[:, 1] - e1[:, 1] * e2[:, 0]
    assert int((det == 0).sum()) == 10
    assert int((det < 0).sum()) == 7659


def test_matrix_conventions():
    lib = ob.load()
    m = np.zeros(16, np.float32)
    lib.orc_mat4_perspective(math.radians(45.0), 16 / 9, 1.0, 100.0, ob.ptr(m))
    f = 1 / math.tan(math.radians(22.5))
    np.testing.assert_allclose(m[0], f / (16 / 9), rtol=1e-6)
    np.testing.assert_allclose(m[5], 2.41421356, rtol=1e-6)
    assert m[11] == -1 and m[15] == 0
    np.testing.assert_allclose([m[10], m[14]], [100 / (1 - 100), 100 / (1 - 100)], rtol=1e-6)   # z in [0,1]: near -> 0, far -> 1
    eye, tgt, up = np.array([0, 0, 5], np.float32), np.zeros(3, np.float32), np.array([0, 1, 0], np.float32)
    lib.orc_mat4_look_at(ob.ptr(eye), ob.ptr(tgt), ob.ptr(up), ob.ptr(m))
    want = np.eye(4, dtype=np.float32); want[3, 2] = -5     # column-major: translation in elements 12..14
    np.testing.assert_array_equal(m.reshape(4, 4), want)
    q = np.array([0.7071068286895752, 0, 0, 0.7071068286895752], np.float32)   # DamagedHelmet.gltf:162-167
    lib.orc_node_matrix(None, None, ob.ptr(q), None, ob.ptr(m))
    rx = np.array([[1, 0, 0, 0], [0, 0, 1, 0], [0, -1, 0, 0], [0, 0, 0, 1]], np.float32)   # +90 degrees about X, columns
    np.testing.assert_allclose(m.reshape(4, 4), rx, atol=1e-6)
    inv = np.zeros(16, np.float32)
    a = np.random.default_rng(2).normal(size=16).astype(np.float32)
    assert lib.orc_mat4_inverse(ob.ptr(a), ob.ptr(inv)) == 0
    prod = np.zeros(16, np.float32)
    lib.orc_mat4_mul(ob.ptr(a), ob.ptr(inv), ob.ptr(prod))
    np.testing.assert_allclose(prod.reshape(4, 4), np.eye(4), atol=1e-4)


# ---- rasteriser known answers on hand-made meshes ----
def _tiny_scene(verts, indices):
    sc = ob.Scene()
    sc.verts = ob.ptr(verts); sc.n_verts = verts.size // 8
    sc.indices = ob.ptr(indices); sc.n_indices = indices.size
    sc.n_instances = 1
    return sc


def _identity_uniforms():
    u = ob.FrameUniforms()
    eye = [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1]
    for name in ("model", "normal", "view", "projection", "inv_view_proj"):
        getattr(u, name)[:] = eye
    return u


def _raster(verts_xyz, tris, W, H, msaa=False):
    lib = ob.load()
    v = np.zeros((len(verts_xyz), 8), np.float32)
    v[:, :3] = verts_xyz
    v[:, 7] = 1
    idx = np.asarray(tris, np.uint16).reshape(-1)
    sc = _tiny_scene(v, idx)
    u = _identity_uniforms()
    u.sampler_index = 16 if msaa else 0
    shape = (H, W, 4) if msaa else (H, W)
    vis = np.zeros(shape, np.uint32); depth = np.zeros(shape, np.float32)
    lib.orc_raster(C.byref(sc), C.byref(u), W, H, 0, H, ob.ptr(vis), ob.ptr(depth))
    return vis, depth


def test_raster_fill_rule_partitions_a_quad():
    """Two triangles sharing a diagonal cover every pixel of the quad exactly once (top-left rule),
    and the viewport maps NDC y = +1 to row 0."""
    W, H = 16, 8
    quad = [(-1, -1, 0.5), (1, -1, 0.5), (1, 1, 0.5), (-1, 1, 0.5)]
    vis, depth = _raster(quad, [(0, 1, 2), (0, 2, 3)], W, H)   # counter-clockwise in NDC = front-facing
    assert (vis > 0).all()
    assert set(np.unique(vis)) == {1, 2}
    assert vis[H - 1, W - 1] == 1 and vis[0, 0] == 2          # triangle (0,1,2) holds the bottom-right corner
    assert (depth == 0.5).all()
    # a half-pixel-aligned inner quad: pixel centres exactly on the left/top edges are inside, on the right/bottom edges outside
    x0, x1, y0, y1 = 4.5, 10.5, 2.5, 5.5
    def ndc(px, py):
        return (px / W * 2 - 1, 1 - py / H * 2, 0.25)
    q2 = [ndc(x0, y1), ndc(x1, y1), ndc(x1, y0), ndc(x0, y0)]
    vis, _ = _raster(q2, [(0, 1, 2), (0, 2, 3)], W, H)
    cov = vis > 0
    assert cov[2:5, 4:10].all() and cov.sum() == 3 * 6


def test_multisampled_raster_covers_the_standard_sample_points():
    """Four samples per pixel at (0.375, 0.125), (0.875, 0.375), (0.125, 0.625), (0.625, 0.875):
    a rectangle whose edges cut pixels in half keeps exactly the samples on its side, a full-screen
    quad is partitioned sample by sample, and depth is evaluated at the sample, not the centre."""
    W, H = 16, 8
    def ndc(px, py, z=0.25):
        return (px / W * 2 - 1, 1 - py / H * 2, z)
    x0, x1, y0, y1 = 4.5, 10.5, 2.5, 5.5
    q = [ndc(x0, y1), ndc(x1, y1), ndc(x1, y0), ndc(x0, y0)]
    vis, _ = _raster(q, [(0, 1, 2), (0, 2, 3)], W, H, msaa=True)
    cov = vis > 0
    sx = np.array([0.375, 0.875, 0.125, 0.625]); sy = np.array([0.125, 0.375, 0.625, 0.875])
    ys, xs = np.mgrid[0:H, 0:W]
    px = xs[..., None] + sx; py = ys[..., None] + sy
    expect = (px >= x0) & (px < x1) & (py >= y0) & (py < y1)
    assert (cov == expect).all()
    assert cov[3, 4].tolist() == [False, True, False, True]       # left edge at x = 4.5 splits pixel 4
    assert cov[2, 6].tolist() == [False, False, True, True]       # top edge at y = 2.5 splits row 2
    assert cov[3, 6].all()
    full = [(-1, -1, 0.5), (1, -1, 0.5), (1, 1, 0.5), (-1, 1, 0.5)]
    vis, depth = _raster(full, [(0, 1, 2), (0, 2, 3)], W, H, msaa=True)
    assert (vis > 0).all() and set(np.unique(vis)) == {1, 2} and (depth == 0.5).all()
    # the diagonal y = 8 - x/2 crosses pixel (0, 7) below all samples but the last one
    assert vis[7, 0].tolist() == [2, 2, 2, 1] and vis[0, 0].tolist() == [2, 2, 2, 2]
    # depth slopes with x only: z = 0.25 + 0.5 * x/W, read back at each sample's own x
    slope = [(-1, -1, 0.25), (1, -1, 0.75), (1, 1, 0.75), (-1, 1, 0.25)]
    _, depth = _raster(slope, [(0, 1, 2), (0, 2, 3)], W, H, msaa=True)
    np.testing.assert_allclose(depth, np.broadcast_to(0.25 + 0.5 * px / W, depth.shape), atol=1e-6)
    assert (np.diff(depth[0, 0][[2, 0, 3, 1]]) > 0).all()


def test_raster_culls_back_faces_and_resolves_depth():
    W, H = 8, 8
    quad = [(-1, -1, 0.5), (1, -1, 0.5), (1, 1, 0.5), (-1, 1, 0.5)]
    vis, _ = _raster(quad, [(0, 2, 1), (0, 3, 2)], W, H)      # clockwise: culled (cullMode "back", renderer.ts:471)
    assert (vis == 0).all()
    near = [(x, y, 0.2) for x, y, _ in quad]
    vis, depth = _raster(quad + near, [(0, 1, 2), (0, 2, 3), (4, 5, 6), (4, 6, 7)], W, H)
    assert set(np.unique(vis)) == {3, 4} and (depth == np.float32(0.2)).all()      # depthCompare "less"
    vis, _ = _raster(quad + quad, [(0, 1, 2), (0, 2, 3), (4, 5, 6), (4, 6, 7)], W, H)
    assert set(np.unique(vis)) == {1, 2}                       # equal depth: the earlier draw stays
    behind = [(x, y, 1.5) for x, y, _ in quad]                 # beyond the far plane: clipped away
    vis, _ = _raster(behind, [(0, 1, 2), (0, 2, 3)], W, H)
    assert (vis == 0).all()


def test_raster_clips_against_the_near_plane():
    """A triangle with one vertex behind the eye (w < 0) is clipped, not dropped: the part in front
    of the near plane still covers pixels."""
    lib = ob.load()
    W, H = 32, 32
    proj = np.zeros(16, np.float32)
    lib.orc_mat4_perspective(math.radians(90.0), 1.0, 1.0, 100.0, ob.ptr(proj))
    v = np.zeros((3, 8), np.float32)
    # The third vertex is behind the camera at the origin; the edges cross the near plane at
    # NDC (-1.2, 0.6) and (1.2, 0.6), above the two visible vertices, so (0, 1, 2) stays counter-clockwise.
    v[:, :3] = [(-2, -1, -3), (2, -1, -3), (0, 3.0, 2.0)]
    idx = np.array([0, 1, 2], np.uint16)
    sc = _tiny_scene(v, idx)
    u = _identity_uniforms()
    u.projection[:] = [float(x) for x in proj]
    vis = np.zeros((H, W), np.uint32); depth = np.zeros((H, W), np.float32)
    lib.orc_raster(C.byref(sc), C.byref(u), W, H, 0, H, ob.ptr(vis), ob.ptr(depth))
    assert (vis > 0).sum() > 50
    assert depth[vis > 0].min() >= 0.0 and depth[vis > 0].max() < 1.0


# ---- fs_main against an independent float64 evaluation ----
@pytest.mark.parametrize("turn_deg,shift", [(0.0, (0.0, 0.0, 0.0)), (50.0, (0.3, -0.2, 0.5)), (-35.0, (-0.4, 0.1, -1.0))])
def test_fragment_shading_matches_a_float64_evaluation(turn_deg, shift):
    """shader.wgsl:37-47, :93-160 evaluated in float64 numpy, written from the WGSL alone, for a scene where every lookup
    is a constant (1x1 material textures, one-texel cube maps, a flat BRDF table): the oracle's f32 frame must agree
    to the last 8-bit step on (almost) every pixel.  A quad in a plane: the varyings of a pixel follow from where the
    ray through the pixel centre meets the plane, which pins perspective-correct interpolation when the plane is
    turned away from the camera, the model / normal matrices, the coarse derivatives and the TBN construction,
    the four-light Cook-Torrance loop, the split-sum ambient term, the sRGB decode and the gamma curve."""
    lib = ob.load()
    W, H = 64, 36
    fov, near, far = math.radians(45.0), 1.0, 100.0
    eye = np.array([0.0, 0.0, 5.0])
    ca, sa = math.cos(math.radians(turn_deg)), math.sin(math.radians(turn_deg))
    R = np.array([[ca, 0, sa], [0, 1, 0], [-sa, 0, ca]])                                 # about +y
    P0 = np.array(shift, float)
    model = np.eye(4); model[:3, :3] = R; model[:3, 3] = P0
    quad = np.float32([[-1.5, -1.5, 0], [1.5, -1.5, 0], [1.5, 1.5, 0], [-1.5, 1.5, 0]])
    v = np.zeros((4, 8), np.float32)
    v[:, :3] = quad
    v[:, 3] = (quad[:, 0] + 1.5) / 3.0; v[:, 4] = (1.5 - quad[:, 1]) / 3.0
    v[:, 5:8] = [0, 0, 1]
    idx = np.array([0, 1, 2, 0, 2, 3], np.uint16)
    texel = {"albedo": (200, 120, 60, 255), "mr": (0, 140, 180, 255), "normal": (150, 100, 240, 255), "emissive": (30, 60, 90, 255), "occlusion": (210, 0, 0, 255)}
    chains = [np.array(texel[k], np.uint8) for k in ("albedo", "mr", "normal", "emissive", "occlusion")]
    irr_c, pre_c, lut_c = np.float32([0.25, 0.5, 0.75, 1.0]), np.float32([0.9, 0.6, 0.3, 1.0]), np.float32([0.7, 0.1, 0.0, 1.0])
    irr = np.tile(irr_c, (6, 1)).astype(np.float32); pre = np.tile(pre_c, (6, 1)).astype(np.float32); sky = np.tile(np.float32([0, 0, 0, 1]), (6, 1))
    lut = np.tile(lut_c.astype(np.float16).view(np.uint16), (4, 1))                     # 2 x 2 table
    sc = _tiny_scene(v, idx)
    for i, ch in enumerate(chains):
        sc.tex[i] = ob.ptr(ch).value; sc.tex_w[i] = 1; sc.tex_h[i] = 1; sc.tex_levels[i] = 1
    sc.irradiance = ob.ptr(irr); sc.irr_size = 1
    sc.prefilter = ob.ptr(pre); sc.pre_size = 1; sc.pre_levels = 1
    sc.lut_f16 = ob.ptr(lut); sc.lut_size = 2
    sc.skybox = ob.ptr(sky); sc.sky_size = 1
    u = _identity_uniforms()
    proj = np.zeros(16, np.float32)
    lib.orc_mat4_perspective(fov, W / H, near, far, ob.ptr(proj))
    u.projection[:] = [float(x) for x in proj]
    u.view[:] = [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, -5, 1]                       # lookAt from (0, 0, 5)
    u.model[:] = [float(x) for x in model.T.reshape(-1)]                                 # column-major
    u.normal[:] = [float(x) for x in np.linalg.inv(model).reshape(-1)]                   # transpose(inverse(model)), renderer.ts:1837, column-major
    u.view_pos[:] = [0, 0, 5, 0]
    lights = [(10.0, 10.0, 10.0), (-10.0, 10.0, 10.0), (10.0, -10.0, 10.0), (-10.0, -10.0, 10.0)]
    for i, lp in enumerate(lights):
        u.light_pos[i][:] = [lp[0], lp[1], lp[2], 0.0]; u.light_color[i][:] = [300.0, 300.0, 300.0, 0.0]
    emissive_factor = np.array([1.0, 0.5, 0.25])
    u.emissive_factor[:] = [1.0, 0.5, 0.25, 0.0]
    u.sampler_index = 15; u.use_mipmaps = 1
    rgba = np.zeros((H, W, 4), np.uint8); vis = np.zeros((H, W), np.uint32)
    lib.orc_draw_frame(C.byref(sc), C.byref(u), W, H, 0, H, ob.ptr(rgba), ob.ptr(vis))
    assert (vis > 0).mean() > 0.12

    def srgb(b):
        c = np.asarray(b, np.float64) / 255.0
        return np.where(c <= 0.04045, c / 12.92, ((c + 0.055) / 1.055) ** 2.4)
    albedo = srgb(texel["albedo"][:3]); emissive = srgb(texel["emissive"][:3]) * emissive_factor
    tn = np.asarray(texel["normal"][:3], np.float64) / 255.0 * 2.0 - 1.0
    roughness, metalness = texel["mr"][1] / 255.0, texel["mr"][2] / 255.0
    occlusion = texel["occlusion"][0] / 255.0
    lut_a, lut_b = float(np.float16(lut_c[0])), float(np.float16(lut_c[1]))
    th = math.tan(fov / 2)
    norm = lambda a: a / np.linalg.norm(a)
    PI = 3.14159265359
    N = R @ np.array([0.0, 0.0, 1.0])

    def world(px, py):   # the ray through the pixel centre meets the quad's plane
        xn, yn = (px + 0.5) / W * 2 - 1, 1 - (py + 0.5) / H * 2
        ray = np.array([xn * th * W / H, yn * th, -1.0])
        return eye + ray * (((P0 - eye) @ N) / (ray @ N))

    def uv_of(p):
        local = R.T @ (p - P0)
        return np.array([(local[0] + 1.5) / 3.0, (1.5 - local[1]) / 3.0])

    def ggx(c, rough):
        k = (rough + 1.0) ** 2 / 8.0
        return c / (c * (1.0 - k) + k)

    worst, off = 0, 0
    for y in range(H):
        for x in range(W):
            if not vis[y, x]:
                continue
            x0, y0 = x & ~1, y & ~1
            p00, p10, p01 = world(x0, y0), world(x0 + 1, y0), world(x0, y0 + 1)
            q1, q2 = p10 - p00, p01 - p00
            st1, st2 = uv_of(p10) - uv_of(p00), uv_of(p01) - uv_of(p00)
            pos = world(x, y)
            T = norm(q1 * st2[1] - q2 * st1[1])
            B = -norm(np.cross(N, T))
            normal = norm(T * tn[0] + B * tn[1] + N * tn[2])
            view_dir = norm(eye - pos)
            f0 = 0.04 * (1 - metalness) + albedo * metalness
            Lo = np.zeros(3)
            for lp in lights:
                to_light = np.asarray(lp) - pos
                light_dir = norm(to_light); halfway = norm(view_dir + light_dir)
                radiance = 300.0 / float(to_light @ to_light)
                a2 = (roughness * roughness) ** 2
                ndoth = max(float(normal @ halfway), 0.0)
                D = a2 / (PI * (ndoth * ndoth * (a2 - 1.0) + 1.0) ** 2)
                F = f0 + (1.0 - f0) * min(max(1.0 - max(float(halfway @ view_dir), 0.0), 0.0), 1.0) ** 5
                ndotv, ndotl = max(float(normal @ view_dir), 0.0), max(float(normal @ light_dir), 0.0)
                G = ggx(ndotv, roughness) * ggx(ndotl, roughness)
                kD = (1.0 - F) * (1.0 - metalness)
                spec = D * F * G / (4.0 * ndotv * ndotl + 0.001)
                Lo += (kD * albedo / PI + spec) * radiance * ndotl
            ndotv = max(float(normal @ view_dir), 0.0)
            kS = f0 + (np.maximum(1.0 - roughness, f0) - f0) * min(max(1.0 - ndotv, 0.0), 1.0) ** 5
            kD = (1.0 - kS) * (1.0 - metalness)
            ambient = (kD * irr_c[:3].astype(np.float64) * albedo + pre_c[:3].astype(np.float64) * (kS * lut_a + lut_b)) * occlusion
            color = (ambient + Lo + emissive) ** (1.0 / 2.2)
            want = np.rint(np.clip(color, 0.0, 1.0) * 255.0).astype(int)
            d = int(np.abs(want - rgba[y, x, :3].astype(int)).max())
            worst = max(worst, d); off += d > 0
    assert worst <= 1 and off <= 0.03 * (vis > 0).sum(), (worst, off)


# ---- samplers ----
def _sample2d(chain, w, h, levels, srgb, sampler, u, v, ddx=(0.0, 0.0), ddy=(0.0, 0.0)):
    out = np.zeros(4, np.float32)
    ob.load().orc_sample_2d_rgba8(ob.ptr(chain), w, h, levels, 1 if srgb else 0, sampler, u, v, ob.ptr(np.float32(ddx)), ob.ptr(np.float32(ddy)), ob.ptr(out))
    return out.astype(np.float64)


def test_material_sampler_filters_wraps_and_picks_levels():
    """The sampler of renderer.ts:822-859 on a small texture: texel addressing, bilinear weights, repeat against
    clamp-to-edge, magFilter / minFilter chosen by the footprint, mipmapFilter "nearest" switching level at
    lambda = 0.5, and the sRGB decode of the *-srgb formats."""
    rng = np.random.default_rng(7)
    w, h = 8, 4
    img = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    chain, levels = ob.mip_chain_rgba8(img, False)
    assert levels == 4
    f = img.astype(np.float64) / 255.0
    REP, MAG, MIN = 3, 4, 8
    centre = lambda i, j: ((i + 0.5) / w, (j + 0.5) / h)
    for i, j in ((0, 0), (5, 2), (7, 3)):
        for sampler in (0, REP, MAG | MIN, REP | MAG | MIN):
            np.testing.assert_allclose(_sample2d(chain, w, h, 1, False, sampler, *centre(i, j)), f[j, i], atol=1e-6)
    # halfway between two texel centres: the mean with linear filtering, the right-hand texel with nearest
    u, v = 3.0 / w, 1.5 / h
    np.testing.assert_allclose(_sample2d(chain, w, h, 1, False, MAG | MIN, u, v), 0.5 * (f[1, 2] + f[1, 3]), atol=1e-6)
    np.testing.assert_allclose(_sample2d(chain, w, h, 1, False, 0, u, v), f[1, 3], atol=1e-6)
    # a general point: bilinear weights
    u, v = (2.5 + 0.3) / w, (1.5 + 0.8) / h
    want = (f[1, 2] * 0.7 + f[1, 3] * 0.3) * 0.2 + (f[2, 2] * 0.7 + f[2, 3] * 0.3) * 0.8
    np.testing.assert_allclose(_sample2d(chain, w, h, 1, False, MAG | MIN, u, v), want, atol=1e-6)
    # the left border: repeat blends the last and the first column, clamp-to-edge stays on the first
    np.testing.assert_allclose(_sample2d(chain, w, h, 1, False, REP | MAG | MIN, 0.0, 0.5 / h), 0.5 * (f[0, 7] + f[0, 0]), atol=1e-6)
    np.testing.assert_allclose(_sample2d(chain, w, h, 1, False, MAG | MIN, 0.0, 0.5 / h), f[0, 0], atol=1e-6)
    np.testing.assert_allclose(_sample2d(chain, w, h, 1, False, REP, 1.0 + 0.5 / w, 2.0 + 1.5 / h), f[1, 0], atol=1e-6)
    np.testing.assert_allclose(_sample2d(chain, w, h, 1, False, 0, 1.7, -0.3), f[0, 7], atol=1e-6)
    # footprint: rho texels per pixel along x.  rho < 1 magnifies (magFilter), rho > 1 minifies (minFilter)
    u, v = 3.0 / w, 1.5 / h
    lin, near_ = 0.5 * (f[1, 2] + f[1, 3]), f[1, 3]
    for sampler, small, large in ((MAG, lin, near_), (MIN, near_, lin)):
        np.testing.assert_allclose(_sample2d(chain, w, h, 1, False, sampler, u, v, (0.5 / w, 0), (0, 0.5 / h)), small, atol=1e-6)
        np.testing.assert_allclose(_sample2d(chain, w, h, 1, False, sampler, u, v, (1.2 / w, 0), (0, 0.5 / h)), large, atol=1e-6)
    # mip level: nearest level of lambda = log2(rho); 1.40 -> level 0, 1.43 -> level 1, 2.8 -> 1, 2.9 -> 2 (sqrt 2 and 2 sqrt 2)
    l1, l2 = ob.chain_level(chain, w, h, 1).astype(np.float64) / 255.0, ob.chain_level(chain, w, h, 2).astype(np.float64) / 255.0
    box = np.floor((img[0::2, 0::2].astype(int) + img[0::2, 1::2] + img[1::2, 0::2] + img[1::2, 1::2]) / 4.0 + 0.5) / 255.0
    np.testing.assert_allclose(l1, box, atol=1e-9)                      # rgba8 levels are stored rounded to nearest
    u, v = 2.5 / w, 1.5 / h                                              # centre of level-0 texel (2, 1), inside level-1 texel (1, 0), level-2 texel (0, 0)
    for rho, want in ((1.40, f[1, 2]), (1.43, l1[0, 1]), (2.8, l1[0, 1]), (2.9, l2[0, 0])):
        np.testing.assert_allclose(_sample2d(chain, w, h, levels, False, 0, u, v, (rho / w, 0), (0, 0.1 / h)), want, atol=1e-6, err_msg=str(rho))
    np.testing.assert_allclose(_sample2d(chain, w, h, levels, False, 0, u, v, (0, 0.1 / h), (0, 2.9 / h)), l2[0, 0], atol=1e-6)   # the larger axis counts
    np.testing.assert_allclose(_sample2d(chain, w, h, levels, False, 0, u, v, (1000.0, 0), (0, 0)), ob.chain_level(chain, w, h, 3)[0, 0] / 255.0, atol=1e-6)
    # sRGB formats decode before filtering (alpha stays linear)
    s = _sample2d(chain, w, h, 1, True, 0, *centre(5, 2))
    c = f[2, 5, :3]
    np.testing.assert_allclose(s[:3], np.where(c <= 0.04045, c / 12.92, ((c + 0.055) / 1.055) ** 2.4), atol=1e-6)
    assert abs(s[3] - f[2, 5, 3]) < 1e-6


def test_cube_sampler_faces_seams_and_levels():
    """Cube lookups (shader.wgsl:145, :149; skybox.wgsl:29): face selection and face coordinates of the Vulkan / WebGPU
    table, bilinear taps that cross an edge onto the neighbouring face, the three-texel mean at a corner, and
    the blend of two levels."""
    n = 4
    rng = np.random.default_rng(11)
    l0, l1 = rng.random((6, n, n, 4)).astype(np.float32), rng.random((6, n // 2, n // 2, 4)).astype(np.float32)
    chain = np.concatenate([l0.reshape(-1), l1.reshape(-1)])

    def sample(d, lod=0.0, levels=2):
        out = np.zeros(4, np.float32)
        ob.load().orc_sample_cube(ob.ptr(chain), n, levels, ob.ptr(np.float32(d)), lod, ob.ptr(out))
        return out.astype(np.float64)
    # major axis -> face; at the face centre the four middle texels blend equally
    for face, axis in enumerate([(1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)]):
        np.testing.assert_allclose(sample(axis), l0[face, 1:3, 1:3].reshape(4, 4).mean(0), atol=1e-6)
    # texel centres of each face: direction from (sc, tc) by the table's inverse
    inv = [lambda s, t: (1, -t, -s), lambda s, t: (-1, -t, s), lambda s, t: (s, 1, t), lambda s, t: (s, -1, -t), lambda s, t: (s, -t, 1), lambda s, t: (-s, -t, -1)]
    for face in range(6):
        for i, j in ((0, 0), (3, 1), (2, 3)):
            sc, tc = (i + 0.5) / n * 2 - 1, (j + 0.5) / n * 2 - 1
            d = np.array(inv[face](sc, tc), np.float64) * 3.7                    # any positive scale
            np.testing.assert_allclose(sample(d), l0[face, j, i], atol=1e-6)
    # on the edge between +X and +Z, at a texel row's centre: half of each face's edge texel
    for j in range(n):
        tc = (j + 0.5) / n * 2 - 1
        np.testing.assert_allclose(sample((1.0, -tc, 1.0)), 0.5 * (l0[0, j, 0] + l0[4, j, n - 1]), atol=1e-6)
    # the corner (1, 1, 1): the mean of the three texels that meet there
    np.testing.assert_allclose(sample((1.0, 1.0, 1.0)), (l0[0, 0, 0] + l0[2, n - 1, n - 1] + l0[4, 0, n - 1]) / 3.0, atol=1e-6)
    # between levels: linear blend; beyond the last level: the last level
    a, b = sample((0.3, -0.2, 1.0), 0.0), sample((0.3, -0.2, 1.0), 1.0)
    np.testing.assert_allclose(sample((0.3, -0.2, 1.0), 0.25), 0.75 * a + 0.25 * b, atol=1e-6)
    np.testing.assert_allclose(sample((0.3, -0.2, 1.0), 7.0), b, atol=1e-6)
    np.testing.assert_allclose(sample((0, 0, 1), 1.0), l1[4].reshape(4, 4).mean(0), atol=1e-6)


def test_skybox_pass_matches_a_float64_evaluation():
    """skybox.wgsl:13-33 and the draw of renderer.ts:2008-2011 on a frame without geometry: pixel centre -> clip
    position (z = w = 1) -> inverse view-projection -> direction with z negated -> bilinear cube lookup -> gamma,
    evaluated again in float64 for every pixel whose four taps lie inside one face."""
    lib = ob.load()
    W, H, n = 96, 54, 32
    rng = np.random.default_rng(3)
    sky = (rng.random((6, n, n, 4)) * 1.2).astype(np.float32)
    sky[..., 3] = 1.0
    proj, view, pv, inv = (np.zeros(16, np.float32) for _ in range(4))
    lib.orc_mat4_perspective(math.radians(45.0), W / H, 1.0, 100.0, ob.ptr(proj))
    eye, target, up = np.float32([3.0, 2.0, 4.0]), np.float32([0.2, -0.1, 0.0]), np.float32([0, 1, 0])
    lib.orc_mat4_look_at(ob.ptr(eye), ob.ptr(target), ob.ptr(up), ob.ptr(view))
    lib.orc_mat4_mul(ob.ptr(proj), ob.ptr(view), ob.ptr(pv))
    lib.orc_mat4_inverse(ob.ptr(pv), ob.ptr(inv))
    np.testing.assert_allclose((pv.astype(np.float64).reshape(4, 4).T @ inv.astype(np.float64).reshape(4, 4).T), np.eye(4), atol=1e-4)
    v = np.zeros((3, 8), np.float32)
    v[:, :3] = [(50, 50, 50), (51, 50, 50), (50, 51, 50)]                               # a triangle far outside the view: nothing is covered
    sc = _tiny_scene(v, np.array([0, 1, 2], np.uint16))
    sc.skybox = ob.ptr(sky); sc.sky_size = n
    u = _identity_uniforms()
    u.view[:] = [float(x) for x in view]; u.projection[:] = [float(x) for x in proj]; u.inv_view_proj[:] = [float(x) for x in inv]
    rgba = np.zeros((H, W, 4), np.uint8); vis = np.zeros((H, W), np.uint32)
    lib.orc_draw_frame(C.byref(sc), C.byref(u), W, H, 0, H, ob.ptr(rgba), ob.ptr(vis))
    assert (vis == 0).all()
    M = inv.astype(np.float64).reshape(4, 4).T                                           # column-major storage
    table = [lambda d: (-d[2], -d[1], d[0]), lambda d: (d[2], -d[1], d[0]), lambda d: (d[0], d[2], d[1]),
             lambda d: (d[0], -d[2], d[1]), lambda d: (d[0], -d[1], d[2]), lambda d: (-d[0], -d[1], d[2])]   # (sc, tc, ma) per face
    checked, worst, off = 0, 0, 0
    for y in range(H):
        for x in range(W):
            clip = np.array([(x + 0.5) / W * 2 - 1, 1 - (y + 0.5) / H * 2, 1.0, 1.0])
            t = M @ clip
            d = t[:3] / t[3]
            d = d / np.linalg.norm(d) * np.array([1.0, 1.0, -1.0])
            ax = np.abs(d)
            face = (4 if d[2] >= 0 else 5) if ax[2] >= ax[0] and ax[2] >= ax[1] else (2 if d[1] >= 0 else 3) if ax[1] >= ax[0] else (0 if d[0] >= 0 else 1)
            s_, t_, ma = table[face](d)
            fu, fv = (0.5 * s_ / abs(ma) + 0.5) * n - 0.5, (0.5 * t_ / abs(ma) + 0.5) * n - 0.5
            i0, j0 = math.floor(fu), math.floor(fv)
            if i0 < 0 or j0 < 0 or i0 + 1 >= n or j0 + 1 >= n:
                continue
            a, b = fu - i0, fv - j0
            f = sky[face].astype(np.float64)
            c = (f[j0, i0] * (1 - a) + f[j0, i0 + 1] * a) * (1 - b) + (f[j0 + 1, i0] * (1 - a) + f[j0 + 1, i0 + 1] * a) * b
            want = np.rint(np.clip(c[:3] ** (1 / 2.2), 0, 1) * 255).astype(int)
            dd = int(np.abs(want - rgba[y, x, :3].astype(int)).max())
            worst = max(worst, dd); off += dd > 0; checked += 1
    assert checked > 0.9 * W * H and worst <= 1 and off < 0.01 * checked, (checked, worst, off)
    assert (rgba[..., 3] == 255).all()

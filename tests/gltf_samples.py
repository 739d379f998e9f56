"""Small glTF files written for the loader tests (nothing of this kind ships with the reference)."""
import json
import os

import numpy as np


def write_interleaved_quad(directory):
    """Two triangles forming a 2 x 2 quad facing +z, stored the way the reference's loader cannot read:
    one interleaved buffer view (byteStride 36, accessors at byteOffset 0 / 12 / 24, four padding bytes
    per vertex), 32-bit indices behind 8 bytes of padding, and a material with no textures."""
    pos = np.float32([[-1, -1, 0], [1, -1, 0], [1, 1, 0], [-1, 1, 0]])
    nrm = np.float32([[0, 0, 1]] * 4)
    uv = np.float32([[0, 1], [1, 1], [1, 0], [0, 0]])
    inter = np.zeros((4, 9), dtype=np.float32)
    inter[:, 0:3], inter[:, 3:6], inter[:, 6:8] = pos, nrm, uv
    idx = np.uint32([0, 1, 2, 0, 2, 3])
    blob = inter.tobytes() + b"\xAA" * 8 + idx.tobytes()
    gltf = {
        "asset": {"version": "2.0"},
        "buffers": [{"uri": "quad.bin", "byteLength": len(blob)}],
        "bufferViews": [{"buffer": 0, "byteOffset": 0, "byteLength": 144, "byteStride": 36},
                        {"buffer": 0, "byteOffset": 144, "byteLength": 8 + 24}],
        "accessors": [{"bufferView": 0, "byteOffset": 0, "componentType": 5126, "count": 4, "type": "VEC3", "min": [-1, -1, 0], "max": [1, 1, 0]},
                      {"bufferView": 0, "byteOffset": 12, "componentType": 5126, "count": 4, "type": "VEC3"},
                      {"bufferView": 0, "byteOffset": 24, "componentType": 5126, "count": 4, "type": "VEC2"},
                      {"bufferView": 1, "byteOffset": 8, "componentType": 5125, "count": 6, "type": "SCALAR"}],
        "materials": [{"pbrMetallicRoughness": {"baseColorFactor": [1, 1, 1, 1]}, "emissiveFactor": [0.0, 0.0, 0.0]}],
        "meshes": [{"primitives": [{"attributes": {"POSITION": 0, "NORMAL": 1, "TEXCOORD_0": 2}, "indices": 3, "material": 0}]}],
        "nodes": [{"mesh": 0, "rotation": [0.0, 0.3826834, 0.0, 0.9238795], "translation": [0.25, 0.0, 0.0]}],
        "scenes": [{"nodes": [0]}], "scene": 0,
    }
    with open(os.path.join(directory, "quad.bin"), "wb") as f:
        f.write(blob)
    path = os.path.join(directory, "quad.gltf")
    with open(path, "w") as f:
        json.dump(gltf, f)
    expected = np.concatenate([pos, uv, nrm], axis=1).astype(np.float32).reshape(-1)
    return path, expected, idx.astype(np.uint16)


def write_large_two_primitive_mesh(directory, n=272):
    """A mesh of two triangle primitives: an n x n vertex height field (n = 272: 73 984 vertices, beyond what
    16-bit indices reach) with 32-bit indices, and a four-vertex quad behind it with 8-bit indices.  Loaded
    with the reference's limits lifted it becomes one geometry of n*n + 4 vertices with 32-bit indices.
    Returns (path, expected vertices [count][8], expected indices uint32)."""
    g = np.linspace(-1.0, 1.0, n, dtype=np.float32)
    x, y = np.meshgrid(g, g[::-1])
    z = (0.15 * np.sin(4.0 * x) * np.cos(3.0 * y)).astype(np.float32)
    pos = np.stack([x, y, z], -1).reshape(-1, 3).astype(np.float32)
    dzdx = 0.6 * np.cos(4.0 * x) * np.cos(3.0 * y); dzdy = -0.45 * np.sin(4.0 * x) * np.sin(3.0 * y)
    nrm = np.stack([-dzdx, -dzdy, np.ones_like(x)], -1).reshape(-1, 3)
    nrm = (nrm / np.linalg.norm(nrm, axis=1, keepdims=True)).astype(np.float32)
    uv = np.stack([(x + 1) * 0.5, (1 - y) * 0.5], -1).reshape(-1, 2).astype(np.float32)
    r, c = np.mgrid[0:n - 1, 0:n - 1]
    a = (r * n + c).reshape(-1); b = a + 1; d = a + n; e = d + 1      # a b / d e, rows run downwards in y
    idx = np.stack([a, d, b, b, d, e], -1).reshape(-1).astype(np.uint32)   # counter-clockwise seen from +z
    qpos = np.float32([[-1.2, -1.2, -0.5], [1.2, -1.2, -0.5], [1.2, 1.2, -0.5], [-1.2, 1.2, -0.5]])
    qnrm = np.float32([[0, 0, 1]] * 4); quv = np.float32([[0, 1], [1, 1], [1, 0], [0, 0]])
    qidx = np.uint8([0, 1, 2, 0, 2, 3])
    chunks = [pos.tobytes(), nrm.tobytes(), uv.tobytes(), idx.tobytes(), qpos.tobytes(), qnrm.tobytes(), quv.tobytes(), qidx.tobytes() + b"\0\0"]
    offs = np.cumsum([0] + [len(ch) for ch in chunks])
    views = [{"buffer": 0, "byteOffset": int(offs[k]), "byteLength": len(chunks[k])} for k in range(8)]
    views[7]["byteLength"] = 6
    nv = n * n
    acc = [{"bufferView": 0, "componentType": 5126, "count": nv, "type": "VEC3"}, {"bufferView": 1, "componentType": 5126, "count": nv, "type": "VEC3"},
           {"bufferView": 2, "componentType": 5126, "count": nv, "type": "VEC2"}, {"bufferView": 3, "componentType": 5125, "count": int(idx.size), "type": "SCALAR"},
           {"bufferView": 4, "componentType": 5126, "count": 4, "type": "VEC3"}, {"bufferView": 5, "componentType": 5126, "count": 4, "type": "VEC3"},
           {"bufferView": 6, "componentType": 5126, "count": 4, "type": "VEC2"}, {"bufferView": 7, "componentType": 5121, "count": 6, "type": "SCALAR"}]
    gltf = {
        "asset": {"version": "2.0"}, "buffers": [{"uri": "field.bin", "byteLength": int(offs[-1])}], "bufferViews": views, "accessors": acc,
        "materials": [{"pbrMetallicRoughness": {}, "emissiveFactor": [0.0, 0.0, 0.0]}],
        "meshes": [{"primitives": [{"attributes": {"POSITION": 0, "NORMAL": 1, "TEXCOORD_0": 2}, "indices": 3, "material": 0},
                                   {"attributes": {"POSITION": 4, "NORMAL": 5, "TEXCOORD_0": 6}, "indices": 7, "material": 0, "mode": 4},
                                   {"attributes": {"POSITION": 4}, "indices": 7, "mode": 1}]}],       # a line primitive: skipped
        "nodes": [{"mesh": 0, "rotation": [0.2588190, 0.0, 0.0, 0.9659258], "scale": [1.5, 1.5, 1.5]}],
        "scenes": [{"nodes": [0]}], "scene": 0,
    }
    with open(os.path.join(directory, "field.bin"), "wb") as f:
        f.write(b"".join(chunks))
    path = os.path.join(directory, "field.gltf")
    with open(path, "w") as f:
        json.dump(gltf, f)
    verts = np.concatenate([np.concatenate([pos, uv, nrm], 1), np.concatenate([qpos, quv, qnrm], 1)]).astype(np.float32)
    indices = np.concatenate([idx, qidx.astype(np.uint32) + np.uint32(nv)])
    return path, verts, indices


def write_scattered_triangles(directory, n=6000, seed=7):
    """n triangles facing +z, scattered over more than the default view in random order: most a few pixels across at
    1280 x 720, one in 250 hundreds of pixels.  Neighbours in the index buffer are nowhere near each other on screen,
    unlike those of a modelled mesh.  Returns the path."""
    rng = np.random.default_rng(seed)
    centre = np.stack([rng.uniform(-3.9, 3.9, n), rng.uniform(-2.2, 2.2, n), rng.uniform(-0.6, 0.6, n)], -1)
    size = np.where(rng.random(n) < 0.004, rng.uniform(0.4, 1.2, n), rng.uniform(0.008, 0.06, n))
    ang = rng.uniform(0.0, 2.0 * np.pi, n)[:, None] + np.float64([0.0, 2.1, 4.2]) + rng.uniform(-0.4, 0.4, (n, 3))   # increasing: counter-clockwise
    corner = centre[:, None, :] + np.stack([np.cos(ang), np.sin(ang), np.zeros_like(ang)], -1) * size[:, None, None]
    corner[:, :, 2] += rng.uniform(-0.05, 0.05, (n, 3))
    pos = corner.reshape(-1, 3).astype(np.float32)
    nrm = np.tile(np.float32([0, 0, 1]), (3 * n, 1))
    uv = rng.random((3 * n, 2)).astype(np.float32)
    idx = np.arange(3 * n, dtype=np.uint32)
    chunks = [pos.tobytes(), nrm.tobytes(), uv.tobytes(), idx.tobytes()]
    offs = np.cumsum([0] + [len(ch) for ch in chunks])
    views = [{"buffer": 0, "byteOffset": int(offs[k]), "byteLength": len(chunks[k])} for k in range(4)]
    acc = [{"bufferView": 0, "componentType": 5126, "count": 3 * n, "type": "VEC3"}, {"bufferView": 1, "componentType": 5126, "count": 3 * n, "type": "VEC3"},
           {"bufferView": 2, "componentType": 5126, "count": 3 * n, "type": "VEC2"}, {"bufferView": 3, "componentType": 5125, "count": 3 * n, "type": "SCALAR"}]
    gltf = {"asset": {"version": "2.0"}, "buffers": [{"uri": "scatter.bin", "byteLength": int(offs[-1])}], "bufferViews": views, "accessors": acc,
            "materials": [{"pbrMetallicRoughness": {"baseColorFactor": [0.9, 0.6, 0.3, 1.0]}, "emissiveFactor": [0.0, 0.0, 0.0]}],
            "meshes": [{"primitives": [{"attributes": {"POSITION": 0, "NORMAL": 1, "TEXCOORD_0": 2}, "indices": 3, "material": 0}]}],
            "nodes": [{"mesh": 0}], "scenes": [{"nodes": [0]}], "scene": 0}
    with open(os.path.join(directory, "scatter.bin"), "wb") as f:
        f.write(b"".join(chunks))
    path = os.path.join(directory, "scatter.gltf")
    with open(path, "w") as f:
        json.dump(gltf, f)
    return path

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

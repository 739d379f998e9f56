"""glTF loader with the reference's surface (src/glTF.ts): `glTFLoader().load(path)` then
`get_geometries()`.  `url` is a filesystem path.  Behaviour follows the reference, including its
limits: one buffer view per attribute read from its start, a fixed 8-float vertex, uint16 indices,
the last primitive of a node wins, and node matrix = I * scale * translate * rotation.

`glTFLoader(reference_limits=False)` reads accessors the way the glTF 2.0 specification defines
them instead (accessor.byteOffset, bufferView.byteStride, 8/16/32-bit indices), joins all triangle
primitives of a node's mesh into one geometry drawn with the first primitive's material (the
renderer binds one material per draw), and tolerates a material without some of the five textures
(the missing URI is None and the renderer substitutes a neutral 1x1 texture).  The joined indices
are uint16 while the vertex count allows it and uint32 beyond."""
import json
import os
from dataclasses import dataclass, field

import numpy as np

from . import mathx

COMPONENT_TYPE = {5120: "sint8", 5121: "uint8", 5122: "sint16", 5123: "uint16", 5125: "uint32", 5126: "float32"}   # glTF.ts:11-18
COMPONENT_COUNT = {"SCALAR": "", "VEC2": "x2", "VEC3": "x3", "VEC4": "x4", "MAT2": "x8", "MAT3": "x12", "MAT4": "x16"}   # glTF.ts:20-28


def calculate_byte_length(fmt=None):
    """utils.ts:26-39."""
    return {"float32": 4, "float32x2": 8, "float32x3": 12, "float32x4": 16}.get(fmt, 0)


def calculate_mip_level_count(*sizes):
    """utils.ts:20-24."""
    import math
    return int(1 + math.log2(max(sizes)))


@dataclass
class Layout:
    format: str
    offset: int


@dataclass
class Geometry:
    """glTF.ts:35-51."""
    model: np.ndarray
    vertices: np.ndarray
    array_stride: int
    position: Layout
    normal: Layout
    tex_coord: Layout
    indices: np.ndarray
    textures: dict = field(default_factory=dict)


_NP_TYPE = {5120: np.int8, 5121: np.uint8, 5122: np.int16, 5123: np.uint16, 5125: np.uint32, 5126: np.float32}
_COMPONENTS = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT2": 4, "MAT3": 9, "MAT4": 16}

# neutral stand-ins for material textures a file does not have (RGBA8): white base colour, roughness 1 /
# metalness 1 (scaled by nothing in the reference shader), flat normal, black emissive, no occlusion
DEFAULT_TEXEL = {"baseColorURI": (255, 255, 255, 255), "metallicRoughnessURI": (255, 255, 255, 255), "normalURI": (128, 128, 255, 255),
                 "emissiveURI": (0, 0, 0, 255), "occlusionURI": (255, 255, 255, 255)}


class glTFLoader:  # noqa: N801 - the reference's class name
    def __init__(self, reference_limits=True):
        self.reference_limits = reference_limits
        self.pathname = ""
        self.gltf = None
        self.buffers = []
        self.geometries = []

    def load(self, url):
        index = url.rfind("/")
        self.pathname = url[:index + 1]
        self._load_json(url)
        self._load_bin_buffers()
        self._generate_geometries()

    def get_geometries(self):
        return self.geometries

    def _load_json(self, url):
        if not os.path.isfile(url):
            raise RuntimeError(f"Failed to fetch .gltf file: {url}")            # glTF.ts:80
        with open(url, "r", encoding="utf-8") as f:
            self.gltf = json.load(f)
        if self.gltf["asset"]["version"] != "2.0":
            raise RuntimeError(f"Unsupported glTF version: {self.gltf['asset']['version']}")   # glTF.ts:85

    def _load_bin_buffers(self):
        self.buffers = []
        for buffer in self.gltf["buffers"]:
            path = self.pathname + buffer["uri"]
            if not os.path.isfile(path):
                raise RuntimeError(f"Failed to fetch .bin file: {buffer['uri']}")  # glTF.ts:96
            with open(path, "rb") as f:
                data = f.read()
            if len(data) != buffer["byteLength"]:
                raise RuntimeError("Wrong buffer byte length.")                   # glTF.ts:101
            self.buffers.append(data)

    def _view_f32(self, accessor):
        view = self.gltf["bufferViews"][accessor["bufferView"]]
        return np.frombuffer(self.buffers[view["buffer"]], dtype=np.float32, count=view["byteLength"] // 4,
                             offset=view.get("byteOffset", 0))

    def _accessor(self, index):
        """Accessor data as the glTF 2.0 specification lays it out: [count, components]."""
        acc = self.gltf["accessors"][index]
        view = self.gltf["bufferViews"][acc["bufferView"]]
        dtype = np.dtype(_NP_TYPE[acc["componentType"]])
        comps = _COMPONENTS[acc["type"]]
        start = view.get("byteOffset", 0) + acc.get("byteOffset", 0)
        stride = view.get("byteStride", 0) or dtype.itemsize * comps
        raw = np.frombuffer(self.buffers[view["buffer"]], dtype=np.uint8)
        count = acc["count"]
        if start + (count - 1) * stride + dtype.itemsize * comps > raw.size:
            raise RuntimeError("Accessor reaches past the end of its buffer.")
        rows = np.lib.stride_tricks.as_strided(raw[start:], shape=(count, dtype.itemsize * comps), strides=(stride, 1))
        return np.ascontiguousarray(rows).view(dtype).reshape(count, comps)

    def _general_primitive(self, primitive):
        g = self.gltf
        pos = self._accessor(primitive["attributes"]["POSITION"]).astype(np.float32)
        count = pos.shape[0]
        attrs = primitive["attributes"]
        uv = self._accessor(attrs["TEXCOORD_0"]).astype(np.float32) if "TEXCOORD_0" in attrs else np.zeros((count, 2), np.float32)
        nrm = self._accessor(attrs["NORMAL"]).astype(np.float32) if "NORMAL" in attrs else np.tile(np.float32([0, 0, 1]), (count, 1))
        vertices = np.concatenate([pos[:, :3], uv[:, :2], nrm[:, :3]], axis=1).astype(np.float32).reshape(-1)
        if "indices" in primitive:
            indices = self._accessor(primitive["indices"]).reshape(-1).astype(np.uint32)
        else:
            indices = np.arange(count, dtype=np.uint32)
        if indices.size and int(indices.max()) >= count:
            raise RuntimeError("Index refers to a vertex the primitive does not have.")
        material = g.get("materials", [{}])[primitive.get("material", 0)] if g.get("materials") else {}
        pbr = material.get("pbrMetallicRoughness", {})

        def uri(slot):
            if slot is None or "index" not in slot:
                return None
            return self.pathname + g["images"][g["textures"][slot["index"]]["source"]]["uri"]

        textures = {
            "baseColorURI": uri(pbr.get("baseColorTexture")), "metallicRoughnessURI": uri(pbr.get("metallicRoughnessTexture")),
            "normalURI": uri(material.get("normalTexture")), "emissive": list(material.get("emissiveFactor", [0.0, 0.0, 0.0])),
            "emissiveURI": uri(material.get("emissiveTexture")), "occlusionURI": uri(material.get("occlusionTexture")),
        }
        return vertices, indices, textures

    def _generate_geometries(self):
        g = self.gltf
        nodes = g["nodes"]
        self.geometries = [None] * len(nodes)
        for index, node in enumerate(nodes):
            if "matrix" in node:
                model = np.asarray(node["matrix"], dtype=np.float32)                  # glTF.ts:113-114
            else:
                model = mathx.identity()
                if "scale" in node:
                    model = mathx.scale(model, node["scale"])                         # glTF.ts:118-120
                if "translation" in node:
                    model = mathx.translate(model, node["translation"])               # glTF.ts:122-124
                if "rotation" in node:
                    q = np.asarray(node["rotation"], dtype=np.float32)
                    model = mathx.mul(model, mathx.from_quat(q))                      # glTF.ts:126-129
            mesh = g["meshes"][node["mesh"]]
            if not self.reference_limits:
                parts = [self._general_primitive(pr) for pr in mesh["primitives"] if pr.get("mode", 4) == 4]
                if parts:
                    first = np.cumsum([0] + [v.size // 8 for v, _, _ in parts])
                    indices = np.concatenate([ix + np.uint32(first[k]) for k, (_, ix, _) in enumerate(parts)])
                    if first[-1] <= 65536:
                        indices = indices.astype(np.uint16)
                    self.geometries[index] = Geometry(model=model, vertices=np.concatenate([v for v, _, _ in parts]), array_stride=32,
                                                      position=Layout("float32x3", 0), tex_coord=Layout("float32x2", 12), normal=Layout("float32x3", 20),
                                                      indices=indices, textures=parts[0][2])
                continue
            for primitive in mesh["primitives"]:
                if "mode" in primitive and primitive["mode"] != 4:                    # glTF.ts:134-136
                    continue
                pos_acc = g["accessors"][primitive["attributes"]["POSITION"]]
                uv_acc = g["accessors"][primitive["attributes"]["TEXCOORD_0"]]
                nrm_acc = g["accessors"][primitive["attributes"]["NORMAL"]]
                pos_type = COMPONENT_TYPE[pos_acc["componentType"]] + COMPONENT_COUNT[pos_acc["type"]]
                uv_type = COMPONENT_TYPE[uv_acc["componentType"]] + COMPONENT_COUNT[uv_acc["type"]]
                nrm_type = COMPONENT_TYPE[nrm_acc["componentType"]] + COMPONENT_COUNT[nrm_acc["type"]]
                array_stride = calculate_byte_length(pos_type) + calculate_byte_length(uv_type) + calculate_byte_length(nrm_type)
                count = pos_acc["count"]
                pos, uv, nrm = self._view_f32(pos_acc), self._view_f32(uv_acc), self._view_f32(nrm_acc)
                vertices = np.zeros(count * array_stride // 4, dtype=np.float32)
                v = vertices.reshape(count, -1) if array_stride == 32 else None
                if v is None:
                    raise RuntimeError("vertex layout other than float32x3 + float32x2 + float32x3 is not handled by the reference loader")
                v[:, 0:3] = pos[:count * 3].reshape(count, 3)                         # glTF.ts:193-214 (stride 8)
                v[:, 3:5] = uv[:count * 2].reshape(count, 2)
                v[:, 5:8] = nrm[:count * 3].reshape(count, 3)
                idx_view = g["bufferViews"][g["accessors"][primitive["indices"]]["bufferView"]]
                indices = np.frombuffer(self.buffers[idx_view["buffer"]], dtype=np.uint16, count=idx_view["byteLength"] // 2,
                                        offset=idx_view.get("byteOffset", 0)).copy()   # glTF.ts:222-226
                material = g["materials"][primitive["material"]]

                def image_uri(tex_index):
                    return self.pathname + g["images"][g["textures"][tex_index]["source"]]["uri"]

                pbr = material["pbrMetallicRoughness"]                                 # glTF.ts:228-252: no guards, KeyError like the reference's TypeError
                textures = {
                    "baseColorURI": image_uri(pbr["baseColorTexture"]["index"]),
                    "metallicRoughnessURI": image_uri(pbr["metallicRoughnessTexture"]["index"]),
                    "normalURI": image_uri(material["normalTexture"]["index"]),
                    "emissive": list(material["emissiveFactor"]),
                    "emissiveURI": image_uri(material["emissiveTexture"]["index"]),
                    "occlusionURI": image_uri(material["occlusionTexture"]["index"]),
                }
                self.geometries[index] = Geometry(
                    model=model, vertices=vertices, array_stride=array_stride,
                    position=Layout(pos_type, calculate_byte_length()),
                    tex_coord=Layout(uv_type, calculate_byte_length(pos_type)),
                    normal=Layout(nrm_type, calculate_byte_length(pos_type) + calculate_byte_length(uv_type)),
                    indices=indices, textures=textures)

'use strict';
// glTF 2.0 loader with the reference's surface and limits (src/glTF.ts): `new glTFLoader()`,
// `await load(url)`, `getGeometries()`.  `url` is a filesystem path.  Like the reference it reads
// whole bufferViews (accessor offsets and strides are ignored), expects float32 POSITION /
// TEXCOORD_0 / NORMAL and uint16 indices, and takes every material texture without guards.
// `new glTFLoader({ referenceLimits: false })` reads accessors as the glTF 2.0 specification defines
// them instead (accessor.byteOffset, bufferView.byteStride, 8/16/32-bit indices), joins all triangle
// primitives of a node's mesh into one geometry drawn with the first primitive's material (the
// renderer binds one material per draw) and tolerates missing material textures (URI null; the
// renderer substitutes a neutral 1x1 texture).  The joined indices are a Uint16Array while the vertex
// count allows it and a Uint32Array beyond.
const fs = require('fs');
const { mat4 } = require('./mat.js');
const { calculateByteLength } = require('./utils.js');

const ComponentType = { 5120: 'sint8', 5121: 'uint8', 5122: 'sint16', 5123: 'uint16', 5125: 'uint32', 5126: 'float32' };
const ComponentCount = { SCALAR: '', VEC2: 'x2', VEC3: 'x3', VEC4: 'x4', MAT2: 'x8', MAT3: 'x12', MAT4: 'x16' };

const TypedArrayOf = { 5120: Int8Array, 5121: Uint8Array, 5122: Int16Array, 5123: Uint16Array, 5125: Uint32Array, 5126: Float32Array };
const ComponentsOf = { SCALAR: 1, VEC2: 2, VEC3: 3, VEC4: 4, MAT2: 4, MAT3: 9, MAT4: 16 };
// neutral stand-ins for material textures a file does not have (RGBA8)
const DEFAULT_TEXEL = { baseColorURI: [255, 255, 255, 255], metallicRoughnessURI: [255, 255, 255, 255], normalURI: [128, 128, 255, 255],
  emissiveURI: [0, 0, 0, 255], occlusionURI: [255, 255, 255, 255] };

class glTFLoader {   // eslint-disable-line camelcase -- the reference's class name
  constructor(options = {}) {
    this.referenceLimits = options.referenceLimits !== false;
    this.pathname = '';
    this.gltf = null;
    this.buffers = [];
    this.geometries = [];
  }

  async load(url) {
    this.pathname = url.substring(0, url.lastIndexOf('/') + 1);
    await this.loadJSON(url);
    await this.loadBinBuffers();
    this.generateGeometries();
  }

  getGeometries() { return this.geometries; }

  async loadJSON(url) {
    let text;
    try {
      text = await fs.promises.readFile(url, 'utf8');
    } catch (err) {
      throw new Error(`Failed to fetch .gltf file: ${url}`);                    // glTF.ts:80
    }
    this.gltf = JSON.parse(text);
    if (this.gltf.asset.version !== '2.0') throw new Error(`Unsupported glTF version: ${this.gltf.asset.version}`);   // glTF.ts:85
  }

  async loadBinBuffers() {
    this.buffers = await Promise.all(this.gltf.buffers.map((buffer) => this.loadBinBuffer(buffer)));
  }

  async loadBinBuffer(buffer) {
    let bytes;
    try {
      bytes = await fs.promises.readFile(this.pathname + buffer.uri);
    } catch (err) {
      throw new Error(`Failed to fetch .bin file: ${buffer.uri}`);              // glTF.ts:96
    }
    if (bytes.byteLength !== buffer.byteLength) throw new Error('Wrong buffer byte length.');   // glTF.ts:101
    // a private, aligned ArrayBuffer (a node Buffer may be a slice of a shared pool)
    return bytes.buffer.slice(bytes.byteOffset, bytes.byteOffset + bytes.byteLength);
  }

  floatView(accessor) {
    const view = this.gltf.bufferViews[accessor.bufferView];
    return new Float32Array(this.buffers[view.buffer], view.byteOffset || 0, view.byteLength / 4);
  }

  // accessor data as the glTF 2.0 specification lays it out: a flat array of count * components values
  accessor(index) {
    const acc = this.gltf.accessors[index];
    const view = this.gltf.bufferViews[acc.bufferView];
    const Typed = TypedArrayOf[acc.componentType];
    const comps = ComponentsOf[acc.type];
    const size = Typed.BYTES_PER_ELEMENT * comps;
    const start = (view.byteOffset || 0) + (acc.byteOffset || 0);
    const stride = view.byteStride || size;
    const bytes = new Uint8Array(this.buffers[view.buffer]);
    if (start + (acc.count - 1) * stride + size > bytes.length) throw new Error('Accessor reaches past the end of its buffer.');
    const packed = new Uint8Array(acc.count * size);
    for (let i = 0; i < acc.count; ++i) packed.set(bytes.subarray(start + i * stride, start + i * stride + size), i * size);
    return { data: new Typed(packed.buffer), comps, count: acc.count };
  }

  generalPrimitive(primitive) {
    const g = this.gltf;
    const pos = this.accessor(primitive.attributes.POSITION);
    const count = pos.count;
    const uv = primitive.attributes.TEXCOORD_0 !== undefined ? this.accessor(primitive.attributes.TEXCOORD_0) : null;
    const nrm = primitive.attributes.NORMAL !== undefined ? this.accessor(primitive.attributes.NORMAL) : null;
    const vertices = new Float32Array(count * 8);
    for (let i = 0; i < count; ++i) {
      for (let k = 0; k < 3; ++k) vertices[8 * i + k] = pos.data[pos.comps * i + k];
      for (let k = 0; k < 2; ++k) vertices[8 * i + 3 + k] = uv ? uv.data[uv.comps * i + k] : 0;
      for (let k = 0; k < 3; ++k) vertices[8 * i + 5 + k] = nrm ? nrm.data[nrm.comps * i + k] : (k === 2 ? 1 : 0);
    }
    let indices;
    if (primitive.indices !== undefined) indices = Uint32Array.from(this.accessor(primitive.indices).data);
    else indices = Uint32Array.from({ length: count }, (_, i) => i);
    for (const v of indices) if (v >= count) throw new Error('Index refers to a vertex the primitive does not have.');
    const material = (g.materials && g.materials[primitive.material || 0]) || {};
    const pbr = material.pbrMetallicRoughness || {};
    const uri = (slot) => (slot && slot.index !== undefined ? this.pathname + g.images[g.textures[slot.index].source].uri : null);
    const textures = {
      baseColorURI: uri(pbr.baseColorTexture), metallicRoughnessURI: uri(pbr.metallicRoughnessTexture), normalURI: uri(material.normalTexture),
      emissive: (material.emissiveFactor || [0, 0, 0]).slice(0, 3), emissiveURI: uri(material.emissiveTexture), occlusionURI: uri(material.occlusionTexture),
    };
    return { vertices, indices, textures };
  }

  generateGeometries() {
    const g = this.gltf;
    this.geometries = new Array(g.nodes.length);
    g.nodes.forEach((node, index) => {
      let model;
      if (node.matrix) {
        model = new Float32Array(node.matrix);                                   // glTF.ts:113-114
      } else {
        model = mat4.identity();
        if (node.scale) model = mat4.scale(model, node.scale);                   // glTF.ts:118-120
        if (node.translation) model = mat4.translate(model, node.translation);   // glTF.ts:122-124
        if (node.rotation) model = mat4.mul(model, mat4.fromQuat(new Float32Array(node.rotation)));   // glTF.ts:126-129
      }
      if (!this.referenceLimits) {
        const parts = g.meshes[node.mesh].primitives.filter((pr) => pr.mode === undefined || pr.mode === 4).map((pr) => this.generalPrimitive(pr));
        if (parts.length === 0) return;
        const nVerts = parts.reduce((n, part) => n + part.vertices.length / 8, 0), nIndices = parts.reduce((n, part) => n + part.indices.length, 0);
        const vertices = new Float32Array(nVerts * 8);
        const indices = nVerts <= 65536 ? new Uint16Array(nIndices) : new Uint32Array(nIndices);
        let firstVertex = 0, firstIndex = 0;
        for (const part of parts) {
          vertices.set(part.vertices, firstVertex * 8);
          for (let i = 0; i < part.indices.length; ++i) indices[firstIndex + i] = part.indices[i] + firstVertex;
          firstVertex += part.vertices.length / 8; firstIndex += part.indices.length;
        }
        this.geometries[index] = { model, vertices, arrayStride: 32, position: { format: 'float32x3', offset: 0 },
          texCoord: { format: 'float32x2', offset: 12 }, normal: { format: 'float32x3', offset: 20 }, indices, textures: parts[0].textures };
        return;
      }
      for (const primitive of g.meshes[node.mesh].primitives) {
        if (primitive.mode !== undefined && primitive.mode !== 4) continue;     // glTF.ts:134-136: triangle lists only
        const posAcc = g.accessors[primitive.attributes.POSITION];
        const uvAcc = g.accessors[primitive.attributes.TEXCOORD_0];
        const nrmAcc = g.accessors[primitive.attributes.NORMAL];
        const posType = ComponentType[posAcc.componentType] + ComponentCount[posAcc.type];
        const uvType = ComponentType[uvAcc.componentType] + ComponentCount[uvAcc.type];
        const nrmType = ComponentType[nrmAcc.componentType] + ComponentCount[nrmAcc.type];
        const arrayStride = calculateByteLength(posType) + calculateByteLength(uvType) + calculateByteLength(nrmType);
        if (arrayStride !== 32) {
          throw new Error('vertex layout other than float32x3 + float32x2 + float32x3 is not handled by the reference loader');
        }
        const count = posAcc.count;
        const pos = this.floatView(posAcc), uv = this.floatView(uvAcc), nrm = this.floatView(nrmAcc);
        const vertices = new Float32Array(count * 8);                            // glTF.ts:193-214: pos3, uv2, normal3
        for (let i = 0; i < count; ++i) {
          vertices[8 * i] = pos[3 * i]; vertices[8 * i + 1] = pos[3 * i + 1]; vertices[8 * i + 2] = pos[3 * i + 2];
          vertices[8 * i + 3] = uv[2 * i]; vertices[8 * i + 4] = uv[2 * i + 1];
          vertices[8 * i + 5] = nrm[3 * i]; vertices[8 * i + 6] = nrm[3 * i + 1]; vertices[8 * i + 7] = nrm[3 * i + 2];
        }
        const idxView = g.bufferViews[g.accessors[primitive.indices].bufferView];
        const indices = new Uint16Array(this.buffers[idxView.buffer].slice(idxView.byteOffset || 0, (idxView.byteOffset || 0) + idxView.byteLength));   // glTF.ts:222-226
        const material = g.materials[primitive.material];
        const imageURI = (texIndex) => this.pathname + g.images[g.textures[texIndex].source].uri;
        const pbr = material.pbrMetallicRoughness;                                // glTF.ts:228-252
        this.geometries[index] = {
          model,
          vertices,
          arrayStride,
          position: { format: posType, offset: calculateByteLength() },
          texCoord: { format: uvType, offset: calculateByteLength(posType) },
          normal: { format: nrmType, offset: calculateByteLength(posType) + calculateByteLength(uvType) },
          indices,
          textures: {
            baseColorURI: imageURI(pbr.baseColorTexture.index),
            metallicRoughnessURI: imageURI(pbr.metallicRoughnessTexture.index),
            normalURI: imageURI(material.normalTexture.index),
            emissive: material.emissiveFactor.slice(0, 3),
            emissiveURI: imageURI(material.emissiveTexture.index),
            occlusionURI: imageURI(material.occlusionTexture.index),
          },
        };
      }
    });
  }
}

module.exports = glTFLoader;
module.exports.ComponentType = ComponentType;
module.exports.ComponentCount = ComponentCount;
module.exports.DEFAULT_TEXEL = DEFAULT_TEXEL;

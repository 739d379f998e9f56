'use strict';
// Prints, as one JSON line, what the JavaScript host computes without a GPU: loader results,
// camera state after a fixed sequence of moves, and the frame's uniform block.  The Python test
// suite compares it with the Python host bit for bit.
const crypto = require('crypto');
const host = require('../index.js');

const sha = (typed) => crypto.createHash('sha256').update(Buffer.from(typed.buffer, typed.byteOffset, typed.byteLength)).digest('hex');
const hex = (typed) => Buffer.from(typed.buffer, typed.byteOffset, typed.byteLength).toString('hex');

(async () => {
  const [modelPath, hdrPath, jpegPath, flatPath, oldRlePath, generalModelPath, largeModelPath] = process.argv.slice(2);
  const out = {};
  const loader = new host.glTFLoader();
  await loader.load(modelPath);
  const g = loader.getGeometries()[0];
  out.geometry = { vertices: sha(g.vertices), indices: sha(g.indices), model: hex(g.model), arrayStride: g.arrayStride,
    position: g.position, texCoord: g.texCoord, normal: g.normal, textures: g.textures };
  const general = new host.glTFLoader({ referenceLimits: false });
  await general.load(generalModelPath);
  const gg = general.getGeometries()[0];
  out.generalGeometry = { vertices: sha(gg.vertices), indices: sha(gg.indices), model: hex(gg.model), textures: gg.textures };
  const large = new host.glTFLoader({ referenceLimits: false });
  await large.load(largeModelPath);
  const lg = large.getGeometries()[0];
  out.largeGeometry = { vertices: sha(lg.vertices), indices: sha(lg.indices), indexBytes: lg.indices.BYTES_PER_ELEMENT, model: hex(lg.model) };
  const hdrLoader = new host.HDRLoader();
  await hdrLoader.load(hdrPath);
  out.hdr = { width: hdrLoader.hdr.width, height: hdrLoader.hdr.height, data: sha(hdrLoader.hdr.data) };
  out.hdrOlder = [];
  for (const older of [flatPath, oldRlePath]) {
    const l = new host.HDRLoader({ referenceLimits: false });
    await l.load(older);
    out.hdrOlder.push({ width: l.hdr.width, height: l.hdr.height, data: sha(l.hdr.data) });
  }
  // a pixel with exponent byte 0, through the reference-faithful and the format-faithful loader
  out.hdrZero = [];
  for (const referenceLimits of [true, false]) {
    const l = new host.HDRLoader({ referenceLimits });
    l.buffer = Buffer.concat([Buffer.from('#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 1 +X 8\n'), Buffer.from([2, 2, 0, 8, 0x88, 200, 0x88, 100, 0x88, 50, 0x84, 0, 0x84, 129])]);
    l.parseHDR();
    out.hdrZero.push(Array.from(l.hdr.data.slice(0, 4)).concat(Array.from(l.hdr.data.slice(16, 20))));
  }
  const image = await host.utils.loadImageBitmap(jpegPath);
  out.image = { width: image.width, height: image.height, data: sha(image.data) };
  const r = new host.Renderer({ width: 1280, height: 720, hdr: hdrLoader.hdr });
  r.geometries = loader.getGeometries();
  out.uniforms = [hex(r.frameUniforms())];
  for (let i = 0; i < 3; ++i) r.camera.updateCameraRotation(-10, 4);
  out.uniforms.push(hex(r.frameUniforms()));
  r.camera.updateCameraDistance(30, -5);
  r.camera.updateCameraTarget(12, -7);
  r.textureOptions.minFilter = 'nearest';
  r.textureOptions.addressModeU = 'clamp-to-edge';
  r.textureOptions.mipmaps = false;
  out.uniforms.push(hex(r.frameUniforms()));
  out.camera = { eye: hex(r.camera.eye), up: hex(r.camera.up), right: hex(r.camera.right), target: hex(r.camera.target), distance: r.camera.distance };
  out.mips = [host.utils.calculateMipLevelCount(2048, 2048), host.utils.calculateMipLevelCount(1, 1), host.utils.calculateMipLevelCount(4096, 2048), host.utils.calculateMipLevelCount(300)];
  const errors = {};
  const attempt = async (name, fn) => { try { await fn(); errors[name] = null; } catch (err) { errors[name] = err.message; } };
  await attempt('gltfMissing', () => new host.glTFLoader().load('/nonexistent/model.gltf'));
  await attempt('hdrMissing', () => new host.HDRLoader().load('/nonexistent/sky.hdr'));
  await attempt('hdrFormat', () => { const l = new host.HDRLoader(); l.buffer = Buffer.from('#?RADIANCE\nFORMAT=nope\n\n-Y 2 +X 8\n'); l.parseHDR(); });
  await attempt('hdrFlat', () => { const l = new host.HDRLoader(); l.buffer = Buffer.concat([Buffer.from('#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 1 +X 8\n'), Buffer.alloc(32, 7)]); l.parseHDR(); });
  await attempt('noDevice', () => new host.Renderer({ hdr: hdrLoader.hdr, device: 999 }).render());
  out.errors = errors;
  console.log(JSON.stringify(out));
})().catch((err) => { console.error(err.stack); process.exit(1); });

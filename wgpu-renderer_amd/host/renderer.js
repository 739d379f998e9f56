'use strict';
// Headless `Renderer` with the surface of the reference's class (src/renderer.ts:29-2052):
// `new Renderer()` then `await renderer.render()`.  Every WebGPU object the reference creates
// becomes a call into libpbr355 through the N-API addon; the browser plumbing (canvas, adapter,
// dat.gui, requestAnimationFrame) is gone.  Headless additions are extra constructor options and
// properties (frame size, paths, the RGBA8 result), not changed signatures.
const path = require('path');
const addon = require('./addon.js');
const Camera = require('./camera.js');
const glTFLoader = require('./gltf.js');
const HDRLoader = require('./hdr.js');
const { mat4, utils } = require('./mat.js');
const { loadImageBitmap } = require('./utils.js');

const REPO = path.resolve(__dirname, '..', '..');
const DEFAULT_MODEL = path.join(REPO, 'assets', 'DamagedHelmet', 'DamagedHelmet.gltf');   // renderer.ts:283

// float offsets inside PbrFrameUniforms (include/pbr355.h)
const U = { model: 0, normal: 16, view: 32, projection: 48, invViewProj: 64, viewPos: 80, lightPos: 84, lightColor: 100, emissive: 116,
  samplerIndex: 120, useMipmaps: 121, floats: 124 };

class Renderer {
  constructor(options = {}) {
    this.camera = new Camera();                                        // renderer.ts:77
    this.width = options.width || 1280;                                // canvas size
    this.height = options.height || 720;
    this.deviceOrdinal = options.device || 0;
    this.modelPath = options.modelPath || DEFAULT_MODEL;
    this.hdrPath = options.hdrPath || null;                            // a .hdr file, or an HDR object via options.hdr
    this.hdrSource = options.hdr || null;
    this.skyboxSize = options.skyboxSize || 1440;                      // renderer.ts:983
    this.irradianceSize = options.irradianceSize || 32;                // renderer.ts:764
    this.prefilterSize = options.prefilterSize || 128;                 // renderer.ts:773
    this.prefilterMips = options.prefilterMips || 5;                   // renderer.ts:1545
    this.lutSize = options.lutSize || 512;                             // renderer.ts:779
    this.referenceLimits = options.referenceLimits !== false;          // false: glTF read per the specification (see gltf.js)
    // dat.gui defaults, renderer.ts:1762-1768, :1793-1795
    this.textureOptions = { mipmaps: true, addressModeU: 'repeat', addressModeV: 'repeat', magFilter: 'linear', minFilter: 'linear' };
    this.msaaOptions = { enable: false };
    this.geometries = null;
    this.hdrs = null;
    this.ctx = null;
    this.ready = false;
    this.pixels = null;                                                // RGBA8 of the last frame
    this.frameTime = 0;                                                // GPU ms of the last frame (the timestamp query of renderer.ts:2040-2044)
  }

  // renderer.ts:102-105; headless `run` is one frame
  async render() {
    await this.init();
    return this.drawFrame();
  }

  // renderer.ts:107-136
  async init() {
    if (this.ready) return;
    this.requestDevice();
    await this.loadModel();
    await this.loadHDR();
    this.createVertexBuffer();
    await this.createTexture();
    this.createSkybox();
    this.ready = true;
  }

  requestDevice() {
    if (!this.ctx) this.ctx = addon.create(this.deviceOrdinal);       // throws Error like renderer.ts:147, :154
  }

  destroy() {
    if (this.ctx) addon.destroy(this.ctx);
    this.ctx = null;
    this.ready = false;
  }

  async loadModel() {
    const loader = new glTFLoader({ referenceLimits: this.referenceLimits });   // renderer.ts:281-286
    await loader.load(this.modelPath);
    this.geometries = loader.getGeometries();
  }

  async loadHDR() {
    if (this.hdrSource) { this.hdrs = [this.hdrSource]; return; }
    if (!this.hdrPath) throw new Error('Renderer: no HDR environment given (options.hdrPath or options.hdr).');
    const loader = new HDRLoader();                                    // renderer.ts:288-294
    await loader.load(this.hdrPath);
    this.hdrs = [loader.hdr];
  }

  createVertexBuffer() {
    const g = this.geometries[0];                                      // renderer.ts:506, :525: only geometry 0 is drawn
    addon.uploadMesh(this.ctx, g.vertices, g.indices);
  }

  // renderer.ts:604-860: material textures with their mip chains, irradiance and prefilter cubes, BRDF LUT
  async createTexture() {
    const t = this.geometries[0].textures;
    const slots = [[addon.TEX_ALBEDO, t.baseColorURI, true], [addon.TEX_METALLIC_ROUGHNESS, t.metallicRoughnessURI, false],
      [addon.TEX_NORMAL, t.normalURI, false], [addon.TEX_EMISSIVE, t.emissiveURI, true], [addon.TEX_OCCLUSION, t.occlusionURI, false]];
    // a loader without the reference's limits may leave a texture out: neutral 1x1 stand-in
    const names = ['baseColorURI', 'metallicRoughnessURI', 'normalURI', 'emissiveURI', 'occlusionURI'];
    const bitmaps = await Promise.all(slots.map((s, i) => (s[1] === null
      ? { width: 1, height: 1, data: new Uint8Array(glTFLoader.DEFAULT_TEXEL[names[i]]) } : loadImageBitmap(s[1]))));
    slots.forEach((s, i) => addon.uploadTexture(this.ctx, s[0], bitmaps[i].data, bitmaps[i].width, bitmaps[i].height, s[2], true));
    const hdr = this.hdrs[0];
    addon.uploadHdr(this.ctx, hdr.data, hdr.width, hdr.height);
    addon.iblIrradiance(this.ctx, this.irradianceSize);
    addon.iblPrefilter(this.ctx, this.prefilterSize, this.prefilterMips, 1024);
    addon.iblBrdfLut(this.ctx, this.lutSize);
  }

  createSkybox() {
    addon.iblEquirectToCube(this.ctx, this.skyboxSize);               // renderer.ts:978-985
    addon.synchronize(this.ctx);
  }

  // what drawFrame writes into its uniform buffers, renderer.ts:1828-1932
  frameUniforms() {
    const f = new Float32Array(U.floats);
    const g = this.geometries[0];
    const model = g.model;
    const view = this.camera.view;
    const projection = mat4.perspective(utils.degToRad(45), this.width / this.height, 1, 100);   // renderer.ts:1846-1851
    f.set(model, U.model);
    f.set(mat4.transpose(mat4.inverse(model)), U.normal);             // renderer.ts:1837
    f.set(view, U.view);
    f.set(projection, U.projection);
    f.set(mat4.inverse(mat4.mul(projection, view)), U.invViewProj);   // renderer.ts:1867
    f.set(this.camera.position, U.viewPos);
    const lights = [[10, 10, 10], [-10, 10, 10], [10, -10, 10], [-10, -10, 10]];   // renderer.ts:1891-1894
    lights.forEach((p, i) => {
      f.set(p, U.lightPos + 4 * i);
      f.set([300, 300, 300], U.lightColor + 4 * i);                   // renderer.ts:1909-1912
    });
    f.set(g.textures.emissive, U.emissive);
    const o = this.textureOptions;
    const words = new Uint32Array(f.buffer);
    words[U.samplerIndex] = (o.addressModeU === 'repeat' ? 1 : 0) + (o.addressModeV === 'repeat' ? 2 : 0) + (o.magFilter === 'linear' ? 4 : 0) +
      (o.minFilter === 'linear' ? 8 : 0) + (this.msaaOptions.enable ? 16 : 0);   // renderer.ts:1927-1932
    words[U.useMipmaps] = o.mipmaps ? 1 : 0;                           // renderer.ts:1977
    return f;
  }

  // renderer.ts:1820-2051; options: { y0, y1, wantVis, wantStats }
  drawFrame(options = {}) {
    if (!this.ready) throw new Error('Renderer: init() has not completed.');
    const y0 = options.y0 || 0;
    const y1 = options.y1 === undefined ? this.height : options.y1;
    const rows = y1 - y0;
    const uniforms = this.frameUniforms();
    const rgba = new Uint8Array(this.width * rows * 4);
    // one triangle id per pixel, or one per sample when multisampling is on
    const vis = options.wantVis ? new Uint32Array(this.width * rows * (this.msaaOptions.enable ? 4 : 1)) : null;
    const stats = addon.drawFrame(this.ctx, new Uint8Array(uniforms.buffer), this.width, this.height, y0, y1, rgba, vis, true);
    this.pixels = rgba;
    this.frameTime = stats.msTotal;
    const result = { rgba, stats };
    if (vis) result.vis = vis;
    return result;
  }

  downloadMap(which, level = 0) { return addon.iblDownload(this.ctx, which, level); }
}

module.exports = Renderer;
module.exports.UNIFORM_LAYOUT = U;

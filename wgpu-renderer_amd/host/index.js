'use strict';
// Entry point: the exports a TypeScript/JavaScript host imports, and — run as a program — the
// counterpart of src/index.ts (`new Renderer().render()`), writing the frame as raw RGBA8.
const Renderer = require('./renderer.js');
const Camera = require('./camera.js');
const glTFLoader = require('./gltf.js');
const HDRLoader = require('./hdr.js');
const utils = require('./utils.js');
const mat = require('./mat.js');

module.exports = { Renderer, Camera, glTFLoader, HDRLoader, utils, mat };

if (require.main === module) {
  const fs = require('fs');
  const args = {};
  for (let i = 2; i + 1 < process.argv.length; i += 2) args[process.argv[i].replace(/^--/, '')] = process.argv[i + 1];
  const int = (name) => (args[name] === undefined ? undefined : parseInt(args[name], 10));
  const renderer = new Renderer({ width: int('width'), height: int('height'), hdrPath: args.hdr, modelPath: args.model, skyboxSize: int('skybox'),
    irradianceSize: int('irradiance'), prefilterSize: int('prefilter'), prefilterMips: int('prefilter-mips'), lutSize: int('lut'), device: int('device') });
  (async () => {
    await renderer.init();
    for (let i = 0; i < (int('orbit') || 0); ++i) renderer.camera.updateCameraRotation(-10, 0);
    if (args.msaa === '1') renderer.msaaOptions.enable = true;   // the MSAA checkbox, renderer.ts:1793-1800
    const frame = renderer.drawFrame({ wantVis: args.vis !== undefined });
    if (args.out) fs.writeFileSync(args.out, frame.rgba);
    if (args.vis) fs.writeFileSync(args.vis, Buffer.from(frame.vis.buffer));
    if (args.uniforms) fs.writeFileSync(args.uniforms, Buffer.from(renderer.frameUniforms().buffer));
    console.log(JSON.stringify({ width: renderer.width, height: renderer.height, stats: frame.stats }));
    renderer.destroy();
  })().catch((err) => { console.error(err.message); process.exit(1); });
}

'use strict';
// Loads the N-API addon built by the Makefile next to this file.  There is no software fallback:
// without the addon (and, at create(), without a gfx950 device) the renderer cannot run.
const path = require('path');

let addon;
try {
  addon = require(path.join(__dirname, 'pbr355_node.node'));
} catch (err) {
  throw new Error(`pbr355 addon is not built or cannot load (run make in ${__dirname}): ${err.message}`);
}
module.exports = addon;

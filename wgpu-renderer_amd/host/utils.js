'use strict';
// Counterparts of src/utils.ts.  `loadImageBitmap` resolves to { width, height, data } with RGBA8
// bytes, top row first, decoded without colour-space conversion (utils.ts:17) by the addon's JPEG
// decoder; `url` is a filesystem path.
const fs = require('fs');

function rand(min, max) {
  if (min === undefined) { min = 0; max = 1; } else if (max === undefined) { max = min; min = 0; }
  return min + Math.random() * (max - min);
}

async function loadImageBitmap(url) {
  const addon = require('./addon.js');
  let bytes;
  try {
    bytes = await fs.promises.readFile(url);
  } catch (err) {
    throw new Error(`Failed to fetch image file: ${url}`);
  }
  if (bytes.length >= 2 && bytes[0] === 0xFF && bytes[1] === 0xD8) return addon.decodeJpeg(bytes);
  throw new Error(`Unsupported image format (only JPEG is decoded): ${url}`);
}

// utils.ts:20-24
function calculateMipLevelCount(...sizes) {
  const maxSize = Math.max(...sizes);
  return (1 + Math.log2(maxSize)) | 0;
}

// utils.ts:26-39
function calculateByteLength(type) {
  switch (type) {
    case 'float32': return 4;
    case 'float32x2': return 8;
    case 'float32x3': return 12;
    case 'float32x4': return 16;
    default: return 0;
  }
}

// utils.ts:41-63
class RollingAverage {
  constructor(sampleCount = 100) {
    this.total = 0;
    this.samples = [];
    this.cursor = 0;
    this.sampleCount = sampleCount;
  }
  set value(value) {
    this.total += value - (this.samples[this.cursor] || 0);
    this.samples[this.cursor] = value;
    this.cursor = (this.cursor + 1) % this.sampleCount;
  }
  get value() { return this.total / this.samples.length; }
}

module.exports = { rand, loadImageBitmap, calculateMipLevelCount, calculateByteLength, RollingAverage };

'use strict';
// Radiance .hdr loader with the reference's surface (src/hdr.ts): `new HDRLoader()`,
// `await load(url)`, `.hdr` = { width, height, data: Float32Array rgba }.  `url` is a filesystem
// path.  By default the limits are the reference's: new-style RLE scanlines only, no special case
// for e == 0.  `new HDRLoader({ referenceLimits: false })` also reads the two older scanline encodings
// of the format (flat RGBE pixels, and runs marked by a (1,1,1,n) pixel), which the reference rejects, and
// decodes a pixel whose exponent byte is 0 to black, as the format defines it (the reference returns
// mantissa * 2^-128 / 255 for it, hdr.ts:105-116).
const fs = require('fs');

class HDRLoader {
  constructor(options = {}) {
    this.referenceLimits = options.referenceLimits !== false;
    this.buffer = null;
    this.width = 0;
    this.height = 0;
    this.data = null;
  }

  get hdr() { return { width: this.width, height: this.height, data: this.data }; }   // hdr.ts:15-21

  async load(url) {
    await this.loadBuffer(url);
    this.parseHDR();
  }

  async loadBuffer(url) {
    try {
      this.buffer = await fs.promises.readFile(url);
    } catch (err) {
      throw new Error(`Failed to fetch .hdr file: ${url}`);                     // hdr.ts:31
    }
  }

  parseHDR() {
    const bytes = this.buffer;
    const n = bytes.length;
    // hdr.ts:45-52: the header ends at the first newline after both "-Y " and " +X " were seen
    let index = 0;
    let header = '';
    while (index < n) {
      const ch = String.fromCharCode(bytes[index++]);
      header += ch;
      if (ch === '\n' && header.includes('-Y ') && header.includes(' +X ')) break;
    }
    if (!header.includes('32-bit_rle_rgbe') && !header.includes('32-bit_rle_xyze')) {
      throw new Error('HDRLoader: Unsupported .hdr format.');                   // hdr.ts:58
    }
    const match = /-Y (\d+) \+X (\d+)/.exec(header);
    if (!match) throw new Error('HDRLoader: No resolution found.');             // hdr.ts:63
    this.height = parseInt(match[1], 10);
    this.width = parseInt(match[2], 10);
    const w = this.width, h = this.height;
    const scan = new Uint8Array(w * h * 4);
    let row = 0;
    while (index < n) {
      if (index + 4 > n || bytes[index] !== 2 || bytes[index + 1] !== 2 || ((bytes[index + 2] << 8) + bytes[index + 3]) !== w) {
        if (this.referenceLimits) throw new Error('HDRLoader: Only support RLE format.');   // hdr.ts:81
        if (row >= h) break;
        index = this.oldScanline(bytes, index, scan, row);
        row += 1;
        continue;
      }
      index += 4;
      for (let c = 0; c < 4; ++c) {
        let pos = 0;
        while (pos < w) {
          let count = bytes[index++];
          if (count > 128) {                                                    // hdr.ts:89-95: a run
            count -= 128;
            const value = bytes[index++];
            if (row < h) for (let i = 0; i < count && pos + i < w; ++i) scan[(row * w + pos + i) * 4 + c] = value;
            pos += count;
          } else {                                                              // hdr.ts:96-100: literals
            if (row < h) for (let i = 0; i < count && pos + i < w; ++i) scan[(row * w + pos + i) * 4 + c] = bytes[index + i];
            index += count;
            pos += count;
          }
        }
      }
      row += 1;
    }
    // hdr.ts:105-116: rgb = mantissa * 2^(e - 128) / 255, alpha 1
    const data = new Float32Array(w * h * 4);
    for (let i = 0; i < w * h; ++i) {
      const e = scan[4 * i + 3];
      const scale = (e === 0 && !this.referenceLimits) ? 0 : Math.pow(2, e - 128) / 255;
      data[4 * i] = scan[4 * i] * scale;
      data[4 * i + 1] = scan[4 * i + 1] * scale;
      data[4 * i + 2] = scan[4 * i + 2] * scale;
      data[4 * i + 3] = 1;
    }
    this.data = data;
  }
}

// One scanline in the pre-1991 encodings: RGBE pixels one after the other; a pixel (1,1,1,n) repeats the
// previous pixel n times, consecutive such pixels shifting their counts by 8 bits each.
HDRLoader.prototype.oldScanline = function oldScanline(bytes, index, scan, row) {
  const w = this.width;
  const n = bytes.length;
  let pos = 0;
  let shift = 0;
  while (pos < w) {
    if (index + 4 > n) throw new Error('HDRLoader: Unexpected end of file.');
    const r = bytes[index], g = bytes[index + 1], b = bytes[index + 2], e = bytes[index + 3];
    index += 4;
    if (r === 1 && g === 1 && b === 1) {
      const count = e << shift;
      if (pos === 0 || pos + count > w) throw new Error('HDRLoader: Bad scanline data.');
      const from = (row * w + pos - 1) * 4;
      for (let i = 0; i < count; ++i) scan.copyWithin((row * w + pos + i) * 4, from, from + 4);
      pos += count;
      shift += 8;
    } else {
      const at = (row * w + pos) * 4;
      scan[at] = r; scan[at + 1] = g; scan[at + 2] = b; scan[at + 3] = e;
      pos += 1;
      shift = 0;
    }
  }
  return index;
};

module.exports = HDRLoader;

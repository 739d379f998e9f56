'use strict';
// Radiance .hdr loader with the reference's surface (src/hdr.ts): `new HDRLoader()`,
// `await load(url)`, `.hdr` = { width, height, data: Float32Array rgba }.  `url` is a filesystem
// path.  Same limits as the reference: new-style RLE scanlines only, no special case for e == 0.
const fs = require('fs');

class HDRLoader {
  constructor() {
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
        throw new Error('HDRLoader: Only support RLE format.');                 // hdr.ts:81
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
      const scale = Math.pow(2, scan[4 * i + 3] - 128) / 255;
      data[4 * i] = scan[4 * i] * scale;
      data[4 * i + 1] = scan[4 * i + 1] * scale;
      data[4 * i + 2] = scan[4 * i + 2] * scale;
      data[4 * i + 3] = 1;
    }
    this.data = data;
  }
}

module.exports = HDRLoader;

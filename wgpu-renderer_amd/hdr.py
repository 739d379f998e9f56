"""Radiance .hdr loading with the reference's surface (src/hdr.ts): `HDRLoader().load(path)`, then
`.hdr` = {width, height, data}.  Also a writer and the deterministic synthetic sky used because
the reference's own .hdr files are not part of its checkout."""
import os
import re
from dataclasses import dataclass

import numpy as np


@dataclass
class HDR:
    width: int
    height: int
    data: np.ndarray   # float32, width*height*4, rows top to bottom, alpha 1 (hdr.ts:112-115)


class HDRLoader:
    """`reference_limits=True` (default) keeps the reference's parser exactly, errors included: only
    new-style RLE scanlines.  With False the loader also reads the two older scanline encodings of
    the Radiance format (flat RGBE pixels, and runs marked by a (1,1,1,n) pixel), which the
    reference rejects with "Only support RLE format.", and gives a pixel whose exponent byte is 0
    the value the format defines for it, black, where the reference (hdr.ts:105-116 has no special
    case) returns mantissa * 2^-128 / 255."""

    def __init__(self, reference_limits=True):
        self.reference_limits = reference_limits
        self.buffer = b""
        self.width = 0
        self.height = 0
        self.data = None

    @property
    def hdr(self):
        return HDR(self.width, self.height, self.data)

    def load(self, url):
        if not os.path.isfile(url):
            raise RuntimeError(f"Failed to fetch .hdr file: {url}")               # hdr.ts:31
        with open(url, "rb") as f:
            self.buffer = f.read()
        self._parse_hdr()

    def load_bytes(self, data):
        self.buffer = bytes(data)
        self._parse_hdr()

    @staticmethod
    def _old_scanline(raw, index, out_row):
        return _old_scanline_impl(raw, index, out_row)

    def _parse_hdr(self):
        b = self.buffer
        n = len(b)
        # hdr.ts:45-52: the header ends at the first newline after both "-Y " and " +X " were seen
        index = 0
        header = ""
        while index < n:
            ch = chr(b[index])
            header += ch
            index += 1
            if ch == "\n" and "-Y " in header and " +X " in header:
                break
        if "32-bit_rle_rgbe" not in header and "32-bit_rle_xyze" not in header:
            raise RuntimeError("HDRLoader: Unsupported .hdr format.")             # hdr.ts:58
        m = re.search(r"-Y (\d+) \+X (\d+)", header)
        if not m:
            raise RuntimeError("HDRLoader: No resolution found.")                 # hdr.ts:63
        self.width, self.height = int(m.group(2)), int(m.group(1))
        w, h = self.width, self.height
        raw = np.frombuffer(b, dtype=np.uint8)
        scan = np.zeros((h, w, 4), dtype=np.uint8)
        row = 0
        while index < n:
            head = raw[index:index + 4]
            new_rle = len(head) == 4 and head[0] == 2 and head[1] == 2 and (int(head[2]) << 8) + int(head[3]) == w
            if not new_rle:
                if self.reference_limits:
                    raise RuntimeError("HDRLoader: Only support RLE format.")     # hdr.ts:81
                if row >= h:
                    break
                index = self._old_scanline(raw, index, scan[row])
                row += 1
                continue
            index += 4
            for c in range(4):
                pos = 0
                while pos < w:
                    count = int(raw[index]); index += 1
                    if count > 128:                                               # hdr.ts:89-95: a run
                        run = count - 128
                        if row < h:
                            scan[row, pos:pos + run, c] = raw[index]
                        index += 1
                        pos += run
                    else:                                                         # hdr.ts:96-100: literals
                        if row < h:
                            scan[row, pos:pos + count, c] = raw[index:index + count]
                        index += count
                        pos += count
            row += 1
        # hdr.ts:105-116: rgb = mantissa * 2^(e-128) / 255, no e == 0 special case
        e = scan[..., 3].astype(np.float64)
        scale = np.exp2(e - 128.0) / 255.0
        if not self.reference_limits:
            scale[scan[..., 3] == 0] = 0.0      # exponent byte 0: the format's encoding of black
        out = np.empty((h, w, 4), dtype=np.float32)
        out[..., :3] = (scan[..., :3].astype(np.float64) * scale[..., None]).astype(np.float32)
        out[..., 3] = 1.0
        self.data = out.reshape(-1)


def _old_scanline_impl(raw, index, out_row):
    """One scanline in the pre-1991 encodings: RGBE pixels one after the other; a pixel (1,1,1,n) repeats
    the previous pixel n times, consecutive such pixels shifting their counts by 8 bits each."""
    w = out_row.shape[0]
    n = len(raw)
    # fast path: a flat scanline with no run markers
    if index + 4 * w <= n:
        px = raw[index:index + 4 * w].reshape(w, 4)
        if not ((px[:, 0] == 1) & (px[:, 1] == 1) & (px[:, 2] == 1)).any():
            out_row[:] = px
            return index + 4 * w
    pos, shift = 0, 0
    while pos < w:
        if index + 4 > n:
            raise RuntimeError("HDRLoader: Unexpected end of file.")
        r, g, b, e = (int(v) for v in raw[index:index + 4])
        index += 4
        if r == 1 and g == 1 and b == 1:
            count = e << shift
            if pos == 0 or pos + count > w:
                raise RuntimeError("HDRLoader: Bad scanline data.")
            out_row[pos:pos + count] = out_row[pos - 1]
            pos += count
            shift += 8
        else:
            out_row[pos] = (r, g, b, e)
            pos += 1
            shift = 0
    return index


def float_to_rgbe(rgb):
    """rgb float array [..., 3] -> uint8 [..., 4] shared-exponent encoding (Radiance convention)."""
    rgb = np.asarray(rgb, dtype=np.float64)
    m = rgb.max(axis=-1)
    out = np.zeros(rgb.shape[:-1] + (4,), dtype=np.uint8)
    ok = m > 1e-32
    mant, exp = np.frexp(np.where(ok, m, 1.0))
    scale = np.where(ok, mant * 256.0 / np.where(ok, m, 1.0), 0.0)
    out[..., :3] = np.clip(rgb * scale[..., None], 0, 255).astype(np.uint8)
    out[..., 3] = np.where(ok, exp + 128, 0).astype(np.uint8)
    return out


def write_hdr_rle(path_or_none, rgb):
    """Encode [h, w, 3] floats as a new-RLE 32-bit_rle_rgbe file; returns the bytes (and writes them
    if a path is given).  Every scanline is stored as literal runs, which is valid RLE."""
    h, w, _ = rgb.shape
    if not 8 <= w < 32768:
        raise ValueError("new-RLE scanlines need 8 <= width < 32768")
    rgbe = float_to_rgbe(rgb)
    parts = [b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n" % (h, w)]
    for y in range(h):
        parts.append(bytes([2, 2, (w >> 8) & 255, w & 255]))
        for c in range(4):
            ch = rgbe[y, :, c].tobytes()
            for s in range(0, w, 128):
                chunk = ch[s:s + 128]
                parts.append(bytes([len(chunk)]) + chunk)
    data = b"".join(parts)
    if path_or_none:
        with open(path_or_none, "wb") as f:
            f.write(data)
    return data


def write_hdr_flat(path_or_none, rgb, old_rle=False):
    """Encode [h, w, 3] floats as flat RGBE scanlines; with old_rle, repeated pixels become (1,1,1,n) runs."""
    h, w, _ = rgb.shape
    rgbe = float_to_rgbe(rgb)
    parts = [b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n" % (h, w)]
    for y in range(h):
        row = rgbe[y]
        if not old_rle:
            parts.append(row.tobytes())
            continue
        x = 0
        while x < w:
            run = 1
            while x + run < w and run < 255 and (row[x + run] == row[x]).all():
                run += 1
            parts.append(row[x].tobytes())
            if run > 1:
                parts.append(bytes([1, 1, 1, run - 1]))
            x += run
    data = b"".join(parts)
    if path_or_none:
        with open(path_or_none, "wb") as f:
            f.write(data)
    return data


def synthetic_sky(width=4096, height=2048, seed=355):
    """Deterministic stand-in for the reference's missing puresky .hdr files: a vertical sky
    gradient L(v) = 0.3 + 2.0 (1-v)^2 tinted (0.6, 0.75, 1.0), a sun lobe of peak 5000 around
    direction (0.4, 0.6, 0.7) with a 2 degree Gaussian width, and exp(N(0, 0.25)) multiplicative
    noise per texel.  Returns an HDR (rgba32f, alpha 1)."""
    rng = np.random.default_rng(seed)
    v = (np.arange(height, dtype=np.float64) + 0.5) / height
    u = (np.arange(width, dtype=np.float64) + 0.5) / width
    # inverse of the equirect mapping the shaders use: u = atan2(z, x) / (2 pi) + 0.5, v = asin(y) / pi + 0.5
    phi = (u - 0.5) * 2.0 * np.pi
    theta = (v - 0.5) * np.pi
    dy = np.sin(theta)[:, None]
    dx = (np.cos(theta)[:, None]) * np.cos(phi)[None, :]
    dz = (np.cos(theta)[:, None]) * np.sin(phi)[None, :]
    sun = np.array([0.4, 0.6, 0.7]); sun /= np.linalg.norm(sun)
    cosang = np.clip(dx * sun[0] + dy * sun[1] + dz * sun[2], -1.0, 1.0)
    ang = np.arccos(cosang)
    lobe = 5000.0 * np.exp(-0.5 * (ang / np.deg2rad(2.0)) ** 2)
    # rows are stored top to bottom and v grows with asin(y); the gradient is brightest where v is small
    grad = 0.3 + 2.0 * (1.0 - v) ** 2
    base = grad[:, None, None] * np.array([0.6, 0.75, 1.0])[None, None, :]
    noise = np.exp(rng.normal(0.0, 0.25, size=(height, width)))
    rgb = (base + lobe[..., None]) * noise[..., None]
    data = np.empty((height, width, 4), dtype=np.float32)
    data[..., :3] = rgb.astype(np.float32)
    data[..., 3] = 1.0
    return HDR(width, height, data.reshape(-1))

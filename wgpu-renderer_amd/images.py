"""Image decode standing in for `loadImageBitmap` (src/utils.ts:13-18): JPEG/PNG file -> RGBA8,
top row first, no colour-space conversion.  The browser's decoder is not available; Pillow's
libjpeg output can differ from it by an LSB or two (IDCT / chroma upsampling choices)."""
import numpy as np
from PIL import Image


def load_image_bitmap(path):
    with Image.open(path) as im:
        rgba = np.asarray(im.convert("RGBA"), dtype=np.uint8)
    return np.ascontiguousarray(rgba)

"""Makes tests/golden/reference_result_helmet_crop.png: the helmet's region of the one rendered output the reference
holds, public/images/result.png (a 2560 x 1305 browser screenshot of the default view, README.md:11), at full
resolution.  The crop leaves out the dat.gui overlay (top right) and most of the sky, which comes from an HDR file
that is missing from the reference's snapshot.  Run in the build container, where /root/reference exists."""
import os

from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CROP = (960, 340, 1600, 1020)        # left, top, right, bottom in the 2560 x 1305 screenshot

if __name__ == "__main__":
    im = Image.open("/root/reference/public/images/result.png").convert("RGB")
    assert im.size == (2560, 1305), im.size
    im.crop(CROP).save(os.path.join(ROOT, "tests", "golden", "reference_result_helmet_crop.png"), optimize=True)

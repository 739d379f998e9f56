"""pbr355: MI355X (gfx950) implementation of the wgpu-renderer PBR frame path and IBL precompute.

The product is libpbr355.so (csrc/, C ABI in include/pbr355.h).  This package is the Python host
that mirrors the reference's Renderer / glTFLoader / Camera / HDRLoader classes on top of it."""
from . import _lib, mathx  # noqa: F401
from .camera import Camera  # noqa: F401
from .gltf import Geometry, glTFLoader  # noqa: F401
from .hdr import HDR, HDRLoader, synthetic_sky, write_hdr_rle  # noqa: F401
from .images import load_image_bitmap  # noqa: F401
from .renderer import Renderer  # noqa: F401

// Cube-map lookup helpers and the skybox fragment arithmetic (fs_main of src/shaders/skybox.wgsl:24-33),
// shared by the rasteriser (which shades the background pixels of the tile it has just resolved) and
// the surface resolve.  Every function enables FMA contraction for its own body, so the code is
// the same whether or not the including file is built without contraction.
#pragma once
#include "pbr_internal.h"
#include "devmath.h"

struct CubeCoord { int face; float sc, tc, rc; };

PBR_DEV CubeCoord cube_select(V3 r) {
#pragma clang fp contract(fast)
  const float ax = fabsf(r.x), ay = fabsf(r.y), az = fabsf(r.z);
  CubeCoord c;
  if (az >= ax && az >= ay) { c.face = r.z >= 0 ? 4 : 5; c.sc = r.z >= 0 ? r.x : -r.x; c.tc = -r.y; c.rc = r.z; }
  else if (ay >= ax)        { c.face = r.y >= 0 ? 2 : 3; c.sc = r.x; c.tc = r.y >= 0 ? r.z : -r.z; c.rc = r.y; }
  else                      { c.face = r.x >= 0 ? 0 : 1; c.sc = r.x >= 0 ? -r.z : r.z; c.tc = -r.y; c.rc = r.x; }
  return c;
}

PBR_DEV uint32_t pack_color(V3 color) {
#pragma clang fp contract(fast)
  // pow(c, 1/2.2) as exp2(log2(c) / 2.2) (shader.wgsl:157, skybox.wgsl:30): 0 -> 0, negative or NaN -> NaN -> 0
  const float g = 1.0f / 2.2f;
  return to_unorm8(exp2_fast(log2_fast(color.x) * g)) | (to_unorm8(exp2_fast(log2_fast(color.y) * g)) << 8) |
         (to_unorm8(exp2_fast(log2_fast(color.z) * g)) << 16) | 0xff000000u;
}


struct CubeFoot { uint32_t o; uint32_t m; float w[4]; };   // taps at o, o + 1, o + m, o + m + 1

// Texel offset (from cube.base) and weights of one seamless bilinear lookup on one level of a bordered
// cube: the border row / column of a face holds the adjoining texels of the neighbouring faces and a
// border corner the mean of the three texels around it, which is what a lookup across an edge or a
// corner blends, so the four taps always sit in one face.
PBR_DEV CubeFoot cube_foot(const DevCube& bordered, int level, const CubeCoord& c) {
#pragma clang fp contract(fast)
  CubeFoot f;
  const int n = max(1, bordered.size >> level), m = n + 2;
  const float ima = rcp_fast(fabsf(c.rc));
  const float s = 0.5f * (c.sc * ima) + 0.5f, t = 0.5f * (c.tc * ima) + 0.5f;
  const float tu = s * (float)n - 0.5f, tv = t * (float)n - 0.5f;
  const float fu = floorf(tu), fv = floorf(tv), a = tu - fu, b = tv - fv;
  const int i0 = min(max((int)fu, -1), n - 1), j0 = min(max((int)fv, -1), n - 1);
  f.w[0] = (1.0f - a) * (1.0f - b); f.w[1] = a * (1.0f - b); f.w[2] = (1.0f - a) * b; f.w[3] = a * b;
  f.o = bordered.off[level] + ((uint32_t)c.face * m + (uint32_t)(j0 + 1)) * m + (uint32_t)(i0 + 1);
  f.m = (uint32_t)m;
  return f;
}
PBR_DEV void cube_load4(const float4* base, const CubeFoot& f, float4 t[4]) {
  const float4* q = base + f.o;
  t[0] = q[0]; t[1] = q[1]; t[2] = q[f.m]; t[3] = q[f.m + 1];
}


PBR_DEV V3 weigh4(const float4 t[4], const float w[4]) {
#pragma clang fp contract(fast)
  return v3(((t[0].x * w[0] + t[1].x * w[1]) + t[2].x * w[2]) + t[3].x * w[3], ((t[0].y * w[0] + t[1].y * w[1]) + t[2].y * w[2]) + t[3].y * w[3],
            ((t[0].z * w[0] + t[1].z * w[1]) + t[2].z * w[2]) + t[3].z * w[3]);
}


// Skybox colour of four horizontally adjacent pixels (x0 .. x0+3, y): the clip-space position is
// unprojected with the inverse view-projection matrix and the bordered cube sampled bilinearly; all
// sixteen taps are issued before any is used.
PBR_DEV void sky_pixels4(const FrameParams* __restrict__ p, const DevCube& sky, int x0, int y, uint32_t out[4]) {
#pragma clang fp contract(fast)
  const float* m = p->u.inv_view_proj;
  const float two_over_w = 2.0f / (float)p->W;
  const float yn = 1.0f - ((float)y + 0.5f) * (2.0f / (float)p->H);
  CubeFoot foot[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float xn = ((float)(x0 + k) + 0.5f) * two_over_w - 1.0f;
    const float tx = ((m[0] * xn + m[4] * yn) + m[8]) + m[12];
    const float ty = ((m[1] * xn + m[5] * yn) + m[9]) + m[13];
    const float tz = ((m[2] * xn + m[6] * yn) + m[10]) + m[14];
    const float tw = ((m[3] * xn + m[7] * yn) + m[11]) + m[15];
    // normalize(xyz / w) only rescales the lookup direction by 1 / (w |xyz|); the cube lookup depends on
    // the direction alone, so only the sign of w is applied.
    const float sg = tw < 0.0f ? -1.0f : 1.0f;
    foot[k] = cube_foot(sky, 0, cube_select(v3(tx * sg, ty * sg, -(tz * sg))));
  }
  float4 tap[4][4];
#pragma unroll
  for (int k = 0; k < 4; ++k) cube_load4(sky.base, foot[k], tap[k]);
#pragma unroll
  for (int k = 0; k < 4; ++k) out[k] = pack_color(weigh4(tap[k], foot[k].w));
}

// The skybox pass is cut into the same 32 x 32 pixel tiles as the rasteriser.  A tile can be shaded by
// any 256-thread workgroup: the geometry kernels of a frame are short and latency-bound, so a share
// of the tiles rides along in their launches as extra workgroups and fills the otherwise idle chip.
struct SkyShare {
  DevCube sky;        // bordered skybox cube
  uint32_t* rgba8;    // colour target of the row range
  int first, count;   // tiles [first, first + count) are shaded by the workgroups after the kernel's own
  int own_blocks;     // number of workgroups that do the kernel's own work
};

// Every pixel of tile `tile` gets the skybox colour (covered ones are overwritten by the surface pass).
PBR_DEV void sky_tile(const FrameParams* __restrict__ p, const DevCube& sky, uint32_t* __restrict__ rgba8, int tile) {
  const int tile_x = tile % p->tiles_x, tile_y = tile / p->tiles_x;
  const int x = tile_x * PBR_TILE + (threadIdx.x & 7) * 4, y = p->y0 + tile_y * PBR_TILE + (threadIdx.x >> 3);
  if (x >= p->W || y >= p->y1) return;
  uint32_t c[4];
  sky_pixels4(p, sky, x, y, c);
  const size_t o = (size_t)(y - p->y0) * p->W + x;
  if (x + 3 < p->W && (o & 3) == 0) { *reinterpret_cast<uint4*>(rgba8 + o) = make_uint4(c[0], c[1], c[2], c[3]); return; }
#pragma unroll
  for (int k = 0; k < 4; ++k) if (x + k < p->W) rgba8[o + k] = c[k];
}

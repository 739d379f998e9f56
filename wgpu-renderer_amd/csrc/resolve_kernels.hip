// gfx950 deferred shading resolve: produces the colour of the covered pixels from the rasteriser's
// visibility data (the quad work lists, or the triangle-id buffer for the general kernel) with the
// arithmetic of fs_main in src/shaders/shader.wgsl:93-160 (normal mapping from screen-space
// derivatives, four point lights with Cook-Torrance, split-sum IBL, gamma).  Background pixels get
// fs_main of src/shaders/skybox.wgsl:24-33 in the geometry launches (sky_shade.h).
//
// A hardware rasteriser shades 2x2 quads so that dpdx/dpdy and implicit texture LODs exist.  Here
// every covered pixel re-derives its quad on its own triangle: the varyings are evaluated at the
// quad's (0,0), (1,0) and (0,1) pixel centres and differenced (coarse derivatives, shared by the
// quad).  The prefilter cube lookup and the BRDF-table lookup take coordinates that depend on the
// normal map, so the normal-mapped reflection vector is evaluated at those three pixels as well.
// Four consecutive lanes hold one 2x2 quad; when all four pixels of a quad show the same triangle
// the three helper evaluations are the neighbours' own results and are fetched with cross-lane
// reads instead of being recomputed.
//
// Sampling rules: lambda = log2(max(rho_x, rho_y)) with Euclidean rho; the material sampler has
// mipmapFilter "nearest" (level ceil(lambda+0.5)-1) and picks magFilter when lambda <= 0; cube
// lookups are trilinear with the Vulkan face-selection / derivative transform and seamless edges.
#include "pbr_internal.h"
#include "devmath.h"
#include "sky_shade.h"

namespace {

constexpr float PI_F = 3.14159265359f;   // shader.wgsl:49

struct Lod2D { int level; bool linear; };

PBR_DEV Lod2D lod_2d(int w, int h, int levels, uint32_t sampler, float2 ddx, float2 ddy) {
  const float fw = (float)w, fh = (float)h;
  const float mux = ddx.x * fw, mvx = ddx.y * fh, muy = ddy.x * fw, mvy = ddy.y * fh;
  const float rho2 = fmaxf(mux * mux + mvx * mvx, muy * muy + mvy * mvy);
  float lambda = 0.5f * log2_fast(rho2);   // = log2(rho); -inf when the footprint is zero
  if (isnan(lambda)) lambda = 0.0f;
  lambda = fminf(fmaxf(lambda, 0.0f), 32.0f);
  Lod2D r;
  r.linear = lambda <= 0.0f ? (sampler & 4u) != 0u : (sampler & 8u) != 0u;
  lambda = fminf(lambda, (float)(levels - 1));
  r.level = min(max((int)ceilf(lambda + 0.5f) - 1, 0), levels - 1);
  return r;
}

PBR_DEV int wrap_coord(int i, int n, bool repeat) {
  if (repeat) {
    if ((n & (n - 1)) == 0) return i & (n - 1);
    i %= n;
    return i < 0 ? i + n : i;
  }
  return min(max(i, 0), n - 1);
}

PBR_DEV float4 decode8(uint32_t t, bool srgb, const float* s_srgb) {
  const uint32_t r = t & 255u, g = (t >> 8) & 255u, b = (t >> 16) & 255u, a = t >> 24;
  const float k = 1.0f / 255.0f;
  if (srgb) return make_float4(s_srgb[r], s_srgb[g], s_srgb[b], (float)a * k);
  return make_float4((float)r * k, (float)g * k, (float)b * k, (float)a * k);
}

PBR_DEV float4 bilerp4(float4 t00, float4 t10, float4 t01, float4 t11, float a, float b) {
  const float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
  return make_float4(((t00.x * w00 + t10.x * w10) + t01.x * w01) + t11.x * w11, ((t00.y * w00 + t10.y * w10) + t01.y * w01) + t11.y * w11,
                     ((t00.z * w00 + t10.z * w10) + t01.z * w01) + t11.z * w11, ((t00.w * w00 + t10.w * w10) + t01.w * w01) + t11.w * w11);
}

PBR_DEV float4 nan4() { const float n = __int_as_float(0x7fc00000); return make_float4(n, n, n, n); }

PBR_DEV float4 sample_2d(const DevTex8& t, Lod2D lod, uint32_t sampler, float u, float v, const float* s_srgb) {
  if (isnan(u) || isnan(v)) return nan4();
  const bool rep_u = sampler & 1u, rep_v = sampler & 2u;
  const int lw = max(1, t.w >> lod.level), lh = max(1, t.h >> lod.level);
  const uint32_t* base = t.base + t.off[lod.level];
  if (!lod.linear) {
    const int x = wrap_coord((int)floorf(u * (float)lw), lw, rep_u), y = wrap_coord((int)floorf(v * (float)lh), lh, rep_v);
    return decode8(base[tex8_block_offset(x, y, lw)], t.srgb, s_srgb);
  }
  const float tu = u * (float)lw - 0.5f, tv = v * (float)lh - 0.5f;
  const float fu = floorf(tu), fv = floorf(tv);
  const int x0 = wrap_coord((int)fu, lw, rep_u), x1 = wrap_coord((int)fu + 1, lw, rep_u);
  const int y0 = wrap_coord((int)fv, lh, rep_v), y1 = wrap_coord((int)fv + 1, lh, rep_v);
  return bilerp4(decode8(base[tex8_block_offset(x0, y0, lw)], t.srgb, s_srgb), decode8(base[tex8_block_offset(x1, y0, lw)], t.srgb, s_srgb),
                 decode8(base[tex8_block_offset(x0, y1, lw)], t.srgb, s_srgb), decode8(base[tex8_block_offset(x1, y1, lw)], t.srgb, s_srgb), tu - fu, tv - fv);
}

// rgba16f split-sum table sampled with the material sampler (shader.wgsl:150): one level, so the
// derivatives only choose between magFilter and minFilter.
PBR_DEV float2 sample_lut(const uint16_t* lut, int size, uint32_t sampler, float u, float v, float2 ddx, float2 ddy) {
  if (isnan(u) || isnan(v)) { const float4 n = nan4(); return make_float2(n.x, n.y); }
  const Lod2D lod = lod_2d(size, size, 1, sampler, ddx, ddy);
  const bool rep_u = sampler & 1u, rep_v = sampler & 2u;
  const ushort4* tx = reinterpret_cast<const ushort4*>(lut);
  const float fs = (float)size;
  if (!lod.linear) {
    const int x = wrap_coord((int)floorf(u * fs), size, rep_u), y = wrap_coord((int)floorf(v * fs), size, rep_v);
    const ushort4 t = tx[(uint32_t)y * size + x];
    return make_float2(__half2float(__ushort_as_half(t.x)), __half2float(__ushort_as_half(t.y)));
  }
  const float tu = u * fs - 0.5f, tv = v * fs - 0.5f;
  const float fu = floorf(tu), fv = floorf(tv), a = tu - fu, b = tv - fv;
  const int x0 = wrap_coord((int)fu, size, rep_u), x1 = wrap_coord((int)fu + 1, size, rep_u);
  const int y0 = wrap_coord((int)fv, size, rep_v), y1 = wrap_coord((int)fv + 1, size, rep_v);
  const ushort4 t00 = tx[(uint32_t)y0 * size + x0], t10 = tx[(uint32_t)y0 * size + x1], t01 = tx[(uint32_t)y1 * size + x0], t11 = tx[(uint32_t)y1 * size + x1];
  const float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
  // NaN texels (column 0 of the table) must poison the result even at zero weight: 0 * NaN = NaN.
  const float rx = ((__half2float(__ushort_as_half(t00.x)) * w00 + __half2float(__ushort_as_half(t10.x)) * w10) +
                    __half2float(__ushort_as_half(t01.x)) * w01) + __half2float(__ushort_as_half(t11.x)) * w11;
  const float ry = ((__half2float(__ushort_as_half(t00.y)) * w00 + __half2float(__ushort_as_half(t10.y)) * w10) +
                    __half2float(__ushort_as_half(t01.y)) * w01) + __half2float(__ushort_as_half(t11.y)) * w11;
  return make_float2(rx, ry);
}

// ---------------- cube maps ----------------
PBR_DEV V3 cube_dir(int face, float sc, float tc) {
  switch (face) {
    case 0: return v3(1.0f, -tc, -sc);
    case 1: return v3(-1.0f, -tc, sc);
    case 2: return v3(sc, 1.0f, tc);
    case 3: return v3(sc, -1.0f, -tc);
    case 4: return v3(sc, -tc, 1.0f);
    default: return v3(-sc, -tc, -1.0f);
  }
}

// Tap (i, j) of a face; an index one texel outside follows the cube onto the neighbouring face.
PBR_DEV bool cube_tap(const float4* level, int n, int face, int i, int j, float4& out) {
  const bool ox = i < 0 || i >= n, oy = j < 0 || j >= n;
  if (ox && oy) return false;   // beyond a corner: no such texel
  if (ox || oy) {
    const float sc = ((float)i + 0.5f) / (float)n * 2.0f - 1.0f, tc = ((float)j + 0.5f) / (float)n * 2.0f - 1.0f;
    const CubeCoord c = cube_select(cube_dir(face, sc, tc));
    const float ima = rcp_fast(fabsf(c.rc));
    const float s = 0.5f * (c.sc * ima) + 0.5f, t = 0.5f * (c.tc * ima) + 0.5f;
    face = c.face;
    i = min(max((int)floorf(s * (float)n), 0), n - 1);
    j = min(max((int)floorf(t * (float)n), 0), n - 1);
  }
  out = level[((size_t)face * n + j) * n + i];
  return true;
}

PBR_DEV float4 cube_sample_level(const DevCube& cube, int level, const CubeCoord& c) {
  const int n = max(1, cube.size >> level);
  const float4* base = cube.base + cube.off[level];
  const float ima = rcp_fast(fabsf(c.rc));
  const float s = 0.5f * (c.sc * ima) + 0.5f, t = 0.5f * (c.tc * ima) + 0.5f;
  const float tu = s * (float)n - 0.5f, tv = t * (float)n - 0.5f;
  const float fu = floorf(tu), fv = floorf(tv);
  const float a = tu - fu, b = tv - fv;
  const int i0 = (int)fu, j0 = (int)fv;
  if (i0 >= 0 && j0 >= 0 && i0 + 1 < n && j0 + 1 < n) {   // interior: four taps on one face
    const float4* row0 = base + ((size_t)c.face * n + j0) * n + i0;
    return bilerp4(row0[0], row0[1], row0[n], row0[n + 1], a, b);
  }
  float4 tap[4];
  bool ok[4];
  ok[0] = cube_tap(base, n, c.face, i0, j0, tap[0]);
  ok[1] = cube_tap(base, n, c.face, i0 + 1, j0, tap[1]);
  ok[2] = cube_tap(base, n, c.face, i0, j0 + 1, tap[2]);
  ok[3] = cube_tap(base, n, c.face, i0 + 1, j0 + 1, tap[3]);
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (!ok[k]) {   // at most one tap can be missing: it becomes the mean of the other three
      float4 s3 = make_float4(0, 0, 0, 0);
#pragma unroll
      for (int m = 0; m < 4; ++m) if (m != k) { s3.x += tap[m].x; s3.y += tap[m].y; s3.z += tap[m].z; s3.w += tap[m].w; }
      tap[k] = make_float4(s3.x * (1.0f / 3.0f), s3.y * (1.0f / 3.0f), s3.z * (1.0f / 3.0f), s3.w * (1.0f / 3.0f));
    }
  return bilerp4(tap[0], tap[1], tap[2], tap[3], a, b);
}

PBR_DEV float cube_lambda(int size, V3 r, V3 drdx, V3 drdy) {
  const CubeCoord c = cube_select(r);
  const float ma = fabsf(c.rc), sgn = c.rc >= 0 ? 1.0f : -1.0f, inv_rc2 = rcp_fast(c.rc * c.rc);
  float rho2[2];
  const V3 d[2] = {drdx, drdy};
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    float dsc, dtc, drc;
    cube_face_coords(c.face, d[k], dsc, dtc, drc);
    const float dma = sgn * drc;
    const float ds = 0.5f * ((ma * dsc - c.sc * dma) * inv_rc2), dt = 0.5f * ((ma * dtc - c.tc * dma) * inv_rc2);
    const float mu = ds * (float)size, mv = dt * (float)size;
    rho2[k] = mu * mu + mv * mv;
  }
  return 0.5f * log2_fast(fmaxf(rho2[0], rho2[1]));
}

PBR_DEV float4 cube_sample(const DevCube& cube, V3 r, float lambda) {
  if (any_nan(r)) return nan4();   // a NaN direction selects no face
  const CubeCoord c = cube_select(r);
  if (isnan(lambda)) lambda = 0.0f;
  lambda = fminf(fmaxf(lambda, 0.0f), (float)(cube.levels - 1));
  const int l0 = (int)floorf(lambda), l1 = min(l0 + 1, cube.levels - 1);
  const float f = lambda - (float)l0;
  const float4 a = cube_sample_level(cube, l0, c);
  if (f == 0.0f || l1 == l0) return a;
  const float4 b = cube_sample_level(cube, l1, c);
  return make_float4(a.x * (1.0f - f) + b.x * f, a.y * (1.0f - f) + b.y * f, a.z * (1.0f - f) + b.z * f, a.w * (1.0f - f) + b.w * f);
}

// ---------------- attribute interpolation ----------------
struct Tri {
  float4 c0, c1, c2;       // clip positions
  V3 p0, p1, p2;           // world positions
  V3 n0, n1, n2;           // transformed normals
  float2 t0, t1, t2;       // texcoords
};
struct Varyings { float2 uv; V3 normal, pos; };

PBR_DEV Varyings blend(const Tri& t, float b0, float b1, float b2) {
  Varyings v;
  v.uv = make_float2((t.t0.x * b0 + t.t1.x * b1) + t.t2.x * b2, (t.t0.y * b0 + t.t1.y * b1) + t.t2.y * b2);
  v.normal = t.n0 * b0 + t.n1 * b1 + t.n2 * b2;
  v.pos = t.p0 * b0 + t.p1 * b1 + t.p2 * b2;
  return v;
}

// Screen-space barycentrics of one triangle over one quad.  They come from the snapped vertices
// through integer edge functions at the quad's first pixel centre; the other three centres are one
// pixel step away, added in f32.
struct QuadBary {
  bool projective;
  float l0, l1;                 // barycentrics of vertices 0 and 1 at pixel (0,0) of the quad
  float l0x, l0y, l1x, l1y;     // their change per pixel in x and y
  float iw0, iw1, iw2;          // 1 / w of the three vertices
};

// Value of lane SRC (0..3) of the caller's quad, as a DPP move: no LDS traffic, no wait.
template <int SRC>
PBR_DEV float quad_bcast(float v) {
  // every lane is written (full row and bank masks) and reads a lane of its own quad, so no previous value is needed
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), SRC * 0x55, 0xf, 0xf, true));
}
template <int SRC>
PBR_DEV V3 quad_bcast3(V3 v) { return v3(quad_bcast<SRC>(v.x), quad_bcast<SRC>(v.y), quad_bcast<SRC>(v.z)); }

// The triangle a pixel shows, from the shading record the setup kernel wrote for it (one round of nine 16-byte
// loads): attributes, and the barycentrics of quad (qx, qy) from the snapped vertices the rasteriser used.
// QUAD_SHARED: the four lanes of the quad show the same triangle; its first lane fetches the record and the other
// three take it across lanes, a quarter of the bytes through the vector memory path.
template <bool QUAD_SHARED = false>
PBR_DEV void load_triangle(const TriRecord* __restrict__ records, uint32_t tri_index, int qx, int qy, Tri& t, QuadBary& qb) {
  const float4* rec = reinterpret_cast<const float4*>(records + tri_index);
  float4 r0, r1, r2, r3, r4, r5, r6, r7, r8;
  if constexpr (QUAD_SHARED) {
    // The other three lanes never use their own values (only the first lane's are read below), so these need no
    // particular content - but they must be defined: each register is claimed through an empty asm statement, which
    // costs no instruction and leaves the compiler nothing undefined to reason about.
    float4 v[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) asm volatile("" : "=v"(v[i].x), "=v"(v[i].y), "=v"(v[i].z), "=v"(v[i].w));
    if ((threadIdx.x & 3u) == 0u) {
#pragma unroll
      for (int i = 0; i < 9; ++i) v[i] = rec[i];
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) v[i] = make_float4(quad_bcast<0>(v[i].x), quad_bcast<0>(v[i].y), quad_bcast<0>(v[i].z), quad_bcast<0>(v[i].w));
    r0 = v[0]; r1 = v[1]; r2 = v[2]; r3 = v[3]; r4 = v[4]; r5 = v[5]; r6 = v[6]; r7 = v[7]; r8 = v[8];
  } else {
    r0 = rec[0]; r1 = rec[1]; r2 = rec[2]; r3 = rec[3]; r4 = rec[4]; r5 = rec[5]; r6 = rec[6]; r7 = rec[7]; r8 = rec[8];
  }
  t.p0 = v3(r3.x, r3.y, r3.z); t.p1 = v3(r3.w, r4.x, r4.y); t.p2 = v3(r4.z, r4.w, r5.x);
  t.n0 = v3(r5.y, r5.z, r5.w); t.n1 = v3(r6.x, r6.y, r6.z); t.n2 = v3(r6.w, r7.x, r7.y);
  t.t0 = make_float2(r7.z, r7.w); t.t1 = make_float2(r8.x, r8.y); t.t2 = make_float2(r8.z, r8.w);
  qb.projective = __float_as_int(r1.w) != 0;
  if (qb.projective) {
    const int X0 = __float_as_int(r0.x), Y0 = __float_as_int(r0.y), X1 = __float_as_int(r0.z), Y1 = __float_as_int(r0.w), X2 = __float_as_int(r1.x),
              Y2 = __float_as_int(r1.y);
    const long long px = (long long)qx * 256 + 128, py = (long long)qy * 256 + 128;
    const int a0 = -(Y2 - Y1), b0 = X2 - X1, a1 = -(Y0 - Y2), b1 = X0 - X2;
    const long long e0 = (long long)b0 * (py - Y1) + (long long)a0 * (px - X1);
    const long long e1 = (long long)b1 * (py - Y2) + (long long)a1 * (px - X2);
    const float inv_area = r1.z;
    qb.l0 = (float)e0 * inv_area; qb.l1 = (float)e1 * inv_area;
    qb.l0x = (float)(a0 * 256) * inv_area; qb.l0y = (float)(b0 * 256) * inv_area;
    qb.l1x = (float)(a1 * 256) * inv_area; qb.l1y = (float)(b1 * 256) * inv_area;
    qb.iw0 = r2.x; qb.iw1 = r2.y; qb.iw2 = r2.z;
  } else {
    // a vertex behind the eye or beyond the guard band (rare): the homogeneous form needs the clip positions
    t.c0 = rec[9]; t.c1 = rec[10]; t.c2 = rec[11];
  }
}

// Perspective-correct varyings at pixel (qx + dx, qy + dy), dx, dy in {0, 1}.
PBR_DEV Varyings varyings_at(const Tri& t, const QuadBary& b, int qx, int qy, int dx, int dy, int W, int H) {
  if (b.projective) {
    const float l0 = b.l0 + (dx ? b.l0x : 0.0f) + (dy ? b.l0y : 0.0f);
    const float l1 = b.l1 + (dx ? b.l1x : 0.0f) + (dy ? b.l1y : 0.0f);
    const float l2 = (1.0f - l0) - l1;
    const float q0 = l0 * b.iw0, q1 = l1 * b.iw1, q2 = l2 * b.iw2;
    const float inv = rcp_fast((q0 + q1) + q2);
    return blend(t, q0 * inv, q1 * inv, q2 * inv);
  }
  // a vertex behind the eye or beyond the guard band: 2D homogeneous barycentrics at the pixel centre
  const float xn = ((float)(qx + dx) + 0.5f) / (float)W * 2.0f - 1.0f, yn = 1.0f - ((float)(qy + dy) + 0.5f) / (float)H * 2.0f;
  const float4 c0 = t.c0, c1 = t.c1, c2 = t.c2;
  const float h0 = (c1.y * c2.w - c2.y * c1.w) * xn + (c2.x * c1.w - c1.x * c2.w) * yn + (c1.x * c2.y - c2.x * c1.y);
  const float h1 = (c2.y * c0.w - c0.y * c2.w) * xn + (c0.x * c2.w - c2.x * c0.w) * yn + (c2.x * c0.y - c0.x * c2.y);
  const float h2 = (c0.y * c1.w - c1.y * c0.w) * xn + (c1.x * c0.w - c0.x * c1.w) * yn + (c0.x * c1.y - c1.x * c0.y);
  const float inv = rcp_fast((h0 + h1) + h2);
  return blend(t, h0 * inv, h1 * inv, h2 * inv);
}

// ---------------- shading ----------------
// FAST = the reference's default sampler (repeat / repeat / linear / linear, renderer.ts:1762-1768) on
// five power-of-two textures of one size: no address-mode or filter branches, one bilinear
// footprint per invocation shared by every texture, and no derivatives for the BRDF table because
// magFilter == minFilter.
struct Footprint { uint32_t o00, o10, o01, o11; float w00, w10, w01, w11; bool nan; };

PBR_DEV Footprint footprint_fast(const DevTex8& t, int level, float u, float v) {
  Footprint f;
  f.nan = isnan(u) || isnan(v);
  const int lw = t.w >> level, lh = t.h >> level;   // powers of two, at least 1
  const float tu = u * (float)lw - 0.5f, tv = v * (float)lh - 0.5f;
  const float fu = floorf(tu), fv = floorf(tv), a = tu - fu, b = tv - fv;
  const int x0 = (int)fu & (lw - 1), x1 = ((int)fu + 1) & (lw - 1), y0 = (int)fv & (lh - 1), y1 = ((int)fv + 1) & (lh - 1);
  // blocked layout (see DevTex8): block row and in-block row of each of the two rows, block column and in-block column of each of the two columns
  const uint32_t base = t.off[level], blocks_x = (uint32_t)(lw + 7) >> 3;
  const uint32_t r0 = base + ((uint32_t)y0 >> 2) * blocks_x * 32u + (((uint32_t)y0 & 3u) << 3), r1 = base + ((uint32_t)y1 >> 2) * blocks_x * 32u + (((uint32_t)y1 & 3u) << 3);
  const uint32_t c0 = ((uint32_t)x0 >> 3) * 32u + ((uint32_t)x0 & 7u), c1 = ((uint32_t)x1 >> 3) * 32u + ((uint32_t)x1 & 7u);
  f.o00 = r0 + c0; f.o10 = r0 + c1; f.o01 = r1 + c0; f.o11 = r1 + c1;
  f.w00 = (1.0f - a) * (1.0f - b); f.w10 = a * (1.0f - b); f.w01 = (1.0f - a) * b; f.w11 = a * b;
  return f;
}

PBR_DEV V3 fetch_fast(const DevTex8& t, const Footprint& f, const float* s_srgb) {
  const uint32_t a = t.base[f.o00], b = t.base[f.o10], c = t.base[f.o01], d = t.base[f.o11];
  V3 r;
  if (t.srgb) {
    r.x = ((s_srgb[a & 255u] * f.w00 + s_srgb[b & 255u] * f.w10) + s_srgb[c & 255u] * f.w01) + s_srgb[d & 255u] * f.w11;
    r.y = ((s_srgb[(a >> 8) & 255u] * f.w00 + s_srgb[(b >> 8) & 255u] * f.w10) + s_srgb[(c >> 8) & 255u] * f.w01) + s_srgb[(d >> 8) & 255u] * f.w11;
    r.z = ((s_srgb[(a >> 16) & 255u] * f.w00 + s_srgb[(b >> 16) & 255u] * f.w10) + s_srgb[(c >> 16) & 255u] * f.w01) + s_srgb[(d >> 16) & 255u] * f.w11;
  } else {
    const float k = 1.0f / 255.0f;
    r.x = (((float)(a & 255u) * f.w00 + (float)(b & 255u) * f.w10) + (float)(c & 255u) * f.w01 + (float)(d & 255u) * f.w11) * k;
    r.y = (((float)((a >> 8) & 255u) * f.w00 + (float)((b >> 8) & 255u) * f.w10) + (float)((c >> 8) & 255u) * f.w01 + (float)((d >> 8) & 255u) * f.w11) * k;
    r.z = (((float)((a >> 16) & 255u) * f.w00 + (float)((b >> 16) & 255u) * f.w10) + (float)((c >> 16) & 255u) * f.w01 + (float)((d >> 16) & 255u) * f.w11) * k;
  }
  if (f.nan) { const float n = __int_as_float(0x7fc00000); r = v3(n, n, n); }
  return r;
}

struct QuadShared { float2 duvdx, duvdy; V3 T; Lod2D lod[5]; };
// What one invocation (owned pixel or helper) contributes to the derivative-dependent lookups.
struct Eval { V3 normal, view_dir, R; float ndotv, roughness, metalness; };

// shader.wgsl:96, :101-107, :111, :148 for one invocation.  `with_mr` also samples the
// metallic-roughness map (helpers need it only for the BRDF-table derivatives of the general samplers).
template <bool FAST>
PBR_DEV Eval eval_invocation(const FrameParams* p, const ResolveInputs& in, const QuadShared& q, const Varyings& v, const float* s_srgb, bool with_mr,
                             Footprint* fp_out) {
  const uint32_t sampler = p->u.sampler_index;
  Eval o;
  V3 tn;
  o.roughness = 0.0f; o.metalness = 0.0f;
  if constexpr (FAST) {
    const Footprint fp = footprint_fast(in.tex[PBR_TEX_NORMAL], q.lod[PBR_TEX_NORMAL].level, v.uv.x, v.uv.y);
    const V3 ns = fetch_fast(in.tex[PBR_TEX_NORMAL], fp, s_srgb);
    tn = v3(ns.x * 2.0f - 1.0f, ns.y * 2.0f - 1.0f, ns.z * 2.0f - 1.0f);
    if (with_mr) { const V3 mr = fetch_fast(in.tex[PBR_TEX_METALLIC_ROUGHNESS], fp, s_srgb); o.metalness = mr.z; o.roughness = mr.y; }
    if (fp_out) *fp_out = fp;
  } else {
    const float4 ns = sample_2d(in.tex[PBR_TEX_NORMAL], q.lod[PBR_TEX_NORMAL], sampler, v.uv.x, v.uv.y, s_srgb);
    tn = v3(ns.x * 2.0f - 1.0f, ns.y * 2.0f - 1.0f, ns.z * 2.0f - 1.0f);
    if (with_mr) {
      const float4 mr = sample_2d(in.tex[PBR_TEX_METALLIC_ROUGHNESS], q.lod[PBR_TEX_METALLIC_ROUGHNESS], sampler, v.uv.x, v.uv.y, s_srgb);
      o.metalness = mr.z; o.roughness = mr.y;
    }
  }
  const V3 N = normalize_fast(v.normal);
  const V3 B = -normalize_fast(cross(N, q.T));
  o.normal = normalize_fast(q.T * tn.x + B * tn.y + N * tn.z);
  o.view_dir = normalize_fast(v3(p->u.view_pos[0], p->u.view_pos[1], p->u.view_pos[2]) - v.pos);
  const V3 I = -o.view_dir;
  const V3 R = I - o.normal * (2.0f * dot(o.normal, I));   // reflect(-viewDir, normal)
  o.R = v3(R.x, R.y, -R.z);
  o.ndotv = fmaxf(dot(o.normal, o.view_dir), 0.0f);
  return o;
}

PBR_DEV float smith_ggx(float c, float k) { return c * rcp_fast(c * (1.0f - k) + k); }   // shader.wgsl:83-91

// fs_main after the derivative-dependent part (shader.wgsl:95, :108-159) for one owned pixel.
template <bool FAST>
PBR_DEV uint32_t shade_owned(const FrameParams* p, const ResolveInputs& in, const QuadShared& q, const Varyings& f, const Eval& fc, const Footprint& fp,
                             float lambda, float2 lut_dx, float2 lut_dy, const float* s_srgb) {
  const uint32_t sampler = p->u.sampler_index;
  const V3 normal = fc.normal, view_dir = fc.view_dir;
  const float metalness = fc.metalness, roughness = fc.roughness, ndotv = fc.ndotv;
  V3 albedo, emissive;
  float occlusion;
  if constexpr (FAST) {
    albedo = fetch_fast(in.tex[PBR_TEX_ALBEDO], fp, s_srgb);
    emissive = fetch_fast(in.tex[PBR_TEX_EMISSIVE], fp, s_srgb);
    occlusion = fetch_fast(in.tex[PBR_TEX_OCCLUSION], fp, s_srgb).x;
  } else {
    const float4 alb4 = sample_2d(in.tex[PBR_TEX_ALBEDO], q.lod[PBR_TEX_ALBEDO], sampler, f.uv.x, f.uv.y, s_srgb);
    albedo = v3(alb4.x, alb4.y, alb4.z);
    const float4 em4 = sample_2d(in.tex[PBR_TEX_EMISSIVE], q.lod[PBR_TEX_EMISSIVE], sampler, f.uv.x, f.uv.y, s_srgb);
    emissive = v3(em4.x, em4.y, em4.z);
    occlusion = sample_2d(in.tex[PBR_TEX_OCCLUSION], q.lod[PBR_TEX_OCCLUSION], sampler, f.uv.x, f.uv.y, s_srgb).x;
  }
  emissive = v3(emissive.x * p->u.emissive_factor[0], emissive.y * p->u.emissive_factor[1], emissive.z * p->u.emissive_factor[2]);
  const float om = 1.0f - metalness;
  const V3 f0 = v3(0.04f * om + albedo.x * metalness, 0.04f * om + albedo.y * metalness, 0.04f * om + albedo.z * metalness);

  // four point lights (shader.wgsl:118-139)
  V3 Lo = v3(0, 0, 0);
  const float a = roughness * roughness, a2 = a * a;
  const float rk = roughness + 1.0f, k = (rk * rk) * 0.125f;
  const float g_v = smith_ggx(ndotv, k);
  // kD * albedo / PI of shader.wgsl:131-138 without its light-dependent factor (1 - F): the same for the four lights
  const float inv_pi = 1.0f / PI_F;
  const V3 lambert = v3(om * albedo.x * inv_pi, om * albedo.y * inv_pi, om * albedo.z * inv_pi);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const V3 to_light = v3(p->u.light_pos[i][0], p->u.light_pos[i][1], p->u.light_pos[i][2]) - f.pos;
    const float dist2 = dot(to_light, to_light), inv_dist = rsq_fast(dist2);
    const V3 light_dir = to_light * inv_dist;
    const V3 halfway = normalize_fast(view_dir + light_dir);
    const float attenuation = inv_dist * inv_dist;
    const float ndoth = fmaxf(dot(normal, halfway), 0.0f);
    float denom = ndoth * ndoth * (a2 - 1.0f) + 1.0f;
    denom = PI_F * denom * denom;
    const float D = a2 * rcp_fast(denom);
    const float fcw = pow5(clamp01(1.0f - fmaxf(dot(halfway, view_dir), 0.0f)));
    const V3 F = v3(f0.x + (1.0f - f0.x) * fcw, f0.y + (1.0f - f0.y) * fcw, f0.z + (1.0f - f0.z) * fcw);
    const float ndotl = fmaxf(dot(normal, light_dir), 0.0f);
    const float G = g_v * smith_ggx(ndotl, k);
    const float dg = D * G * rcp_fast(4.0f * ndotv * ndotl + 0.001f);
    const float sc = attenuation * ndotl;
    Lo.x += ((1.0f - F.x) * lambert.x + dg * F.x) * (p->u.light_color[i][0] * sc);
    Lo.y += ((1.0f - F.y) * lambert.y + dg * F.y) * (p->u.light_color[i][1] * sc);
    Lo.z += ((1.0f - F.z) * lambert.z + dg * F.z) * (p->u.light_color[i][2] * sc);
  }

  // image-based lighting (shader.wgsl:141-153)
  const float fcw = pow5(clamp01(1.0f - ndotv));
  const float omr = 1.0f - roughness;
  const V3 kS = v3(f0.x + (fmaxf(omr, f0.x) - f0.x) * fcw, f0.y + (fmaxf(omr, f0.y) - f0.y) * fcw, f0.z + (fmaxf(omr, f0.z) - f0.z) * fcw);
  const V3 kD = v3((1.0f - kS.x) * om, (1.0f - kS.y) * om, (1.0f - kS.z) * om);
  const float4 irr = cube_sample(in.irr, v3(normal.x, normal.y, -normal.z), 0.0f);
  const V3 diffuse = v3(irr.x * albedo.x, irr.y * albedo.y, irr.z * albedo.z);
  const float4 pf = cube_sample(in.pre, fc.R, lambda);
  const float2 brdf = sample_lut(in.lut, in.lut_size, sampler, ndotv, roughness, lut_dx, lut_dy);
  const V3 specular = v3(pf.x * (kS.x * brdf.x + brdf.y), pf.y * (kS.y * brdf.x + brdf.y), pf.z * (kS.z * brdf.x + brdf.y));
  const V3 ambient = v3((kD.x * diffuse.x + specular.x) * occlusion, (kD.y * diffuse.y + specular.y) * occlusion, (kD.z * diffuse.z + specular.z) * occlusion);
  return pack_color(v3(ambient.x + Lo.x + emissive.x, ambient.y + Lo.y + emissive.y, ambient.z + Lo.z + emissive.z));
}

PBR_DEV V3 shfl3(V3 v, int src) { return v3(__shfl(v.x, src), __shfl(v.y, src), __shfl(v.z, src)); }

// fs_main for pixel `sub` (0..3: (0,0), (1,0), (0,1), (1,1)) of quad (qx, qy) on triangle id - 1 (0 = nothing to
// shade).  With ALONE unset, lanes 4q..4q+3 of a wavefront are the four pixels of one quad in that order and
// every lane of the wavefront must make the call: a lane whose three helper pixels show the same triangle
// borrows their results.  Otherwise (and always with ALONE set, where lanes are unrelated work items) the
// lane evaluates its own triangle at the helper pixels itself.
template <bool FAST, bool ALONE>
PBR_DEV uint32_t shade_quad_lane(const FrameParams* p, const ResolveInputs& in, const float* s_srgb, uint32_t id, int qx, int qy, int lane, int sub) {
  // ids of the quad's pixels (0,0), (1,0), (0,1): helper results can be borrowed from a lane that shows the same triangle
  const int qbase = lane & ~3;
  bool need_explicit = id != 0u;
  if constexpr (!ALONE) {
    const uint32_t id0 = __shfl(id, qbase), id1 = __shfl(id, qbase + 1), id2 = __shfl(id, qbase + 2);
    need_explicit = id != 0u && !(id0 == id && id1 == id && id2 == id);
  }

  uint32_t color = 0;
  Varyings f;
  Eval fc;
  QuadShared q;
  Footprint fp;
  V3 R0, R1, R2;
  float2 lut0 = make_float2(0, 0), lut1 = make_float2(0, 0), lut2 = make_float2(0, 0);
  Tri t;
  QuadBary qb;
  if (id != 0u) {
    load_triangle(in.records, id - 1u, qx, qy, t, qb);
    const Varyings h0 = varyings_at(t, qb, qx, qy, 0, 0, p->W, p->H), h1 = varyings_at(t, qb, qx, qy, 1, 0, p->W, p->H),
                   h2 = varyings_at(t, qb, qx, qy, 0, 1, p->W, p->H);
    f = sub == 0 ? h0 : sub == 1 ? h1 : sub == 2 ? h2 : varyings_at(t, qb, qx, qy, 1, 1, p->W, p->H);
    // quad-shared derivatives (shader.wgsl:97-102)
    const V3 q1 = h1.pos - h0.pos, q2 = h2.pos - h0.pos;
    q.duvdx = make_float2(h1.uv.x - h0.uv.x, h1.uv.y - h0.uv.y);
    q.duvdy = make_float2(h2.uv.x - h0.uv.x, h2.uv.y - h0.uv.y);
    q.T = normalize_fast(q1 * q.duvdy.y - q2 * q.duvdx.y);
    const uint32_t sampler = p->u.sampler_index;
    const int use_mips = p->u.use_mipmaps;
    if constexpr (FAST) {
      const Lod2D l = lod_2d(in.tex[0].w, in.tex[0].h, use_mips ? in.tex[0].levels : 1, sampler, q.duvdx, q.duvdy);
#pragma unroll
      for (int i = 0; i < 5; ++i) q.lod[i] = l;
    } else {
#pragma unroll
      for (int i = 0; i < 5; ++i) q.lod[i] = lod_2d(in.tex[i].w, in.tex[i].h, use_mips ? in.tex[i].levels : 1, sampler, q.duvdx, q.duvdy);
    }
    fc = eval_invocation<FAST>(p, in, q, f, s_srgb, true, &fp);
    if (need_explicit) {
      // this pixel's quad shows more than one triangle: evaluate the missing helpers on this pixel's own triangle
      const Eval e0 = eval_invocation<FAST>(p, in, q, h0, s_srgb, !FAST, nullptr);
      const Eval e1 = eval_invocation<FAST>(p, in, q, h1, s_srgb, !FAST, nullptr);
      const Eval e2 = eval_invocation<FAST>(p, in, q, h2, s_srgb, !FAST, nullptr);
      R0 = e0.R; R1 = e1.R; R2 = e2.R;
      lut0 = make_float2(e0.ndotv, e0.roughness); lut1 = make_float2(e1.ndotv, e1.roughness); lut2 = make_float2(e2.ndotv, e2.roughness);
    }
  }
  // borrow the helper results of uniform quads from the neighbouring lanes (all lanes take part in the exchange)
  if constexpr (!ALONE) {
    const V3 s0 = shfl3(fc.R, qbase), s1 = shfl3(fc.R, qbase + 1), s2 = shfl3(fc.R, qbase + 2);
    const float n0 = __shfl(fc.ndotv, qbase), n1 = __shfl(fc.ndotv, qbase + 1), n2 = __shfl(fc.ndotv, qbase + 2);
    const float r0 = __shfl(fc.roughness, qbase), r1 = __shfl(fc.roughness, qbase + 1), r2 = __shfl(fc.roughness, qbase + 2);
    if (!need_explicit) { R0 = s0; R1 = s1; R2 = s2; lut0 = make_float2(n0, r0); lut1 = make_float2(n1, r1); lut2 = make_float2(n2, r2); }
  }
  if (id != 0u) {
    const float lambda = cube_lambda(in.pre.size, fc.R, R1 - R0, R2 - R0);   // implicit LOD of shader.wgsl:149
    color = shade_owned<FAST>(p, in, q, f, fc, fp, lambda, make_float2(lut1.x - lut0.x, lut1.y - lut0.y), make_float2(lut2.x - lut0.x, lut2.y - lut0.y), s_srgb);
  }
  return color;
}

// The resolve of the multisampled pipeline.  vis holds the triangle of each of the pixel's four samples.
// Each distinct triangle of a pixel is shaded once, at the pixel centre, and its 8-bit colour counts once
// per sample it owns; samples without a triangle count the skybox colour the pixel already holds; the
// pixel becomes the average of the four, rounded half to even.
// The triangle of sample 0 is shaded with the pixel's lane in its quad, as in the single-sample kernel.
// The other triangles of a pixel (edges: a minority of the lanes of any wavefront) are queued in LDS as
// (pixel, triangle, weight) items and then shaded with all lanes busy, each item deriving its helper
// pixels itself; the weighted colours meet in a per-pixel LDS accumulator (three 10-bit sums).
template <bool FAST>
__global__ void __launch_bounds__(256) resolve_surface_msaa_kernel(const FrameParams frame_params, ResolveInputs in, const float* __restrict__ srgb_lut,
                                                                   const uint4* __restrict__ vis, const uint32_t* __restrict__ tile_live,
                                                                   uint32_t* __restrict__ rgba8) {
  const FrameParams* __restrict__ p = &frame_params;
  __shared__ float s_srgb[256];
  __shared__ uint2 s_item[768];     // x: pixel (thread) | weight << 8, y: triangle id + 1
  __shared__ uint32_t s_sum[256];   // r | g << 10 | b << 20, each a sum of up to four 8-bit values
  __shared__ uint32_t s_items;
  // the workgroup's 32 x 8 pixels lie in one tile of the rasteriser, which left the tile's entries unwritten if it binned nothing there
  if (tile_live[(blockIdx.y * 8 / PBR_TILE) * p->tiles_x + blockIdx.x] == 0u) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane & 3, quad = lane >> 2;
  const int qx = blockIdx.x * 32 + quad * 2, qy = p->y0 + blockIdx.y * 8 + wave * 2;
  const int x = qx + (sub & 1), y = qy + (sub >> 1);
  const bool live = x < p->W && y < p->y1;
  const size_t o = live ? (size_t)(y - p->y0) * p->W + x : 0;
  const uint4 ids = live ? vis[o] : make_uint4(0u, 0u, 0u, 0u);
  const bool any = (ids.x | ids.y | ids.z | ids.w) != 0u;
  if (threadIdx.x == 0) s_items = 0u;
  s_sum[threadIdx.x] = 0u;
  if (__syncthreads_or(any) == 0) return;
  s_srgb[threadIdx.x] = srgb_lut[threadIdx.x];
  // queue the triangles of samples 1..3 that differ from every earlier sample's
  {
    const bool f1 = ids.y != 0u && ids.y != ids.x, f2 = ids.z != 0u && ids.z != ids.x && ids.z != ids.y,
               f3 = ids.w != 0u && ids.w != ids.x && ids.w != ids.y && ids.w != ids.z;
    const uint32_t n_new = (uint32_t)f1 + (uint32_t)f2 + (uint32_t)f3;
    if (n_new) {
      uint32_t at = atomicAdd(&s_items, n_new);
      if (f1) s_item[at++] = make_uint2(threadIdx.x | ((1u + (uint32_t)(ids.z == ids.y) + (uint32_t)(ids.w == ids.y)) << 8), ids.y);
      if (f2) s_item[at++] = make_uint2(threadIdx.x | ((1u + (uint32_t)(ids.w == ids.z)) << 8), ids.z);
      if (f3) s_item[at++] = make_uint2(threadIdx.x | (1u << 8), ids.w);
    }
  }
  __syncthreads();
  uint32_t sum = 0u, covered = 0u;
  if (__ballot(ids.x != 0u) != 0ull) {
    const uint32_t c = shade_quad_lane<FAST, false>(p, in, s_srgb, ids.x, qx, qy, lane, sub);
    if (ids.x != 0u) {
      covered = 1u + (uint32_t)(ids.y == ids.x) + (uint32_t)(ids.z == ids.x) + (uint32_t)(ids.w == ids.x);
      sum = covered * ((c & 255u) | (((c >> 8) & 255u) << 10) | (((c >> 16) & 255u) << 20));
    }
  }
  const uint32_t n_items = s_items;
  for (uint32_t i = threadIdx.x; i < n_items; i += 256u) {
    const uint2 item = s_item[i];
    const int t = (int)(item.x & 255u), weight = (int)(item.x >> 8);
    const int tl = t & 63, ts = tl & 3;
    const int iqx = blockIdx.x * 32 + (tl >> 2) * 2, iqy = p->y0 + blockIdx.y * 8 + (t >> 6) * 2;
    const uint32_t c = shade_quad_lane<FAST, true>(p, in, s_srgb, item.y, iqx, iqy, lane, ts);
    atomicAdd(&s_sum[t], (uint32_t)weight * ((c & 255u) | (((c >> 8) & 255u) << 10) | (((c >> 16) & 255u) << 20)));
  }
  __syncthreads();
  if (!any) return;
  sum += s_sum[threadIdx.x];
  covered += (uint32_t)(ids.y != 0u && ids.y != ids.x) + (uint32_t)(ids.z != 0u && ids.z != ids.x) + (uint32_t)(ids.w != 0u && ids.w != ids.x);
  if (covered < 4u) {
    const uint32_t sky = rgba8[o];
    sum += (4u - covered) * ((sky & 255u) | (((sky >> 8) & 255u) << 10) | (((sky >> 16) & 255u) << 20));
  }
  auto quarter = [](uint32_t v) { const uint32_t q = v >> 2, rem = v & 3u; return q + (uint32_t)(rem > 2u || (rem == 2u && (q & 1u))); };
  rgba8[o] = quarter(sum & 1023u) | (quarter((sum >> 10) & 1023u) << 8) | (quarter((sum >> 20) & 1023u) << 16) | 0xff000000u;
}

// ---------------------------------------------------------------------------------------------
// The specialised surface kernel.  Same arithmetic as the FAST branch of the general kernel, arranged so
// that a pixel's memory accesses happen in four sets of independent loads — list entry, triangle
// record, all material texels (own pixel and helpers), all IBL texels — instead of one per lookup:
// each set costs a memory latency, so their number sets the time a wavefront needs per item.
// ---------------------------------------------------------------------------------------------
// A colour as red|green in one register pair and blue alone: channel-wise arithmetic then takes two instructions
// instead of three, each channel seeing the operations it would see written out one by one.
struct P3 { f2 rg; float b; };
PBR_DEV P3 p3(V3 v) { return P3{(f2){v.x, v.y}, v.z}; }
PBR_DEV P3 p3(float x, float y, float z) { return P3{(f2){x, y}, z}; }
PBR_DEV P3 p3s(float v) { return P3{(f2){v, v}, v}; }
PBR_DEV P3 operator+(P3 a, P3 b) { return P3{a.rg + b.rg, a.b + b.b}; }
PBR_DEV P3 operator-(P3 a, P3 b) { return P3{a.rg - b.rg, a.b - b.b}; }
PBR_DEV P3 operator*(P3 a, P3 b) { return P3{a.rg * b.rg, a.b * b.b}; }
PBR_DEV P3 operator*(P3 a, float k) { return P3{a.rg * k, a.b * k}; }
PBR_DEV P3 operator+(P3 a, float k) { return P3{a.rg + k, a.b + k}; }
PBR_DEV P3 operator-(float k, P3 a) { return P3{k - a.rg, k - a.b}; }
PBR_DEV P3 max_s(float k, P3 a) { return P3{(f2){fmaxf(k, a.rg.x), fmaxf(k, a.rg.y)}, fmaxf(k, a.b)}; }

// Red and green of the four taps are blended as one two-wide chain (the converted bytes are free to sit in register
// pairs), blue on its own: each channel sees the operations, in the order, of the general kernel's lookup.
template <bool SRGB>
PBR_DEV V3 decode_fast(const uint32_t t[4], const Footprint& f, const float* s_srgb) {
  V3 r;
  if constexpr (SRGB) {
    const f2 rg = (((f2){s_srgb[t[0] & 255u], s_srgb[(t[0] >> 8) & 255u]} * f.w00 + (f2){s_srgb[t[1] & 255u], s_srgb[(t[1] >> 8) & 255u]} * f.w10) +
                   (f2){s_srgb[t[2] & 255u], s_srgb[(t[2] >> 8) & 255u]} * f.w01) + (f2){s_srgb[t[3] & 255u], s_srgb[(t[3] >> 8) & 255u]} * f.w11;
    r.x = rg.x; r.y = rg.y;
    r.z = ((s_srgb[(t[0] >> 16) & 255u] * f.w00 + s_srgb[(t[1] >> 16) & 255u] * f.w10) + s_srgb[(t[2] >> 16) & 255u] * f.w01) + s_srgb[(t[3] >> 16) & 255u] * f.w11;
  } else {
    const float k = 1.0f / 255.0f;
    // green and blue as the pair: the metallic-roughness lookup reads exactly those two
    const f2 gb = ((((f2){(float)((t[0] >> 8) & 255u), (float)((t[0] >> 16) & 255u)} * f.w00 + (f2){(float)((t[1] >> 8) & 255u), (float)((t[1] >> 16) & 255u)} * f.w10) +
                    (f2){(float)((t[2] >> 8) & 255u), (float)((t[2] >> 16) & 255u)} * f.w01) + (f2){(float)((t[3] >> 8) & 255u), (float)((t[3] >> 16) & 255u)} * f.w11) * k;
    r.x = ((((float)(t[0] & 255u) * f.w00 + (float)(t[1] & 255u) * f.w10) + (float)(t[2] & 255u) * f.w01) + (float)(t[3] & 255u) * f.w11) * k;
    r.y = gb.x; r.z = gb.y;
  }
  if (f.nan) { const float n = __int_as_float(0x7fc00000); r = v3(n, n, n); }
  return r;
}

// shader.wgsl:101-105, :111, :148 from an already fetched normal-map sample.
PBR_DEV void tbn_reflect(const FrameParams* p, const V3& T, const Varyings& v, V3 ns, V3& normal, V3& view_dir, V3& R) {
  const V3 tn = v3(ns.x * 2.0f - 1.0f, ns.y * 2.0f - 1.0f, ns.z * 2.0f - 1.0f);
  const V3 N = normalize_fast(v.normal);
  const V3 B = -normalize_fast(cross(N, T));
  normal = normalize_fast(T * tn.x + B * tn.y + N * tn.z);
  view_dir = normalize_fast(v3(p->u.view_pos[0], p->u.view_pos[1], p->u.view_pos[2]) - v.pos);
  const V3 I = -view_dir;
  const V3 r = I - normal * (2.0f * dot(normal, I));
  R = v3(r.x, r.y, -r.z);
}

// Four consecutive lanes shade the 2x2 pixel quad at (qx, qy); `id` is the lane's own visibility entry.
// UNIFORM: the four pixels show one triangle, so the helper invocations the derivatives need are the
// other three lanes and their values are read across lanes.  Otherwise every covered lane evaluates
// its own triangle at the pixels (0,0), (1,0), (0,1) of the quad itself.
template <bool UNIFORM>
PBR_DEV void surface_fast_quad(const FrameParams* __restrict__ p, const ResolveInputs& in, const TriRecord* __restrict__ records, const float* s_srgb,
                               uint32_t* __restrict__ rgba8, int qx, int qy, uint32_t id, bool live) {
  const int sub = threadIdx.x & 3;
  const int x = qx + (sub & 1), y = qy + (sub >> 1);
  const bool cov = live && id != 0u;
  const size_t o = (size_t)(y - p->y0) * p->W + x;

  // first dependent load: the triangle's record (idle lanes read triangle 0; their results are unused)
  const uint32_t tri_id = cov ? id - 1u : 0u;
  Tri t;
  QuadBary qb;
  load_triangle<UNIFORM>(records, tri_id, qx, qy, t, qb);
  Varyings f, h0, h1, h2;
  if constexpr (UNIFORM) {
    f = varyings_at(t, qb, qx, qy, sub & 1, sub >> 1, p->W, p->H);
    h0.pos = quad_bcast3<0>(f.pos); h1.pos = quad_bcast3<1>(f.pos); h2.pos = quad_bcast3<2>(f.pos);
    h0.uv = make_float2(quad_bcast<0>(f.uv.x), quad_bcast<0>(f.uv.y));
    h1.uv = make_float2(quad_bcast<1>(f.uv.x), quad_bcast<1>(f.uv.y));
    h2.uv = make_float2(quad_bcast<2>(f.uv.x), quad_bcast<2>(f.uv.y));
  } else {
    h0 = varyings_at(t, qb, qx, qy, 0, 0, p->W, p->H); h1 = varyings_at(t, qb, qx, qy, 1, 0, p->W, p->H); h2 = varyings_at(t, qb, qx, qy, 0, 1, p->W, p->H);
    f = sub == 0 ? h0 : sub == 1 ? h1 : sub == 2 ? h2 : varyings_at(t, qb, qx, qy, 1, 1, p->W, p->H);
  }
  // quad-shared derivatives (shader.wgsl:97-102)
  const V3 q1 = h1.pos - h0.pos, q2 = h2.pos - h0.pos;
  const float2 duvdx = make_float2(h1.uv.x - h0.uv.x, h1.uv.y - h0.uv.y), duvdy = make_float2(h2.uv.x - h0.uv.x, h2.uv.y - h0.uv.y);
  const V3 T = normalize_fast(q1 * duvdy.y - q2 * duvdx.y);
  const DevTex8& tx0 = in.tex[0];
  const int level = lod_2d(tx0.w, tx0.h, p->u.use_mipmaps ? tx0.levels : 1, 15u, duvdx, duvdy).level;

  // second dependent load: every material texel of this pixel, and the normal-map texels of the helper pixels
  const Footprint fp = footprint_fast(tx0, level, f.uv.x, f.uv.y);
  // (one 16-byte load per tap from the interleaved copy of the five textures)
  uint32_t t_nrm[4], t_orm[4], t_alb[4], t_emi[4], t_h0[4], t_h1[4], t_h2[4];
  {
#ifdef PBR_ABLATE_ATLAS_TAPS
    // measurement build only (wrong colours): no loads, the addresses kept alive
    const uint4 m0 = make_uint4(fp.o00, fp.o10, 0x80808080u, fp.o11), m1 = m0, m2 = m0, m3 = m0;
#else
    const uint4 m0 = in.atlas[fp.o00], m1 = in.atlas[fp.o10], m2 = in.atlas[fp.o01], m3 = in.atlas[fp.o11];
#endif
    t_alb[0] = m0.x; t_alb[1] = m1.x; t_alb[2] = m2.x; t_alb[3] = m3.x;
    t_emi[0] = m0.y; t_emi[1] = m1.y; t_emi[2] = m2.y; t_emi[3] = m3.y;
    t_nrm[0] = m0.z; t_nrm[1] = m1.z; t_nrm[2] = m2.z; t_nrm[3] = m3.z;
    t_orm[0] = m0.w; t_orm[1] = m1.w; t_orm[2] = m2.w; t_orm[3] = m3.w;
  }
  Footprint fp0 = fp, fp1 = fp, fp2 = fp;
  if constexpr (!UNIFORM) {
    fp0 = footprint_fast(tx0, level, h0.uv.x, h0.uv.y); fp1 = footprint_fast(tx0, level, h1.uv.x, h1.uv.y); fp2 = footprint_fast(tx0, level, h2.uv.x, h2.uv.y);
    const uint32_t* nrm = reinterpret_cast<const uint32_t*>(in.atlas) + 2;   // the normal-map texel of atlas entry i is dword 4 i + 2
    auto normals = [&](const Footprint& f, uint32_t t[4]) { t[0] = nrm[4u * f.o00]; t[1] = nrm[4u * f.o10]; t[2] = nrm[4u * f.o01]; t[3] = nrm[4u * f.o11]; };
    normals(fp0, t_h0); normals(fp1, t_h1); normals(fp2, t_h2);
  }

  V3 normal, view_dir, R, R0, R1, R2;
  tbn_reflect(p, T, f, decode_fast<false>(t_nrm, fp, s_srgb), normal, view_dir, R);
  if constexpr (UNIFORM) {
    R0 = quad_bcast3<0>(R); R1 = quad_bcast3<1>(R); R2 = quad_bcast3<2>(R);
  } else {
    V3 n_, v_;
    tbn_reflect(p, T, h0, decode_fast<false>(t_h0, fp0, s_srgb), n_, v_, R0);
    tbn_reflect(p, T, h1, decode_fast<false>(t_h1, fp1, s_srgb), n_, v_, R1);
    tbn_reflect(p, T, h2, decode_fast<false>(t_h2, fp2, s_srgb), n_, v_, R2);
  }
  const float ndotv = fmaxf(dot(normal, view_dir), 0.0f);
  const V3 orm = decode_fast<false>(t_orm, fp, s_srgb);   // occlusion.r, metallicRoughness.g, metallicRoughness.b
  const float metalness = orm.z, roughness = orm.y;

  // third dependent load: irradiance (one level), prefilter (two levels of the implicit LOD), BRDF table
  float lambda = cube_lambda(in.pre.size, R, R1 - R0, R2 - R0);
  if (isnan(lambda)) lambda = 0.0f;
  lambda = fminf(fmaxf(lambda, 0.0f), (float)(in.pre.levels - 1));
  const int l0 = (int)floorf(lambda), l1 = min(l0 + 1, in.pre.levels - 1);
  const float lf = lambda - (float)l0;
  const V3 nd = v3(normal.x, normal.y, -normal.z);
  const bool bad_n = any_nan(nd), bad_r = any_nan(R);
  const CubeCoord cn = cube_select(bad_n ? v3(0, 0, 1) : nd), cr = cube_select(bad_r ? v3(0, 0, 1) : R);
  const CubeFoot fi = cube_foot(in.irr_b, 0, cn), fa = cube_foot(in.pre_b, l0, cr), fb = cube_foot(in.pre_b, l1, cr);
  const int ls = in.lut_size;
  const float lu = ndotv * (float)ls - 0.5f, lv = roughness * (float)ls - 0.5f;
  const float lfu = floorf(lu), lfv = floorf(lv), la = lu - lfu, lb = lv - lfv;
  int lx0 = (int)lfu, ly0 = (int)lfv;
  const bool bad_lut = isnan(lu) || isnan(lv);
  if ((ls & (ls - 1)) == 0) { lx0 &= ls - 1; ly0 &= ls - 1; }
  else { lx0 = ((lx0 % ls) + ls) % ls; ly0 = ((ly0 % ls) + ls) % ls; }
  if (bad_lut) { lx0 = ly0 = 0; }
  Rgb ti[4], ta[4], tb[4];
  cube_load4(in.irr_b.base, fi, ti); cube_load4(in.pre_b.base, fa, ta); cube_load4(in.pre_b.base, fb, tb);
  const uint4 lq = in.lut_quad[(uint32_t)ly0 * ls + lx0];   // the footprint's four texels, wrapped (see ResolveInputs::lut_quad)
  const uint32_t u00 = lq.x, u10 = lq.y, u01 = lq.z, u11 = lq.w;

  // the rest of fs_main (shader.wgsl:95, :108-159)
  const P3 albedo = p3(decode_fast<true>(t_alb, fp, s_srgb));
  const P3 emissive = p3(decode_fast<true>(t_emi, fp, s_srgb)) * p3(p->u.emissive_factor[0], p->u.emissive_factor[1], p->u.emissive_factor[2]);
  const float occlusion = orm.x;
  const float om = 1.0f - metalness;
  const P3 f0c = p3s(0.04f * om) + albedo * metalness;
  const V3 f0 = v3(f0c.rg.x, f0c.rg.y, f0c.b);
  V3 Lo = v3(0, 0, 0);
  const float a = roughness * roughness, a2 = a * a;
  const float rk = roughness + 1.0f, k = (rk * rk) * 0.125f;
  const float g_v = smith_ggx(ndotv, k);
  // kD * albedo / PI of shader.wgsl:131-138 without its light-dependent factor (1 - F): the same for the four lights
  const float inv_pi = 1.0f / PI_F;
  const P3 lambert_c = albedo * om * inv_pi;
  const V3 lambert = v3(lambert_c.rg.x, lambert_c.rg.y, lambert_c.b);
  // The four lights two at a time: element k of every two-wide value belongs to light i + k, and goes through exactly
  // the operations of the one-light loop of the general kernel (shader.wgsl:118-139), so the results are the same bits;
  // the sum over the lights is taken in the shader's order.
  const float a2m1 = a2 - 1.0f, omk = 1.0f - k, four_ndotv = 4.0f * ndotv;
  const P3 omf0c = 1.0f - f0c;
  const V3 omf0 = v3(omf0c.rg.x, omf0c.rg.y, omf0c.b);
#pragma unroll
  for (int i = 0; i < 4; i += 2) {
    const f2 tlx = (f2){p->u.light_pos[i][0], p->u.light_pos[i + 1][0]} - f.pos.x, tly = (f2){p->u.light_pos[i][1], p->u.light_pos[i + 1][1]} - f.pos.y,
             tlz = (f2){p->u.light_pos[i][2], p->u.light_pos[i + 1][2]} - f.pos.z;
    const f2 dist2 = tlx * tlx + tly * tly + tlz * tlz;
    const f2 inv_dist = {rsq_fast(dist2.x), rsq_fast(dist2.y)};
    const f2 ldx = tlx * inv_dist, ldy = tly * inv_dist, ldz = tlz * inv_dist;
    const f2 hx = view_dir.x + ldx, hy = view_dir.y + ldy, hz = view_dir.z + ldz;
    const f2 hh = hx * hx + hy * hy + hz * hz;
    const f2 hr = {rsq_fast(hh.x), rsq_fast(hh.y)};
    const f2 hnx = hx * hr, hny = hy * hr, hnz = hz * hr;
    const f2 nh = normal.x * hnx + normal.y * hny + normal.z * hnz;
    const f2 ndoth = {fmaxf(nh.x, 0.0f), fmaxf(nh.y, 0.0f)};
    f2 denom = ndoth * ndoth * a2m1 + 1.0f;
    denom = PI_F * denom * denom;
    const f2 D = a2 * (f2){rcp_fast(denom.x), rcp_fast(denom.y)};
    const f2 hv = hnx * view_dir.x + hny * view_dir.y + hnz * view_dir.z;
    const f2 omhv = 1.0f - (f2){fmaxf(hv.x, 0.0f), fmaxf(hv.y, 0.0f)};
    const f2 cl = {clamp01(omhv.x), clamp01(omhv.y)};
    const f2 cl2 = cl * cl, fcw2 = cl2 * cl2 * cl;
    const f2 Fx = f0.x + omf0.x * fcw2, Fy = f0.y + omf0.y * fcw2, Fz = f0.z + omf0.z * fcw2;
    const f2 nl = normal.x * ldx + normal.y * ldy + normal.z * ldz;
    const f2 ndotl = {fmaxf(nl.x, 0.0f), fmaxf(nl.y, 0.0f)};
    const f2 gden = ndotl * omk + k;
    const f2 g_l = ndotl * (f2){rcp_fast(gden.x), rcp_fast(gden.y)};
    const f2 sden = four_ndotv * ndotl + 0.001f;
    const f2 dg = D * (g_v * g_l) * (f2){rcp_fast(sden.x), rcp_fast(sden.y)};
    const f2 sc = (inv_dist * inv_dist) * ndotl;
    const f2 tx_ = ((1.0f - Fx) * lambert.x + dg * Fx), ty_ = ((1.0f - Fy) * lambert.y + dg * Fy), tz_ = ((1.0f - Fz) * lambert.z + dg * Fz);
    const f2 cx = (f2){p->u.light_color[i][0], p->u.light_color[i + 1][0]} * sc, cy = (f2){p->u.light_color[i][1], p->u.light_color[i + 1][1]} * sc,
             cz = (f2){p->u.light_color[i][2], p->u.light_color[i + 1][2]} * sc;
    Lo.x += tx_.x * cx.x; Lo.y += ty_.x * cy.x; Lo.z += tz_.x * cz.x;
    Lo.x += tx_.y * cx.y; Lo.y += ty_.y * cy.y; Lo.z += tz_.y * cz.y;
  }
  const float fcw = pow5(clamp01(1.0f - ndotv));
  const float omr = 1.0f - roughness;
  const P3 kS = f0c + (max_s(omr, f0c) - f0c) * fcw;
  const P3 kD = (1.0f - kS) * om;
  const float nanv = __int_as_float(0x7fc00000);
  V3 irr = weigh4(ti, fi.w);
  if (bad_n) irr = v3(nanv, nanv, nanv);
  const P3 pa = p3(weigh4(ta, fa.w)), pb = p3(weigh4(tb, fb.w));
  const P3 pl = pa * (1.0f - lf) + pb * lf;
  const bool one_level = lf == 0.0f || l1 == l0;
  P3 pf = P3{(f2){one_level ? pa.rg.x : pl.rg.x, one_level ? pa.rg.y : pl.rg.y}, one_level ? pa.b : pl.b};
  if (bad_r) pf = p3s(nanv);
  const float w00 = (1.0f - la) * (1.0f - lb), w10 = la * (1.0f - lb), w01 = (1.0f - la) * lb, w11 = la * lb;
  // NaN texels (column 0 of the table) poison the result even at zero weight: 0 * NaN = NaN
  auto ab = [](uint32_t v) { return (f2){__half2float(__ushort_as_half((unsigned short)(v & 0xffffu))), __half2float(__ushort_as_half((unsigned short)(v >> 16)))}; };
  f2 brdf = ((ab(u00) * w00 + ab(u10) * w10) + ab(u01) * w01) + ab(u11) * w11;
  if (bad_lut) brdf = (f2){nanv, nanv};
  const P3 specular = pf * (kS * brdf.x + brdf.y);
  const P3 ambient = (kD * p3(irr) * albedo + specular) * occlusion;
  const P3 color = ambient + p3(Lo) + emissive;
  if (cov) __builtin_nontemporal_store(pack_color(v3(color.rg.x, color.rg.y, color.b)), rgba8 + o);   // streamed past the L2s: see store_colour4
}

// Work list driven: the rasteriser leaves two lists of 2x2 pixel quads that hold geometry, and background
// pixels never start a wavefront here.  An item is sixteen consecutive quads of one list (one
// wavefront, four lanes per quad).  One workgroup per compute unit, as many wavefronts as the unit
// holds.  The lists are in tile order, so a contiguous stretch of a list is a compact piece of the
// screen and of the textures: workgroups are numbered XCD-major (hardware deals workgroup i to XCD
// i mod 8) and each takes one contiguous stretch of either list, which keeps a unit's sixteen waves
// on neighbouring quads (L1) and an XCD's units on one eighth of the frame (its private L2).  The
// waves of a workgroup claim its items one at a time from a counter in LDS, mixed quads (the
// costlier ones) first, so they finish within one item of each other.
// Two shapes of the same kernel.  One 1024-thread workgroup per compute unit (four waves per SIMD: the kernel fits the
// 128 VGPRs that leaves it without spilling) is the faster one with the chip to itself: sixteen waves draw from one
// pool of items.  But such a workgroup only starts on a compute unit with nothing else on it, so with several frames
// in flight the surface kernels take turns at the whole chip; three 256-thread workgroups per compute unit find room
// among the other frames' kernels and leave them a wave slot per SIMD (56.8 -> 55.1 us per 4K frame with four in
// flight, 28.5 -> 31.5 us alone).  The host picks by whether the context shares its scene with forks.
constexpr int kSurfaceThreads = 1024, kSurfaceSmallThreads = 256, kSurfaceSmallPerCu = 3;
constexpr int kXcds = 8;
#ifdef PBR_SURFACE_TRACE
// Diagnostic build only (make TRACE=1): every wavefront of the surface kernel's last launch leaves the 100 MHz
// timestamps of its start and of the end of each of its items; pbr_debug_surface_trace copies them out.
constexpr int kTraceSlots = 16;
__device__ unsigned long long g_surface_trace[1024 * 16][kTraceSlots];
#endif
// The quad lists of one view as the surface kernels walk them.  The rasteriser appends to PBR_LIST_SEGMENTS segments
// of each list; here the segments' quads are cut into items of 16 quads (one wavefront's worth) and the items of all
// segments numbered in one run, segment after segment: per segment (the number of its first item, its quads).
struct SegmentTable { uint2 uniform[PBR_LIST_SEGMENTS], mixed[PBR_LIST_SEGMENTS]; };
static_assert(PBR_LIST_SEGMENTS == 32, "one half wavefront scans a view's segments");
// x + (x of the lane `shift` places down in the same row of 16 lanes, or 0)
template <int SHIFT>
PBR_DEV uint32_t add_row_shr(uint32_t x) { return x + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x110 + SHIFT, 0xf, 0xf, true); }
// Built by 32 consecutive lanes (seg = lane & 31, the wavefront's other half may build another view's) from the
// frame's counters; the lists' totals in items come back in every lane.
PBR_DEV void build_segment_table(SegmentTable* t, const uint32_t* __restrict__ counters, int lane, uint32_t& items_u, uint32_t& items_m) {
  const int seg = lane & 31;
  const uint2 n = *reinterpret_cast<const uint2*>(counters + PBR_SEGMENT_FIRST + PBR_SEGMENT_WORDS * seg);
  const uint32_t own_u = (n.x + 15u) >> 4, own_m = (n.y + 15u) >> 4;
  // running sums over the 32 lanes: within the rows of 16 by four shifted additions, then the upper row takes the lower row's total
  uint32_t run_u = add_row_shr<8>(add_row_shr<4>(add_row_shr<2>(add_row_shr<1>(own_u))));
  uint32_t run_m = add_row_shr<8>(add_row_shr<4>(add_row_shr<2>(add_row_shr<1>(own_m))));
  const bool upper = (lane & 32) != 0;   // (values of lanes that are switched off are read below, and not used)
  auto from_lane = [&](uint32_t x, int l) { return (uint32_t)(upper ? __builtin_amdgcn_readlane((int)x, 32 + l) : __builtin_amdgcn_readlane((int)x, l)); };
  const uint32_t low_u = from_lane(run_u, 15), low_m = from_lane(run_m, 15);
  if (seg >= 16) { run_u += low_u; run_m += low_m; }
  t->uniform[seg] = make_uint2(run_u - own_u, n.x); t->mixed[seg] = make_uint2(run_m - own_m, n.y);
  items_u = from_lane(run_u, 31); items_m = from_lane(run_m, 31);
}
// Where item `item` (the same in every lane) of a list lies: the position in the list of the quad this lane's group of
// four works on, and whether there is one (the last item of a segment may be short).  One LDS read: the lanes hold the
// table's entries, and the segment's own is read from its lane.
PBR_DEV uint32_t locate_quad(const uint2* table, uint32_t item, uint32_t segment_cap, int lane, bool& live) {
  const uint2 e = table[lane & 31];
  const int seg = __popc((uint32_t)__ballot(e.x <= item)) - 1;   // the last segment that starts at or before the item
  const uint32_t first = (uint32_t)__builtin_amdgcn_readlane((int)e.x, seg), n = (uint32_t)__builtin_amdgcn_readlane((int)e.y, seg);
  const uint32_t q = (item - first) * 16u + (uint32_t)(lane >> 2);
  live = q < n;
  return live ? (uint32_t)seg * segment_cap + q : 0u;
}

// What the views of one pass differ in, as kernel arguments (a single frame is a pass of one view).
struct SurfaceViews {
  FrameParams fp[PBR_MAX_VIEWS];
  const TriRecord* records[PBR_MAX_VIEWS];
  QuadLists lists[PBR_MAX_VIEWS];
  uint32_t* rgba8[PBR_MAX_VIEWS];
  const uint32_t* counters[PBR_MAX_VIEWS];
};
static_assert(sizeof(SurfaceViews) + sizeof(ResolveInputs) + 32 <= 4096, "kernel arguments are limited to 4 KB");
// One launch shades the listed quads of all the views of a pass.  A workgroup takes its stretch of the quad lists of
// every view, view after view (of each view the mixed quads first), and its waves claim the items of all of them from
// the one counter: with several views a wave shades more items one after the other, and the launch's fixed costs (its
// first round of loads into cold caches, its tail) are paid once per pass instead of once per frame.  The stretch a
// workgroup takes is the same relative piece of every view, so with neighbouring camera positions it stays on the same
// texels.  Single frames and passes run this same code, so a frame's bytes do not depend on how it was submitted.
template <int THREADS>
__global__ void __launch_bounds__(THREADS) resolve_surface_fast_kernel(const SurfaceViews sv, int n_views, ResolveInputs in,
                                                                       const float* __restrict__ srgb_lut) {
  __shared__ float s_srgb[256];
  __shared__ uint32_t s_next;
  // per view: items of this workgroup in all, its mixed ones, first mixed item, first uniform item, quads in the two lists
  __shared__ uint32_t s_own[PBR_MAX_VIEWS], s_own_mixed[PBR_MAX_VIEWS], s_m_begin[PBR_MAX_VIEWS], s_u_begin[PBR_MAX_VIEWS];
  __shared__ SegmentTable s_segments[PBR_MAX_VIEWS];
  static_assert(THREADS >= 32 * PBR_MAX_VIEWS, "32 threads per view build its segment table");
  for (int i = threadIdx.x; i < 256; i += THREADS) s_srgb[i] = srgb_lut[i];
  if (threadIdx.x == 0) s_next = 0u;
  if ((int)threadIdx.x < 32 * n_views) {
    const int view = threadIdx.x >> 5;
    uint32_t items_uniform, items_mixed;
    build_segment_table(&s_segments[view], sv.counters[view], threadIdx.x & 63, items_uniform, items_mixed);
    if ((threadIdx.x & 31) == 0) {
      // gridDim.x is a multiple of kXcds; v runs over the workgroups XCD by XCD
      const uint32_t per_xcd = gridDim.x / kXcds, v = (blockIdx.x % kXcds) * per_xcd + blockIdx.x / kXcds, nv = gridDim.x;
      const uint32_t m_begin = (uint32_t)((unsigned long long)items_mixed * v / nv), m_end = (uint32_t)((unsigned long long)items_mixed * (v + 1) / nv);
      const uint32_t u_begin = (uint32_t)((unsigned long long)items_uniform * v / nv), u_end = (uint32_t)((unsigned long long)items_uniform * (v + 1) / nv);
      s_own_mixed[view] = m_end - m_begin; s_own[view] = (m_end - m_begin) + (u_end - u_begin);
      s_m_begin[view] = m_begin; s_u_begin[view] = u_begin;
    }
  } else if ((int)threadIdx.x < 32 * PBR_MAX_VIEWS && (threadIdx.x & 31) == 0) {
    s_own[threadIdx.x >> 5] = 0u;
  }
  __syncthreads();
  // every wave keeps the views' figures in scalar registers (read once from LDS) and picks its item's by comparisons
  auto uniform_u32 = [](const uint32_t* lds) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)*lds); };
  struct PerView { uint32_t own, own_mixed, m_begin, u_begin; };
  PerView pv[PBR_MAX_VIEWS];
#pragma unroll
  for (int v = 0; v < PBR_MAX_VIEWS; ++v) {
    pv[v] = PerView{0u, 0u, 0u, 0u};
    if (v < n_views)   // (a single frame reads one view's four figures, not four views')
      pv[v] = PerView{uniform_u32(&s_own[v]), uniform_u32(&s_own_mixed[v]), uniform_u32(&s_m_begin[v]), uniform_u32(&s_u_begin[v])};
  }
  const uint32_t own0 = pv[0].own, own1 = pv[1].own, own2 = pv[2].own, own3 = pv[3].own;
  const uint32_t own_all = own0 + own1 + own2 + own3;
  const int lane = threadIdx.x & 63;
#ifdef PBR_SURFACE_TRACE
  unsigned long long* trace = g_surface_trace[(blockIdx.x * (THREADS / 64) + (threadIdx.x >> 6)) % (1024 * 16)];
  int trace_n = 2;
  if (lane == 0) { trace[0] = wall_clock64(); trace[1] = ((unsigned long long)blockIdx.x << 32) | (pv[0].own_mixed << 16) | own_all; }
#endif
  for (;;) {
    uint32_t k = 0;
    if (lane == 0) k = atomicAdd(&s_next, 1u);
    k = (uint32_t)__builtin_amdgcn_readfirstlane((int)k);
    if (k >= own_all) break;
    int view = 0;
    if (k >= own0) { k -= own0; view = 1; if (k >= own1) { k -= own1; view = 2; if (k >= own2) { k -= own2; view = 3; } } }
    auto pick = [&](uint32_t a, uint32_t b, uint32_t c, uint32_t d) { return view == 0 ? a : view == 1 ? b : view == 2 ? c : d; };
    const FrameParams* __restrict__ p = &sv.fp[view];   // parameters travel as kernel arguments: no per-frame copy to device memory
    const TriRecord* __restrict__ records = sv.records[view];   // (in.records is not used on this path)
    uint32_t* __restrict__ rgba8 = sv.rgba8[view];
    const uint32_t own_mixed = pick(pv[0].own_mixed, pv[1].own_mixed, pv[2].own_mixed, pv[3].own_mixed);
    const bool uniform_item = k >= own_mixed;
    const SegmentTable& segments = s_segments[view];
    const uint32_t segment_cap = quad_segment_cap((uint32_t)(p->tiles_x * p->tiles_y));
    bool live;
    if (uniform_item) {
      const uint32_t q = locate_quad(segments.uniform, pick(pv[0].u_begin, pv[1].u_begin, pv[2].u_begin, pv[3].u_begin) + k - own_mixed, segment_cap, lane, live);
      const uint2 e = sv.lists[view].uniform[q];
      surface_fast_quad<true>(p, in, records, s_srgb, rgba8, (int)(e.x & 0xffffu) * 2, p->y0 + (int)(e.x >> 16) * 2, e.y, live);
    } else {
      const uint32_t q = locate_quad(segments.mixed, pick(pv[0].m_begin, pv[1].m_begin, pv[2].m_begin, pv[3].m_begin) + k, segment_cap, lane, live);
      const uint32_t pos = sv.lists[view].mixed_pos[q];
      const uint32_t id = reinterpret_cast<const uint32_t*>(sv.lists[view].mixed_ids)[q * 4u + (uint32_t)(lane & 3)];
      surface_fast_quad<false>(p, in, records, s_srgb, rgba8, (int)(pos & 0xffffu) * 2, p->y0 + (int)(pos >> 16) * 2, id, live);
    }
#ifdef PBR_SURFACE_TRACE
    if (lane == 0 && trace_n < kTraceSlots) trace[trace_n] = (wall_clock64() << 1) | (uniform_item ? 1ull : 0ull);
    ++trace_n;
#endif
  }
#ifdef PBR_SURFACE_TRACE
  if (lane == 0) for (int i = trace_n; i < kTraceSlots; ++i) trace[i] = 0ull;
#endif
}

// The same work lists for every other sampler: the general per-lane shading (address modes, nearest / linear
// filters and mipmap use decided per lookup, transformed vertices read from memory) on the listed quads,
// so background pixels start no wavefront here either and the visibility buffer is not read.
constexpr int kGeneralThreads = 768;
__global__ void __launch_bounds__(kGeneralThreads) resolve_surface_lists_kernel(const FrameParams frame_params, ResolveInputs in,
                                                                                const float* __restrict__ srgb_lut, uint32_t* __restrict__ rgba8,
                                                                                QuadLists lists, const uint32_t* __restrict__ counters) {
  const FrameParams* __restrict__ p = &frame_params;
  __shared__ float s_srgb[256];
  __shared__ uint32_t s_next;
  __shared__ SegmentTable s_segments;
  __shared__ uint32_t s_items[2];
  if (threadIdx.x < 256) s_srgb[threadIdx.x] = srgb_lut[threadIdx.x];
  if (threadIdx.x == 0) s_next = 0u;
  if (threadIdx.x < 32) {
    uint32_t iu, im;
    build_segment_table(&s_segments, counters, threadIdx.x, iu, im);
    if (threadIdx.x == 0) { s_items[0] = iu; s_items[1] = im; }
  }
  __syncthreads();
  const uint32_t items_uniform = s_items[0], items_mixed = s_items[1];
  const uint32_t segment_cap = quad_segment_cap((uint32_t)(p->tiles_x * p->tiles_y));
  const uint32_t per_xcd = gridDim.x / kXcds, v = (blockIdx.x % kXcds) * per_xcd + blockIdx.x / kXcds, nv = gridDim.x;
  const uint32_t m_begin = (uint32_t)((unsigned long long)items_mixed * v / nv), m_end = (uint32_t)((unsigned long long)items_mixed * (v + 1) / nv);
  const uint32_t u_begin = (uint32_t)((unsigned long long)items_uniform * v / nv), u_end = (uint32_t)((unsigned long long)items_uniform * (v + 1) / nv);
  const uint32_t own_mixed = m_end - m_begin, own = own_mixed + (u_end - u_begin);
  const int lane = threadIdx.x & 63, sub = lane & 3;
  for (;;) {
    uint32_t k = 0;
    if (lane == 0) k = atomicAdd(&s_next, 1u);
    k = (uint32_t)__builtin_amdgcn_readfirstlane((int)k);
    if (k >= own) break;
    uint32_t pos, id;
    bool live;
    if (k >= own_mixed) {
      const uint2 e = lists.uniform[locate_quad(s_segments.uniform, u_begin + k - own_mixed, segment_cap, lane, live)];
      pos = e.x; id = e.y;
    } else {
      const uint32_t q = locate_quad(s_segments.mixed, m_begin + k, segment_cap, lane, live);
      pos = lists.mixed_pos[q];
      id = reinterpret_cast<const uint32_t*>(lists.mixed_ids)[q * 4u + (uint32_t)sub];
    }
    if (!live) id = 0u;
    const int qx = (int)(pos & 0xffffu) * 2, qy = p->y0 + (int)(pos >> 16) * 2;
    const uint32_t color = shade_quad_lane<false, false>(p, in, s_srgb, id, qx, qy, lane, sub);
    if (id != 0u) rgba8[(size_t)(qy + (sub >> 1) - p->y0) * p->W + qx + (sub & 1)] = color;
  }
}

// Builds one level of a bordered cube (see cube_foot): [6][n + 2][n + 2] from [6][n][n].
__global__ void __launch_bounds__(256) cube_border_kernel(DevCube src, int level, float* __restrict__ dst) {
  const int n = max(1, src.size >> level), m = n + 2;
  const uint32_t idx = blockIdx.x * 256u + threadIdx.x;
  if (idx >= 6u * m * m) return;
  const int face = (int)(idx / (uint32_t)(m * m)), r = (int)(idx % (uint32_t)(m * m));
  const int j = r / m - 1, i = r % m - 1;
  const float4* lv = src.base + src.off[level];
  float4 out;
  if (!cube_tap(lv, n, face, i, j, out)) {
    // beyond a corner there is no texel: a lookup there blends the three that exist
    const int ci = min(max(i, 0), n - 1), cj = min(max(j, 0), n - 1);
    float4 a, b, c;
    cube_tap(lv, n, face, ci, cj, a); cube_tap(lv, n, face, i, cj, b); cube_tap(lv, n, face, ci, j, c);
    out = make_float4(((a.x + b.x) + c.x) * (1.0f / 3.0f), ((a.y + b.y) + c.y) * (1.0f / 3.0f), ((a.z + b.z) + c.z) * (1.0f / 3.0f),
                      ((a.w + b.w) + c.w) * (1.0f / 3.0f));
  }
  dst[3 * (size_t)idx] = out.x; dst[3 * (size_t)idx + 1] = out.y; dst[3 * (size_t)idx + 2] = out.z;
}

// DevCubeRgb::pairs: level 0 of a bordered cube in halves, every entry the texel and its right-hand neighbour.  Finite
// values beyond the largest half (65504, far above anything the 8-bit target distinguishes from white) are clamped to
// it; infinities and NaNs stay what they are.
__global__ void __launch_bounds__(256) cube_pairs_kernel(const float* __restrict__ rgb, uint4* __restrict__ pairs, int m) {
  const uint32_t idx = blockIdx.x * 256u + threadIdx.x;
  if (idx >= 6u * m * m) return;
  const bool last = idx % (uint32_t)m == (uint32_t)(m - 1);      // the last column starts no footprint; it repeats itself
  const float* a = rgb + 3 * (size_t)idx;
  const float* b = last ? a : a + 3;
  auto half_bits = [](float v) -> uint32_t {
    const float lim = 65504.0f;
    if (fabsf(v) <= 3.0e38f) v = fminf(fmaxf(v, -lim), lim);
    const _Float16 h = (_Float16)v;      // round to nearest even
    uint16_t bits; __builtin_memcpy(&bits, &h, 2);
    return bits;
  };
  pairs[idx] = make_uint4(half_bits(a[0]) | half_bits(b[0]) << 16, half_bits(a[1]) | half_bits(b[1]) << 16, half_bits(a[2]) | half_bits(b[2]) << 16, 0u);
}

// The two channels fs_main reads from the rgba16f BRDF table (shader.wgsl:150-151), packed for the resolve: per texel
// the whole bilinear footprint that starts there (see ResolveInputs::lut_quad).
__global__ void __launch_bounds__(256) lut_quad_kernel(const uint16_t* __restrict__ lut, uint4* __restrict__ quad, int size) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= size * size) return;
  const int x = i % size, y = i / size, x1 = (x + 1) % size, y1 = (y + 1) % size;
  auto pair = [&](int xx, int yy) { const size_t t = (size_t)yy * size + xx; return (uint32_t)lut[4 * t] | ((uint32_t)lut[4 * t + 1] << 16); };
  quad[i] = make_uint4(pair(x, y), pair(x1, y), pair(x, y1), pair(x1, y1));
}

__global__ void __launch_bounds__(256) material_atlas_kernel(const uint32_t* __restrict__ albedo, const uint32_t* __restrict__ emissive,
                                                             const uint32_t* __restrict__ normal, const uint32_t* __restrict__ mr,
                                                             const uint32_t* __restrict__ occlusion, uint4* __restrict__ atlas, size_t texels) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= texels) return;
  const uint32_t m = mr[i], o = occlusion[i];
  atlas[i] = make_uint4(albedo[i], emissive[i], normal[i], (o & 255u) | (((m >> 8) & 0xffffu) << 8));
}

// Number of covered pixels, for the frame statistics only: one atomic per 16384 pixels (a single
// counter sustains under a hundred atomics per microsecond, so it must not sit in the resolve).
__global__ void __launch_bounds__(256) count_covered_kernel(const uint32_t* __restrict__ vis, size_t n, int samples, uint32_t* __restrict__ covered) {
  __shared__ uint32_t s_part[4];
  uint32_t c = 0;
  const size_t begin = (size_t)blockIdx.x * 16384;
  for (size_t i = begin + threadIdx.x; i < begin + 16384 && i < n; i += 256) {
    if (samples == 4) { const uint4 v = reinterpret_cast<const uint4*>(vis)[i]; c += (v.x | v.y | v.z | v.w) != 0u; }
    else c += vis[i] != 0u;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) { const uint32_t t = s_part[0] + s_part[1] + s_part[2] + s_part[3]; if (t) atomicAdd(covered, t); }
}

}  // namespace

static int general_blocks_per_cu() {
  static int blocks = 0;
  if (blocks == 0 && (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, resolve_surface_lists_kernel, kGeneralThreads, 0) != hipSuccess || blocks < 1)) blocks = 1;
  return blocks;
}

int surface_fast_blocks_per_cu() {
  int blocks = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, resolve_surface_fast_kernel<kSurfaceThreads>, kSurfaceThreads, 0) != hipSuccess || blocks < 1) blocks = 1;
  return blocks;
}

void launch_surface(hipStream_t s, const FrameParams& p, const ResolveInputs& in, const float* srgb_lut, const uint32_t* vis, uint8_t* rgba8,
                    const QuadLists& lists, const uint32_t* counters, int W, int rows, int tiles_x, int tiles_y, bool fast, int resident_blocks, bool shared_chip) {
  uint32_t* out = reinterpret_cast<uint32_t*>(rgba8);
  if (fast) {
    ViewArgs one = {};   // a single frame is a pass of one view
    one.fp = p; one.records = in.records; one.ra.lists = lists; one.ra.rgba8 = out; one.ra.counters = const_cast<uint32_t*>(counters);
    launch_surface_views(s, &one, 1, in, srgb_lut, resident_blocks, shared_chip);
  } else {
    const int per_cu = std::max(1, general_blocks_per_cu());
    const int blocks = std::max(kXcds, (resident_blocks / std::max(1, surface_fast_blocks_per_cu())) * per_cu / kXcds * kXcds);
    hipLaunchKernelGGL(resolve_surface_lists_kernel, dim3(blocks), dim3(kGeneralThreads), 0, s, p, in, srgb_lut, out, lists, counters);
  }
}

// A pass of several views gets the large shape even when other passes are in flight: its launch is long enough to fill
// the chip by itself (4 views: 86 us against 100 us in the small shape, measured alone), and with whole passes in
// flight the frame time no longer depends on the shape (50.4 against 50.3 us per 4K frame, three passes of four views).
#ifndef PBR_LARGE_SHAPE_FROM_VIEWS
#define PBR_LARGE_SHAPE_FROM_VIEWS 2
#endif
void launch_surface_views(hipStream_t s, const ViewArgs* views, int n_views, const ResolveInputs& in, const float* srgb_lut, int resident_blocks,
                          bool shared_chip) {
  SurfaceViews sv = {};
  for (int v = 0; v < n_views; ++v) {
    sv.fp[v] = views[v].fp; sv.records[v] = views[v].records; sv.lists[v] = views[v].ra.lists; sv.rgba8[v] = views[v].ra.rgba8; sv.counters[v] = views[v].ra.counters;
  }
  // as many workgroups as the chip holds at once, a multiple of the XCD count
  const int blocks = std::max(kXcds, resident_blocks / kXcds * kXcds);
  if (shared_chip && n_views < PBR_LARGE_SHAPE_FROM_VIEWS) {
    const int small = std::max(kXcds, resident_blocks / std::max(1, surface_fast_blocks_per_cu()) * kSurfaceSmallPerCu / kXcds * kXcds);
    hipLaunchKernelGGL(resolve_surface_fast_kernel<kSurfaceSmallThreads>, dim3(small), dim3(kSurfaceSmallThreads), 0, s, sv, n_views, in, srgb_lut);
  } else {
    hipLaunchKernelGGL(resolve_surface_fast_kernel<kSurfaceThreads>, dim3(blocks), dim3(kSurfaceThreads), 0, s, sv, n_views, in, srgb_lut);
  }
}

void launch_surface_msaa(hipStream_t s, const FrameParams& p, const ResolveInputs& in, const float* srgb_lut, const uint32_t* vis, const uint32_t* tile_live,
                         uint8_t* rgba8, int W, int rows, bool fast) {
  static_assert(PBR_TILE == 32, "a 32 x 8 pixel workgroup must lie inside one tile");
  dim3 grid((W + 31) / 32, (rows + 7) / 8);
  const uint4* vis4 = reinterpret_cast<const uint4*>(vis);
  uint32_t* out = reinterpret_cast<uint32_t*>(rgba8);
  if (fast) hipLaunchKernelGGL(resolve_surface_msaa_kernel<true>, grid, dim3(256), 0, s, p, in, srgb_lut, vis4, tile_live, out);
  else hipLaunchKernelGGL(resolve_surface_msaa_kernel<false>, grid, dim3(256), 0, s, p, in, srgb_lut, vis4, tile_live, out);
}

void launch_lut_quad(hipStream_t s, const uint16_t* lut_rgba16f, uint4* quad, int size) {
  hipLaunchKernelGGL(lut_quad_kernel, dim3((unsigned)(((size_t)size * size + 255) / 256)), dim3(256), 0, s, lut_rgba16f, quad, size);
}

void launch_material_atlas(hipStream_t s, const uint32_t* const chains[5], uint4* atlas, size_t texels) {
  hipLaunchKernelGGL(material_atlas_kernel, dim3((unsigned)((texels + 255) / 256)), dim3(256), 0, s, chains[PBR_TEX_ALBEDO], chains[PBR_TEX_EMISSIVE],
                     chains[PBR_TEX_NORMAL], chains[PBR_TEX_METALLIC_ROUGHNESS], chains[PBR_TEX_OCCLUSION], atlas, texels);
}

void launch_cube_border(hipStream_t s, DevCube src, int level, float* dst_level) {
  const int n = std::max(1, src.size >> level), m = n + 2;
  hipLaunchKernelGGL(cube_border_kernel, dim3((unsigned)((6u * m * m + 255u) / 256u)), dim3(256), 0, s, src, level, dst_level);
}

void launch_cube_pairs(hipStream_t s, const float* bordered_level0, uint4* pairs, int size) {
  const int m = std::max(1, size) + 2;
  hipLaunchKernelGGL(cube_pairs_kernel, dim3((unsigned)((6u * m * m + 255u) / 256u)), dim3(256), 0, s, bordered_level0, pairs, m);
}

void launch_count_covered(hipStream_t s, const uint32_t* vis, size_t n, int samples, uint32_t* covered) {
  hipLaunchKernelGGL(count_covered_kernel, dim3((unsigned)((n + 16383) / 16384)), dim3(256), 0, s, vis, n, samples, covered);
}

#ifdef PBR_SURFACE_TRACE
extern "C" int pbr_debug_surface_trace(unsigned long long* out, size_t bytes) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_surface_trace), bytes < sizeof(g_surface_trace) ? bytes : sizeof(g_surface_trace));
}
#endif

// gfx950 deferred shading resolve: reads the triangle-id visibility buffer and produces the
// frame's colour.  Covered pixels run the arithmetic of fs_main in src/shaders/shader.wgsl:93-160
// (normal mapping from screen-space derivatives, four point lights with Cook-Torrance, split-sum
// IBL, gamma); background pixels run fs_main of src/shaders/skybox.wgsl:24-33.
//
// A hardware rasteriser shades 2x2 quads so that dpdx/dpdy and implicit texture LODs exist.  Here
// every pixel re-derives its quad: the varyings of the pixel's own triangle are evaluated at the
// quad's (0,0), (1,0) and (0,1) pixels and differenced (coarse derivatives, shared by the quad).
// The prefilter cube lookup and the BRDF-table lookup take coordinates that depend on the normal
// map, so those coordinates are evaluated at the three quad pixels as well.
//
// Sampling rules: lambda = log2(max(rho_x, rho_y)) with Euclidean rho; the material sampler has
// mipmapFilter "nearest" (level ceil(lambda+0.5)-1) and picks magFilter when lambda <= 0; cube
// lookups are trilinear with the Vulkan face-selection / derivative transform and seamless edges.
#include "pbr_internal.h"
#include "devmath.h"

namespace {

constexpr float PI_F = 3.14159265359f;   // shader.wgsl:49

struct Lod2D { int level; bool linear; };

PBR_DEV Lod2D lod_2d(int w, int h, int levels, uint32_t sampler, float2 ddx, float2 ddy) {
  const float fw = (float)w, fh = (float)h;
  const float mux = ddx.x * fw, mvx = ddx.y * fh, muy = ddy.x * fw, mvy = ddy.y * fh;
  const float rho2 = fmaxf(mux * mux + mvx * mvx, muy * muy + mvy * mvy);
  float lambda = 0.5f * log2f(rho2);     // = log2(rho); -inf when the footprint is zero
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
  if (srgb) return make_float4(s_srgb[r], s_srgb[g], s_srgb[b], (float)a / 255.0f);
  return make_float4((float)r / 255.0f, (float)g / 255.0f, (float)b / 255.0f, (float)a / 255.0f);
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
    return decode8(base[(uint32_t)y * lw + x], t.srgb, s_srgb);
  }
  const float tu = u * (float)lw - 0.5f, tv = v * (float)lh - 0.5f;
  const float fu = floorf(tu), fv = floorf(tv);
  const int x0 = wrap_coord((int)fu, lw, rep_u), x1 = wrap_coord((int)fu + 1, lw, rep_u);
  const int y0 = wrap_coord((int)fv, lh, rep_v), y1 = wrap_coord((int)fv + 1, lh, rep_v);
  const uint32_t r0 = (uint32_t)y0 * lw, r1 = (uint32_t)y1 * lw;
  return bilerp4(decode8(base[r0 + x0], t.srgb, s_srgb), decode8(base[r0 + x1], t.srgb, s_srgb), decode8(base[r1 + x0], t.srgb, s_srgb),
                 decode8(base[r1 + x1], t.srgb, s_srgb), tu - fu, tv - fv);
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
struct CubeCoord { int face; float sc, tc, rc; };

PBR_DEV CubeCoord cube_select(V3 r) {
  const float ax = fabsf(r.x), ay = fabsf(r.y), az = fabsf(r.z);
  CubeCoord c;
  if (az >= ax && az >= ay) { c.face = r.z >= 0 ? 4 : 5; c.sc = r.z >= 0 ? r.x : -r.x; c.tc = -r.y; c.rc = r.z; }
  else if (ay >= ax)        { c.face = r.y >= 0 ? 2 : 3; c.sc = r.x; c.tc = r.y >= 0 ? r.z : -r.z; c.rc = r.y; }
  else                      { c.face = r.x >= 0 ? 0 : 1; c.sc = r.x >= 0 ? -r.z : r.z; c.tc = -r.y; c.rc = r.x; }
  return c;
}
PBR_DEV void cube_table(int face, V3 d, float& sc, float& tc, float& rc) {
  switch (face) {
    case 0: sc = -d.z; tc = -d.y; rc = d.x; break;
    case 1: sc = d.z;  tc = -d.y; rc = d.x; break;
    case 2: sc = d.x;  tc = d.z;  rc = d.y; break;
    case 3: sc = d.x;  tc = -d.z; rc = d.y; break;
    case 4: sc = d.x;  tc = -d.y; rc = d.z; break;
    default: sc = -d.x; tc = -d.y; rc = d.z; break;
  }
}
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
    const float ma = fabsf(c.rc);
    const float s = 0.5f * (c.sc / ma) + 0.5f, t = 0.5f * (c.tc / ma) + 0.5f;
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
  const float ma = fabsf(c.rc);
  const float s = 0.5f * (c.sc / ma) + 0.5f, t = 0.5f * (c.tc / ma) + 0.5f;
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
      tap[k] = make_float4(s3.x / 3.0f, s3.y / 3.0f, s3.z / 3.0f, s3.w / 3.0f);
    }
  return bilerp4(tap[0], tap[1], tap[2], tap[3], a, b);
}

PBR_DEV float cube_lambda(int size, V3 r, V3 drdx, V3 drdy) {
  const CubeCoord c = cube_select(r);
  const float ma = fabsf(c.rc), sgn = c.rc >= 0 ? 1.0f : -1.0f, inv_rc2 = 1.0f / (c.rc * c.rc);
  float rho2[2];
  const V3 d[2] = {drdx, drdy};
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    float dsc, dtc, drc;
    cube_table(c.face, d[k], dsc, dtc, drc);
    const float dma = sgn * drc;
    const float ds = 0.5f * ((ma * dsc - c.sc * dma) * inv_rc2), dt = 0.5f * ((ma * dtc - c.tc * dma) * inv_rc2);
    const float mu = ds * (float)size, mv = dt * (float)size;
    rho2[k] = mu * mu + mv * mv;
  }
  return 0.5f * log2f(fmaxf(rho2[0], rho2[1]));
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

// Screen-space barycentrics from the snapped vertices (exact integer edge functions), weighted by 1/w.
struct ProjSetup { long long e0, e1; int a0, b0, a1, b1; float inv_area, iw0, iw1, iw2; };

PBR_DEV Varyings interp_proj(const Tri& t, const ProjSetup& s, int dx, int dy) {
  const long long e0 = s.e0 + (long long)s.a0 * (dx * 256) + (long long)s.b0 * (dy * 256);
  const long long e1 = s.e1 + (long long)s.a1 * (dx * 256) + (long long)s.b1 * (dy * 256);
  const float l0 = (float)e0 * s.inv_area, l1 = (float)e1 * s.inv_area, l2 = (1.0f - l0) - l1;
  const float q0 = l0 * s.iw0, q1 = l1 * s.iw1, q2 = l2 * s.iw2;
  const float inv = 1.0f / ((q0 + q1) + q2);
  return blend(t, q0 * inv, q1 * inv, q2 * inv);
}

// A vertex behind the eye or beyond the guard band: 2D homogeneous barycentrics at the pixel centre.
PBR_DEV Varyings interp_homog(const Tri& t, int x, int y, int W, int H) {
  const float xn = ((float)x + 0.5f) / (float)W * 2.0f - 1.0f, yn = 1.0f - ((float)y + 0.5f) / (float)H * 2.0f;
  const float4 c0 = t.c0, c1 = t.c1, c2 = t.c2;
  const float h0 = (c1.y * c2.w - c2.y * c1.w) * xn + (c2.x * c1.w - c1.x * c2.w) * yn + (c1.x * c2.y - c2.x * c1.y);
  const float h1 = (c2.y * c0.w - c0.y * c2.w) * xn + (c0.x * c2.w - c2.x * c0.w) * yn + (c2.x * c0.y - c0.x * c2.y);
  const float h2 = (c0.y * c1.w - c1.y * c0.w) * xn + (c1.x * c0.w - c0.x * c1.w) * yn + (c0.x * c1.y - c1.x * c0.y);
  const float inv = 1.0f / ((h0 + h1) + h2);
  return blend(t, h0 * inv, h1 * inv, h2 * inv);
}

PBR_DEV void snap_xy(float4 c, int W, int H, int& X, int& Y) {
  const float nx = div_rn(c.x, c.w), ny = div_rn(c.y, c.w);
  X = __float2int_rn(mul_rn(mul_rn(mul_rn(add_rn(nx, 1.0f), 0.5f), (float)W), 256.0f));
  Y = __float2int_rn(mul_rn(mul_rn(mul_rn(sub_rn(1.0f, ny), 0.5f), (float)H), 256.0f));
}

// ---------------- shading ----------------
struct QuadShared { float2 duvdx, duvdy; V3 T; Lod2D lod_normal, lod_mr; };
struct FragCommon { V3 normal, view_dir, R; float ndotv, roughness, metalness; };

// shader.wgsl:96, :101-107, :111, :148 for one invocation (own pixel or a quad neighbour).
PBR_DEV FragCommon eval_common(const FrameParams* p, const ResolveInputs& in, const QuadShared& q, const Varyings& v, const float* s_srgb) {
  const uint32_t sampler = p->u.sampler_index;
  FragCommon o;
  const float4 ns = sample_2d(in.tex[PBR_TEX_NORMAL], q.lod_normal, sampler, v.uv.x, v.uv.y, s_srgb);
  const V3 tn = v3(ns.x * 2.0f - 1.0f, ns.y * 2.0f - 1.0f, ns.z * 2.0f - 1.0f);
  const V3 N = normalize(v.normal);
  const V3 B = -normalize(cross(N, q.T));
  o.normal = normalize(q.T * tn.x + B * tn.y + N * tn.z);
  const float4 mr = sample_2d(in.tex[PBR_TEX_METALLIC_ROUGHNESS], q.lod_mr, sampler, v.uv.x, v.uv.y, s_srgb);
  o.metalness = mr.z; o.roughness = mr.y;
  o.view_dir = normalize(v3(p->u.view_pos[0], p->u.view_pos[1], p->u.view_pos[2]) - v.pos);
  const V3 I = -o.view_dir;
  const V3 R = I - o.normal * (2.0f * dot(o.normal, I));   // reflect(-viewDir, normal)
  o.R = v3(R.x, R.y, -R.z);
  o.ndotv = fmaxf(dot(o.normal, o.view_dir), 0.0f);
  return o;
}

PBR_DEV float smith_ggx(float c, float k) { return c / (c * (1.0f - k) + k); }   // shader.wgsl:83-91

PBR_DEV uint32_t pack_color(V3 color) {
  const float g = 1.0f / 2.2f;   // shader.wgsl:157, skybox.wgsl:30
  return to_unorm8(powf(color.x, g)) | (to_unorm8(powf(color.y, g)) << 8) | (to_unorm8(powf(color.z, g)) << 16) | 0xff000000u;
}

PBR_DEV uint32_t shade_sky(const FrameParams* p, const DevCube& sky, int x, int y) {
  const float* m = p->u.inv_view_proj;
  const float xn = ((float)x + 0.5f) / (float)p->W * 2.0f - 1.0f, yn = 1.0f - ((float)y + 0.5f) / (float)p->H * 2.0f;
  const float tx = ((m[0] * xn + m[4] * yn) + m[8]) + m[12];
  const float ty = ((m[1] * xn + m[5] * yn) + m[9]) + m[13];
  const float tz = ((m[2] * xn + m[6] * yn) + m[10]) + m[14];
  const float tw = ((m[3] * xn + m[7] * yn) + m[11]) + m[15];
  const V3 d = normalize(v3(tx / tw, ty / tw, tz / tw));
  const float4 c = cube_sample(sky, v3(d.x, d.y, -d.z), 0.0f);
  return pack_color(v3(c.x, c.y, c.z));
}

PBR_DEV uint32_t shade_surface(const FrameParams* p, const ResolveInputs& in, uint32_t id, int x, int y, const float* s_srgb) {
  const uint32_t sampler = p->u.sampler_index;
  // ---- fetch the triangle ----
  uint32_t inst = 0, tri = id;
  if (p->n_instances > 1) { inst = id / (uint32_t)p->n_tris; tri = id - inst * (uint32_t)p->n_tris; }
  const VertexOut* vb = in.vout + (size_t)inst * p->n_verts;
  const VertexOut v0 = vb[in.indices[3 * tri]], v1 = vb[in.indices[3 * tri + 1]], v2 = vb[in.indices[3 * tri + 2]];
  Tri t;
  t.c0 = v0.clip; t.c1 = v1.clip; t.c2 = v2.clip;
  t.p0 = v3(v0.world_u.x, v0.world_u.y, v0.world_u.z); t.p1 = v3(v1.world_u.x, v1.world_u.y, v1.world_u.z); t.p2 = v3(v2.world_u.x, v2.world_u.y, v2.world_u.z);
  t.n0 = v3(v0.nrm_v.x, v0.nrm_v.y, v0.nrm_v.z); t.n1 = v3(v1.nrm_v.x, v1.nrm_v.y, v1.nrm_v.z); t.n2 = v3(v2.nrm_v.x, v2.nrm_v.y, v2.nrm_v.z);
  t.t0 = make_float2(v0.world_u.w, v0.nrm_v.w); t.t1 = make_float2(v1.world_u.w, v1.nrm_v.w); t.t2 = make_float2(v2.world_u.w, v2.nrm_v.w);

  // ---- varyings at the quad's three reference pixels and at this pixel ----
  const int qx = x & ~1, qy = y & ~1;
  bool projective = t.c0.w > 0.0f && t.c1.w > 0.0f && t.c2.w > 0.0f && fabsf(t.c0.x) <= PBR_GUARD * t.c0.w && fabsf(t.c0.y) <= PBR_GUARD * t.c0.w &&
                    fabsf(t.c1.x) <= PBR_GUARD * t.c1.w && fabsf(t.c1.y) <= PBR_GUARD * t.c1.w && fabsf(t.c2.x) <= PBR_GUARD * t.c2.w &&
                    fabsf(t.c2.y) <= PBR_GUARD * t.c2.w;
  ProjSetup ps = {};
  if (projective) {
    int X0, Y0, X1, Y1, X2, Y2;
    snap_xy(t.c0, p->W, p->H, X0, Y0); snap_xy(t.c1, p->W, p->H, X1, Y1); snap_xy(t.c2, p->W, p->H, X2, Y2);
    const long long area = (long long)(X1 - X0) * (Y2 - Y0) - (long long)(X2 - X0) * (Y1 - Y0);
    if (area == 0) projective = false;
    else {
      const long long px = (long long)qx * 256 + 128, py = (long long)qy * 256 + 128;
      ps.a0 = -(Y2 - Y1); ps.b0 = X2 - X1; ps.a1 = -(Y0 - Y2); ps.b1 = X0 - X2;
      ps.e0 = (long long)ps.b0 * (py - Y1) + (long long)ps.a0 * (px - X1);
      ps.e1 = (long long)ps.b1 * (py - Y2) + (long long)ps.a1 * (px - X2);
      ps.inv_area = 1.0f / (float)area;
      ps.iw0 = 1.0f / t.c0.w; ps.iw1 = 1.0f / t.c1.w; ps.iw2 = 1.0f / t.c2.w;
    }
  }
  Varyings h0, h1, h2, f;
  if (projective) {
    h0 = interp_proj(t, ps, 0, 0); h1 = interp_proj(t, ps, 1, 0); h2 = interp_proj(t, ps, 0, 1); f = interp_proj(t, ps, x - qx, y - qy);
  } else {
    h0 = interp_homog(t, qx, qy, p->W, p->H); h1 = interp_homog(t, qx + 1, qy, p->W, p->H); h2 = interp_homog(t, qx, qy + 1, p->W, p->H);
    f = interp_homog(t, x, y, p->W, p->H);
  }

  // ---- quad-shared derivatives (shader.wgsl:97-102) ----
  QuadShared q;
  const V3 q1 = h1.pos - h0.pos, q2 = h2.pos - h0.pos;
  q.duvdx = make_float2(h1.uv.x - h0.uv.x, h1.uv.y - h0.uv.y);
  q.duvdy = make_float2(h2.uv.x - h0.uv.x, h2.uv.y - h0.uv.y);
  q.T = normalize(q1 * q.duvdy.y - q2 * q.duvdx.y);
  const int use_mips = p->u.use_mipmaps;
  Lod2D lod[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) lod[i] = lod_2d(in.tex[i].w, in.tex[i].h, use_mips ? in.tex[i].levels : 1, sampler, q.duvdx, q.duvdy);
  q.lod_normal = lod[PBR_TEX_NORMAL]; q.lod_mr = lod[PBR_TEX_METALLIC_ROUGHNESS];

  const FragCommon c0 = eval_common(p, in, q, h0, s_srgb), c1 = eval_common(p, in, q, h1, s_srgb), c2 = eval_common(p, in, q, h2, s_srgb);
  const FragCommon fc = eval_common(p, in, q, f, s_srgb);
  const V3 normal = fc.normal, view_dir = fc.view_dir;
  const float metalness = fc.metalness, roughness = fc.roughness, ndotv = fc.ndotv;

  // ---- material (shader.wgsl:95, :108-109) ----
  const float4 alb4 = sample_2d(in.tex[PBR_TEX_ALBEDO], lod[PBR_TEX_ALBEDO], sampler, f.uv.x, f.uv.y, s_srgb);
  const V3 albedo = v3(alb4.x, alb4.y, alb4.z);
  const float4 em4 = sample_2d(in.tex[PBR_TEX_EMISSIVE], lod[PBR_TEX_EMISSIVE], sampler, f.uv.x, f.uv.y, s_srgb);
  const V3 emissive = v3(em4.x * p->u.emissive_factor[0], em4.y * p->u.emissive_factor[1], em4.z * p->u.emissive_factor[2]);
  const float occlusion = sample_2d(in.tex[PBR_TEX_OCCLUSION], lod[PBR_TEX_OCCLUSION], sampler, f.uv.x, f.uv.y, s_srgb).x;
  const float om = 1.0f - metalness;
  const V3 f0 = v3(0.04f * om + albedo.x * metalness, 0.04f * om + albedo.y * metalness, 0.04f * om + albedo.z * metalness);

  // ---- four point lights (shader.wgsl:118-139) ----
  V3 Lo = v3(0, 0, 0);
  const float a = roughness * roughness, a2 = a * a;
  const float rk = roughness + 1.0f, k = (rk * rk) / 8.0f;
  const float g_v = smith_ggx(ndotv, k);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const V3 to_light = v3(p->u.light_pos[i][0], p->u.light_pos[i][1], p->u.light_pos[i][2]) - f.pos;
    const float dist2 = dot(to_light, to_light), dist = sqrtf(dist2);
    const V3 light_dir = v3(to_light.x / dist, to_light.y / dist, to_light.z / dist);
    const V3 halfway = normalize(view_dir + light_dir);
    const float attenuation = 1.0f / (dist * dist);
    const float ndoth = fmaxf(dot(normal, halfway), 0.0f);
    float denom = ndoth * ndoth * (a2 - 1.0f) + 1.0f;
    denom = PI_F * denom * denom;
    const float D = a2 / denom;
    const float fcw = pow5(clamp01(1.0f - fmaxf(dot(halfway, view_dir), 0.0f)));
    const V3 F = v3(f0.x + (1.0f - f0.x) * fcw, f0.y + (1.0f - f0.y) * fcw, f0.z + (1.0f - f0.z) * fcw);
    const float ndotl = fmaxf(dot(normal, light_dir), 0.0f);
    const float G = g_v * smith_ggx(ndotl, k);
    const float sden = 4.0f * ndotv * ndotl + 0.001f;
    const V3 kD = v3((1.0f - F.x) * om, (1.0f - F.y) * om, (1.0f - F.z) * om);
    const V3 radiance = v3(p->u.light_color[i][0] * attenuation, p->u.light_color[i][1] * attenuation, p->u.light_color[i][2] * attenuation);
    Lo.x += (kD.x * albedo.x / PI_F + D * F.x * G / sden) * radiance.x * ndotl;
    Lo.y += (kD.y * albedo.y / PI_F + D * F.y * G / sden) * radiance.y * ndotl;
    Lo.z += (kD.z * albedo.z / PI_F + D * F.z * G / sden) * radiance.z * ndotl;
  }

  // ---- image-based lighting (shader.wgsl:141-153) ----
  const float fcw = pow5(clamp01(1.0f - ndotv));
  const float omr = 1.0f - roughness;
  const V3 kS = v3(f0.x + (fmaxf(omr, f0.x) - f0.x) * fcw, f0.y + (fmaxf(omr, f0.y) - f0.y) * fcw, f0.z + (fmaxf(omr, f0.z) - f0.z) * fcw);
  const V3 kD = v3((1.0f - kS.x) * om, (1.0f - kS.y) * om, (1.0f - kS.z) * om);
  const float4 irr = cube_sample(in.irr, v3(normal.x, normal.y, -normal.z), 0.0f);
  const V3 diffuse = v3(irr.x * albedo.x, irr.y * albedo.y, irr.z * albedo.z);
  const float lambda = cube_lambda(in.pre.size, fc.R, c1.R - c0.R, c2.R - c0.R);
  const float4 pf = cube_sample(in.pre, fc.R, lambda);
  const float2 brdf = sample_lut(in.lut, in.lut_size, sampler, ndotv, roughness, make_float2(c1.ndotv - c0.ndotv, c1.roughness - c0.roughness),
                                 make_float2(c2.ndotv - c0.ndotv, c2.roughness - c0.roughness));
  const V3 specular = v3(pf.x * (kS.x * brdf.x + brdf.y), pf.y * (kS.y * brdf.x + brdf.y), pf.z * (kS.z * brdf.x + brdf.y));
  const V3 ambient = v3((kD.x * diffuse.x + specular.x) * occlusion, (kD.y * diffuse.y + specular.y) * occlusion, (kD.z * diffuse.z + specular.z) * occlusion);
  return pack_color(v3(ambient.x + Lo.x + emissive.x, ambient.y + Lo.y + emissive.y, ambient.z + Lo.z + emissive.z));
}

__global__ void __launch_bounds__(256) resolve_kernel(const FrameParams* __restrict__ p, ResolveInputs in, const float* __restrict__ srgb_lut,
                                                      const uint32_t* __restrict__ vis, uint32_t* __restrict__ rgba8, uint32_t* __restrict__ covered) {
  __shared__ float s_srgb[256];
  s_srgb[threadIdx.x] = srgb_lut[threadIdx.x];
  __syncthreads();
  const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = p->y0 + blockIdx.y * 8 + (threadIdx.x >> 5);
  const bool live = x < p->W && y < p->y1;
  uint32_t id = 0;
  size_t o = 0;
  if (live) { o = (size_t)(y - p->y0) * p->W + x; id = vis[o]; }
  if (live) rgba8[o] = id ? shade_surface(p, in, id - 1u, x, y, s_srgb) : shade_sky(p, in.sky, x, y);
  const unsigned long long m = __ballot(live && id != 0u);
  if (covered && (threadIdx.x & 63) == 0 && m) atomicAdd(covered, (uint32_t)__popcll(m));
}

}  // namespace

void launch_resolve(hipStream_t s, const FrameParams* p, const ResolveInputs& in, const float* srgb_lut, const uint32_t* vis, uint8_t* rgba8,
                    uint32_t* covered, int W, int rows) {
  dim3 grid((W + 31) / 32, (rows + 7) / 8);
  hipLaunchKernelGGL(resolve_kernel, grid, dim3(256), 0, s, p, in, srgb_lut, vis, reinterpret_cast<uint32_t*>(rgba8), covered);
}

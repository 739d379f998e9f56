// Oracle (test infrastructure): scalar f32 restatement of the reference's draw pass —
// vs_main / fs_main of src/shaders/shader.wgsl, fs_main of src/shaders/skybox.wgsl — under the
// fixed-function state of src/renderer.ts:469-480, :822-859, :910-914, :948-956, :1943-1966.
//
// What the WGSL does not define is taken from the WebGPU / Vulkan rules, each written where it is
// used:
//   * clip volume 0<=z<=w, viewport x_fb=(x_ndc+1)/2*W, y_fb=(1-y_ndc)/2*H, pixel centres at +0.5
//   * vertices snapped to 1/256 pixel, top-left fill rule, front face = counter-clockwise
//   * depth = z_ndc interpolated linearly in screen space, f32, compare "less", first draw wins ties
//   * varyings are perspective-correct
//   * dpdx/dpdy and implicit texture LODs use coarse 2x2-quad differences (pixel (1,0)-(0,0) and
//     (0,1)-(0,0) of the even-aligned quad, shared by its four pixels); helper pixels are
//     evaluated on the plane of the pixel's own triangle
//   * LOD lambda=log2(max(rho_x,rho_y)) with Euclidean rho; mipmapFilter "nearest" picks level
//     ceil(lambda+0.5)-1; lambda<=0 selects magFilter
//   * cube maps: major-axis face selection with z over y over x on ties, the Vulkan derivative
//     transform for the LOD, seamless bilinear taps across face edges, a missing corner tap is the
//     mean of the other three
//   * unorm8 stores clamp to [0,1], round to nearest even, NaN stores 0
#include "oracle.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

namespace {

struct V2 { float x, y; };
struct V3 { float x, y, z; };
struct V4 { float x, y, z, w; };
inline V3 add(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 mul(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 mulv(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline V3 divs(V3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float length(V3 a) { return std::sqrt(dot(a, a)); }
inline V3 normalize(V3 a) { float l = length(a); return {a.x / l, a.y / l, a.z / l}; }
inline V3 neg(V3 a) { return {-a.x, -a.y, -a.z}; }
inline bool has_nan(V3 a) { return std::isnan(a.x) || std::isnan(a.y) || std::isnan(a.z); }
inline float clamp01(float x) { return std::min(std::max(x, 0.0f), 1.0f); }

const float PI = 3.14159265359f;   // shader.wgsl:49

inline void mat_mul(const float* a, const float* b, float* o) {
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r)
      o[c * 4 + r] = ((a[r] * b[c * 4] + a[4 + r] * b[c * 4 + 1]) + a[8 + r] * b[c * 4 + 2]) + a[12 + r] * b[c * 4 + 3];
}
inline V4 mat_vec(const float* m, float x, float y, float z, float w) {
  V4 o;
  o.x = ((m[0] * x + m[4] * y) + m[8] * z) + m[12] * w;
  o.y = ((m[1] * x + m[5] * y) + m[9] * z) + m[13] * w;
  o.z = ((m[2] * x + m[6] * y) + m[10] * z) + m[14] * w;
  o.w = ((m[3] * x + m[7] * y) + m[11] * z) + m[15] * w;
  return o;
}

inline int level_dim(int d, int l) { return std::max(1, d >> l); }
inline int64_t chain_off(int w, int h, int level) {
  int64_t o = 0;
  for (int l = 0; l < level; ++l) o += (int64_t)level_dim(w, l) * level_dim(h, l);
  return o;
}

float g_srgb_lut[256];
bool g_lut_ready = false;
void init_lut() {
  if (g_lut_ready) return;
  for (int i = 0; i < 256; ++i) {
    float c = (float)i / 255.0f;
    g_srgb_lut[i] = c <= 0.04045f ? c / 12.92f : std::pow((c + 0.055f) / 1.055f, 2.4f);
  }
  g_lut_ready = true;
}

inline uint8_t to_unorm8(float x) {
  if (!(x > 0.0f)) return 0;
  if (x > 1.0f) x = 1.0f;
  return (uint8_t)std::nearbyint(x * 255.0f);
}

/* ---------------- 2D sampling (material sampler, renderer.ts:822-829) ---------------- */
inline int wrap(int i, int n, bool repeat) {
  if (repeat) { i %= n; return i < 0 ? i + n : i; }
  return std::min(std::max(i, 0), n - 1);
}

struct Tex8 { const uint8_t* chain; int w, h, levels; bool srgb; };

// Test instrumentation, not part of the restated pipeline: the smallest distance of any DISCRETE sampling decision of
// the pixel being shaded from its switching point - the level switch of a "nearest" mipmap filter at lambda = k + 1/2
// (in levels), the min / mag filter switch at lambda = 0 when the two filters differ (in levels), the texel switch of a
// "nearest" texel filter (in texels).  A pixel whose margin is tiny may legitimately come out differently on an
// implementation whose derivatives or log2 differ in the last bits; the parity tests use this to show that every pixel
// beyond the colour tolerance is such a pixel.
static thread_local float t_margin = 1e30f;
inline void note_margin(float m) { if (m == m && m < t_margin) t_margin = m; }
// Test instrumentation too (orc_draw_frame_bounds): 0 = the restated pipeline; 1 - 4 = the tangent of shader.wgsl:102
// turned by as much as it is uncertain, one way or the other about the normal (1, 2) or towards it (3, 4).  The tangent
// is built from DIFFERENCES of neighbouring pixels' world positions and texcoords: positions of magnitude |p| known to
// an ulp or a few differ by one pixel's footprint, so the direction of their difference is only known to
// ulps * eps * |p| / footprint - 1e-4 to 1e-3 radians in a 4K frame - and two correct implementations (another order
// of the same interpolation, a fused multiply-add) differ by that much.  Nearly everywhere it moves the colour by
// nothing; at grazing pixels, where 4 N.V N.L + 0.001 is small, it moves it by several 8-bit steps, and where N.V
// crosses 1.5 / size of the BRDF table it decides whether the table's NaN column (brdf.wgsl:85-86) is in the footprint,
// i.e. whether the pixel is black.
static thread_local int t_tangent_variant = 0;
static thread_local float t_tangent_ulps = 4.0f;
inline float to_nearest_integer(float v) { return std::fabs(v - std::nearbyint(v)); }

inline V4 texel8(const Tex8& t, int level, int x, int y) {
  const uint8_t* p = t.chain + 4 * (chain_off(t.w, t.h, level) + (int64_t)y * level_dim(t.w, level) + x);
  if (t.srgb) return {g_srgb_lut[p[0]], g_srgb_lut[p[1]], g_srgb_lut[p[2]], (float)p[3] / 255.0f};
  return {(float)p[0] / 255.0f, (float)p[1] / 255.0f, (float)p[2] / 255.0f, (float)p[3] / 255.0f};
}

inline V4 bilerp(V4 t00, V4 t10, V4 t01, V4 t11, float a, float b) {
  float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
  return {((t00.x * w00 + t10.x * w10) + t01.x * w01) + t11.x * w11,
          ((t00.y * w00 + t10.y * w10) + t01.y * w01) + t11.y * w11,
          ((t00.z * w00 + t10.z * w10) + t01.z * w01) + t11.z * w11,
          ((t00.w * w00 + t10.w * w10) + t01.w * w01) + t11.w * w11};
}

// textureSample(t, sampler2D, uv) with implicit derivatives duvdx/duvdy.
V4 sample_2d(const Tex8& t, uint32_t sampler_index, V2 uv, V2 ddx, V2 ddy) {
  const bool rep_u = sampler_index & 1, rep_v = sampler_index & 2, mag_lin = sampler_index & 4, min_lin = sampler_index & 8;
  float fw = (float)t.w, fh = (float)t.h;
  float mux = ddx.x * fw, mvx = ddx.y * fh, muy = ddy.x * fw, mvy = ddy.y * fh;
  float rho = std::max(std::sqrt(mux * mux + mvx * mvx), std::sqrt(muy * muy + mvy * mvy));
  float lambda = std::log2(rho);                       // -inf for rho = 0
  if (std::isnan(lambda)) lambda = 0.0f;
  if (mag_lin != min_lin) note_margin(std::fabs(lambda));
  lambda = std::min(std::max(lambda, 0.0f), 32.0f);    // sampler lodMinClamp / lodMaxClamp defaults
  const bool linear = lambda <= 0.0f ? mag_lin : min_lin;
  lambda = std::min(lambda, (float)(t.levels - 1));
  if (t.levels > 1 && lambda < (float)(t.levels - 1)) note_margin(std::fabs(lambda + 0.5f - std::nearbyint(lambda + 0.5f)));
  int level = (int)std::ceil(lambda + 0.5f) - 1;       // mipmapFilter defaults to "nearest"
  level = std::min(std::max(level, 0), t.levels - 1);
  int lw = level_dim(t.w, level), lh = level_dim(t.h, level);
  if (std::isnan(uv.x) || std::isnan(uv.y)) { float n = std::nanf(""); return {n, n, n, n}; }
  if (!linear) {
    note_margin(std::min(to_nearest_integer(uv.x * (float)lw), to_nearest_integer(uv.y * (float)lh)));
    int x = wrap((int)std::floor(uv.x * (float)lw), lw, rep_u), y = wrap((int)std::floor(uv.y * (float)lh), lh, rep_v);
    return texel8(t, level, x, y);
  }
  float tu = uv.x * (float)lw - 0.5f, tv = uv.y * (float)lh - 0.5f;
  float fu = std::floor(tu), fv = std::floor(tv);
  float a = tu - fu, b = tv - fv;
  int x0 = wrap((int)fu, lw, rep_u), x1 = wrap((int)fu + 1, lw, rep_u);
  int y0 = wrap((int)fv, lh, rep_v), y1 = wrap((int)fv + 1, lh, rep_v);
  return bilerp(texel8(t, level, x0, y0), texel8(t, level, x1, y0), texel8(t, level, x0, y1), texel8(t, level, x1, y1), a, b);
}

/* rgba16float LUT sampled with the same material sampler (shader.wgsl:150); one level. */
V4 sample_lut(const uint16_t* lut, int size, uint32_t sampler_index, V2 uv, V2 ddx, V2 ddy) {
  const bool rep_u = sampler_index & 1, rep_v = sampler_index & 2, mag_lin = sampler_index & 4, min_lin = sampler_index & 8;
  float fs = (float)size;
  float mux = ddx.x * fs, mvx = ddx.y * fs, muy = ddy.x * fs, mvy = ddy.y * fs;
  float rho = std::max(std::sqrt(mux * mux + mvx * mvx), std::sqrt(muy * muy + mvy * mvy));
  float lambda = std::log2(rho);
  if (std::isnan(lambda)) lambda = 0.0f;
  if (mag_lin != min_lin) note_margin(std::fabs(lambda));
  lambda = std::min(std::max(lambda, 0.0f), 32.0f);
  const bool linear = lambda <= 0.0f ? mag_lin : min_lin;   // the single level is used either way
  auto tx = [&](int x, int y) -> V4 {
    const uint16_t* p = lut + 4 * ((int64_t)y * size + x);
    return {orc_f16_to_f32(p[0]), orc_f16_to_f32(p[1]), orc_f16_to_f32(p[2]), orc_f16_to_f32(p[3])};
  };
  if (std::isnan(uv.x) || std::isnan(uv.y)) { float n = std::nanf(""); return {n, n, n, n}; }
  if (!linear) {
    note_margin(std::min(to_nearest_integer(uv.x * fs), to_nearest_integer(uv.y * fs)));
    return tx(wrap((int)std::floor(uv.x * fs), size, rep_u), wrap((int)std::floor(uv.y * fs), size, rep_v));
  }
  float tu = uv.x * fs - 0.5f, tv = uv.y * fs - 0.5f;
  float fu = std::floor(tu), fv = std::floor(tv);
  int x0 = wrap((int)fu, size, rep_u), x1 = wrap((int)fu + 1, size, rep_u);
  int y0 = wrap((int)fv, size, rep_v), y1 = wrap((int)fv + 1, size, rep_v);
  return bilerp(tx(x0, y0), tx(x1, y0), tx(x0, y1), tx(x1, y1), tu - fu, tv - fv);
}

/* ---------------- cube sampling (renderer.ts:831-839, :948-956) ---------------- */
struct CubeCoord { int face; float sc, tc, rc; };

inline CubeCoord cube_select(V3 r) {
  float ax = std::fabs(r.x), ay = std::fabs(r.y), az = std::fabs(r.z);
  CubeCoord c;
  if (az >= ax && az >= ay) {
    if (r.z >= 0) { c.face = 4; c.sc = r.x; c.tc = -r.y; } else { c.face = 5; c.sc = -r.x; c.tc = -r.y; }
    c.rc = r.z;
  } else if (ay >= ax) {
    if (r.y >= 0) { c.face = 2; c.sc = r.x; c.tc = r.z; } else { c.face = 3; c.sc = r.x; c.tc = -r.z; }
    c.rc = r.y;
  } else {
    if (r.x >= 0) { c.face = 0; c.sc = -r.z; c.tc = -r.y; } else { c.face = 1; c.sc = r.z; c.tc = -r.y; }
    c.rc = r.x;
  }
  return c;
}
// (sc, tc, rc) of a derivative vector seen through the table row of `face`.
inline void cube_table(int face, V3 d, float* sc, float* tc, float* rc) {
  switch (face) {
    case 0: *sc = -d.z; *tc = -d.y; *rc = d.x; break;
    case 1: *sc = d.z;  *tc = -d.y; *rc = d.x; break;
    case 2: *sc = d.x;  *tc = d.z;  *rc = d.y; break;
    case 3: *sc = d.x;  *tc = -d.z; *rc = d.y; break;
    case 4: *sc = d.x;  *tc = -d.y; *rc = d.z; break;
    default: *sc = -d.x; *tc = -d.y; *rc = d.z; break;
  }
}
// Direction through face-plane point (sc, tc) of `face` (|rc| = 1).
inline V3 cube_dir(int face, float sc, float tc) {
  switch (face) {
    case 0: return {1.0f, -tc, -sc};
    case 1: return {-1.0f, -tc, sc};
    case 2: return {sc, 1.0f, tc};
    case 3: return {sc, -1.0f, -tc};
    case 4: return {sc, -tc, 1.0f};
    default: return {-sc, -tc, -1.0f};
  }
}

struct CubeTex { const float* chain; int size, levels; };
inline int64_t cube_level_off(int size, int level) {   // in texels
  int64_t o = 0;
  for (int l = 0; l < level; ++l) { int64_t s = level_dim(size, l); o += 6 * s * s; }
  return o;
}
inline V4 cube_texel(const CubeTex& t, int level, int face, int x, int y) {
  int n = level_dim(t.size, level);
  const float* p = t.chain + 4 * (cube_level_off(t.size, level) + ((int64_t)face * n + y) * n + x);
  return {p[0], p[1], p[2], p[3]};
}
// Tap (i, j) of `face`, following the cube across an edge when one index is outside [0, n).
// Returns false for a corner tap (both outside), which has no texel.
bool cube_tap(const CubeTex& t, int level, int face, int i, int j, V4* out) {
  int n = level_dim(t.size, level);
  bool ox = i < 0 || i >= n, oy = j < 0 || j >= n;
  if (!ox && !oy) { *out = cube_texel(t, level, face, i, j); return true; }
  if (ox && oy) return false;
  float sc = ((float)i + 0.5f) / (float)n * 2.0f - 1.0f, tc = ((float)j + 0.5f) / (float)n * 2.0f - 1.0f;
  CubeCoord c = cube_select(cube_dir(face, sc, tc));
  float ma = std::fabs(c.rc);
  float s = 0.5f * (c.sc / ma) + 0.5f, tt = 0.5f * (c.tc / ma) + 0.5f;
  int x = std::min(std::max((int)std::floor(s * (float)n), 0), n - 1);
  int y = std::min(std::max((int)std::floor(tt * (float)n), 0), n - 1);
  *out = cube_texel(t, level, c.face, x, y);
  return true;
}

V4 cube_sample_level(const CubeTex& t, int level, const CubeCoord& c) {
  int n = level_dim(t.size, level);
  float ma = std::fabs(c.rc);
  float s = 0.5f * (c.sc / ma) + 0.5f, tt = 0.5f * (c.tc / ma) + 0.5f;
  float tu = s * (float)n - 0.5f, tv = tt * (float)n - 0.5f;
  float fu = std::floor(tu), fv = std::floor(tv);
  float a = tu - fu, b = tv - fv;
  int i0 = (int)fu, j0 = (int)fv;
  V4 tap[4]; bool ok[4];
  ok[0] = cube_tap(t, level, c.face, i0, j0, &tap[0]);
  ok[1] = cube_tap(t, level, c.face, i0 + 1, j0, &tap[1]);
  ok[2] = cube_tap(t, level, c.face, i0, j0 + 1, &tap[2]);
  ok[3] = cube_tap(t, level, c.face, i0 + 1, j0 + 1, &tap[3]);
  for (int k = 0; k < 4; ++k)
    if (!ok[k]) {
      V4 s3 = {0, 0, 0, 0};
      for (int m = 0; m < 4; ++m) if (m != k) { s3.x += tap[m].x; s3.y += tap[m].y; s3.z += tap[m].z; s3.w += tap[m].w; }
      tap[k] = {s3.x / 3.0f, s3.y / 3.0f, s3.z / 3.0f, s3.w / 3.0f};
    }
  return bilerp(tap[0], tap[1], tap[2], tap[3], a, b);
}

// lambda of an implicit-LOD cube lookup of direction r with quad differences drdx, drdy.
float cube_lambda(int size, V3 r, V3 drdx, V3 drdy) {
  CubeCoord c = cube_select(r);
  {
    // (test instrumentation, see note_margin) the face is a discrete decision, and the derivatives are projected through
    // the chosen face's table: across a face edge the level of detail jumps.  Margin = how far the two largest components
    // of the direction are apart, relative to the larger.
    float a[3] = {std::fabs(r.x), std::fabs(r.y), std::fabs(r.z)};
    std::sort(a, a + 3);
    if (a[2] > 0.0f) note_margin((a[2] - a[1]) / a[2]);
  }
  float ma = std::fabs(c.rc), sgn = c.rc >= 0 ? 1.0f : -1.0f;
  float rho[2];
  const V3 d[2] = {drdx, drdy};
  for (int k = 0; k < 2; ++k) {
    float dsc, dtc, drc; cube_table(c.face, d[k], &dsc, &dtc, &drc);
    float dma = sgn * drc;
    float ds = 0.5f * ((ma * dsc - c.sc * dma) / (c.rc * c.rc));
    float dt = 0.5f * ((ma * dtc - c.tc * dma) / (c.rc * c.rc));
    float mu = ds * (float)size, mv = dt * (float)size;
    rho[k] = std::sqrt(mu * mu + mv * mv);
  }
  return std::log2(std::max(rho[0], rho[1]));
}

// textureSample(cube, samplerCube, r) at LOD lambda: linear min/mag/mip, seamless.
V4 cube_sample(const CubeTex& t, V3 r, float lambda) {
  if (has_nan(r)) { float n = std::nanf(""); return {n, n, n, n}; }   // a NaN direction selects no face
  CubeCoord c = cube_select(r);
  if (std::isnan(lambda)) lambda = 0.0f;
  lambda = std::min(std::max(lambda, 0.0f), (float)(t.levels - 1));
  int l0 = (int)std::floor(lambda), l1 = std::min(l0 + 1, t.levels - 1);
  float f = lambda - (float)l0;
  V4 a = cube_sample_level(t, l0, c);
  if (f == 0.0f || l1 == l0) return a;
  V4 b = cube_sample_level(t, l1, c);
  return {a.x * (1.0f - f) + b.x * f, a.y * (1.0f - f) + b.y * f, a.z * (1.0f - f) + b.z * f, a.w * (1.0f - f) + b.w * f};
}

/* ---------------- vertex stage and triangle setup ---------------- */
struct Ctx {
  const OrcScene* sc;
  const OrcFrameUniforms* u;
  int W, H;
  int n_tris;
  std::vector<float> clip, world, nrm;   // per instance vertex
};

void run_vertex_stage(const OrcScene* sc, const OrcFrameUniforms* u, float* clip, float* world, float* nrm) {
  float pv[16];
  mat_mul(u->projection, u->view, pv);
  for (int inst = 0; inst < sc->n_instances; ++inst) {
    const float* model = sc->inst_model ? sc->inst_model + 16 * inst : u->model;
    const float* nmat = sc->inst_normal ? sc->inst_normal + 16 * inst : u->normal;
    float pvm[16];
    mat_mul(pv, model, pvm);   // projection * view * model, left to right (shader.wgsl:41)
#pragma omp parallel for schedule(static)
    for (int i = 0; i < sc->n_verts; ++i) {
      const float* v = sc->verts + 8 * (int64_t)i;
      int64_t o = (int64_t)inst * sc->n_verts + i;
      V4 c = mat_vec(pvm, v[0], v[1], v[2], 1.0f);
      V4 n = mat_vec(nmat, v[5], v[6], v[7], 0.0f);
      V4 w = mat_vec(model, v[0], v[1], v[2], 1.0f);
      clip[4 * o] = c.x; clip[4 * o + 1] = c.y; clip[4 * o + 2] = c.z; clip[4 * o + 3] = c.w;
      nrm[3 * o] = n.x; nrm[3 * o + 1] = n.y; nrm[3 * o + 2] = n.z;
      world[3 * o] = w.x; world[3 * o + 1] = w.y; world[3 * o + 2] = w.z;
    }
  }
}

struct Snapped { int32_t X, Y; float z; };
inline Snapped snap(V4 c, int W, int H) {
  float nx = c.x / c.w, ny = c.y / c.w, nz = c.z / c.w;
  float xf = (nx + 1.0f) * 0.5f * (float)W;
  float yf = (1.0f - ny) * 0.5f * (float)H;
  Snapped s;
  s.X = (int32_t)std::nearbyint(xf * 256.0f);
  s.Y = (int32_t)std::nearbyint(yf * 256.0f);
  s.z = nz;
  return s;
}

struct SetupTri {
  uint32_t id;
  int32_t X[3], Y[3];
  float z0, dz1, dz2;
  int64_t area;
  int minx, maxx, miny, maxy;   // pixel bounds, inclusive
};

const float GUARD = 8.0f;   // guard band: |x|,|y| <= GUARD*w keeps fixed-point coordinates in range

inline float plane_dist(int p, V4 c) {
  switch (p) {
    case 0: return c.z;                 // near: z >= 0
    case 1: return c.w - c.z;           // far: z <= w
    case 2: return GUARD * c.w - c.x;
    case 3: return GUARD * c.w + c.x;
    case 4: return GUARD * c.w - c.y;
    default: return GUARD * c.w + c.y;
  }
}

// Emits the snapped, front-facing setup record of one (sub)triangle, or nothing.
void emit_triangle(uint32_t id, V4 a, V4 b, V4 c, int W, int H, int y0, int y1, bool msaa, std::vector<SetupTri>& out) {
  Snapped s[3] = {snap(a, W, H), snap(b, W, H), snap(c, W, H)};
  int64_t area2 = (int64_t)(s[1].X - s[0].X) * (s[2].Y - s[0].Y) - (int64_t)(s[2].X - s[0].X) * (s[1].Y - s[0].Y);
  if (area2 >= 0) return;            // back-facing (clockwise on screen) or zero area: cullMode "back"
  std::swap(s[1], s[2]);             // make the edge functions positive inside
  SetupTri t;
  t.id = id;
  for (int k = 0; k < 3; ++k) { t.X[k] = s[k].X; t.Y[k] = s[k].Y; }
  t.z0 = s[0].z; t.dz1 = s[1].z - s[0].z; t.dz2 = s[2].z - s[0].z;
  t.area = -area2;
  int minX = std::min(s[0].X, std::min(s[1].X, s[2].X)), maxX = std::max(s[0].X, std::max(s[1].X, s[2].X));
  int minY = std::min(s[0].Y, std::min(s[1].Y, s[2].Y)), maxY = std::max(s[0].Y, std::max(s[1].Y, s[2].Y));
  if (msaa) {
    // every pixel whose square meets [min, max]: its sample points lie strictly inside the square
    t.minx = std::max(minX >> 8, 0); t.maxx = std::min(maxX >> 8, W - 1);
    t.miny = std::max(minY >> 8, y0); t.maxy = std::min(maxY >> 8, y1 - 1);
  } else {
    // pixel centres px*256+128 inside [min, max]
    t.minx = std::max((minX - 128 + 255) >> 8, 0);
    t.maxx = std::min((maxX - 128) >> 8, W - 1);
    t.miny = std::max((minY - 128 + 255) >> 8, y0);
    t.maxy = std::min((maxY - 128) >> 8, y1 - 1);
  }
  if (t.minx > t.maxx || t.miny > t.maxy) return;
  out.push_back(t);
}

void setup_triangle(uint32_t id, V4 a, V4 b, V4 c, int W, int H, int y0, int y1, bool msaa, std::vector<SetupTri>& out) {
  const V4 v[3] = {a, b, c};
  bool inside_all = true;
  for (int p = 0; p < 6; ++p) {
    int outside = 0;
    for (int k = 0; k < 3; ++k) outside += plane_dist(p, v[k]) < 0.0f;
    if (outside == 3) return;         // entirely beyond one plane
    if (outside) inside_all = false;
  }
  // Entirely outside one side of the viewport: covers no pixel.
  {
    int l = 0, r = 0, bt = 0, tp = 0;
    for (int k = 0; k < 3; ++k) { l += v[k].x < -v[k].w; r += v[k].x > v[k].w; bt += v[k].y < -v[k].w; tp += v[k].y > v[k].w; }
    if (l == 3 || r == 3 || bt == 3 || tp == 3) return;
  }
  if (inside_all) { emit_triangle(id, a, b, c, W, H, y0, y1, msaa, out); return; }
  // Sutherland-Hodgman against the six planes; the polygon is fanned from its first vertex.
  V4 poly[16], tmp[16];
  int n = 3;
  poly[0] = a; poly[1] = b; poly[2] = c;
  for (int p = 0; p < 6 && n >= 3; ++p) {
    int m = 0;
    for (int k = 0; k < n; ++k) {
      V4 cur = poly[k], nxt = poly[(k + 1) % n];
      float dc = plane_dist(p, cur), dn = plane_dist(p, nxt);
      if (dc >= 0.0f) tmp[m++] = cur;
      if ((dc >= 0.0f) != (dn >= 0.0f)) {
        float t = dc / (dc - dn);
        tmp[m++] = {cur.x + (nxt.x - cur.x) * t, cur.y + (nxt.y - cur.y) * t, cur.z + (nxt.z - cur.z) * t, cur.w + (nxt.w - cur.w) * t};
      }
    }
    n = m;
    std::memcpy(poly, tmp, sizeof(V4) * n);
  }
  for (int k = 1; k + 1 < n; ++k) emit_triangle(id, poly[0], poly[k], poly[k + 1], W, H, y0, y1, msaa, out);
}

inline V4 load4(const float* p) { return {p[0], p[1], p[2], p[3]}; }
inline V3 load3(const float* p) { return {p[0], p[1], p[2]}; }

inline int64_t vertex_index(const OrcScene* sc, int64_t i) { return sc->indices32 ? (int64_t)sc->indices32[i] : (int64_t)sc->indices[i]; }

// Sample positions inside a pixel, in 1/256 pixel: the centre, or the standard four-sample pattern of
// D3D / Vulkan / Metal that WebGPU's `count: 4` (renderer.ts:492-501) maps to:
// (0.375, 0.125), (0.875, 0.375), (0.125, 0.625), (0.625, 0.875).
const int SAMPLE_X[4] = {96, 224, 32, 160}, SAMPLE_Y[4] = {32, 96, 160, 224};

// vis / depth hold `samples` entries per pixel ([y][x][sample]).  Coverage, the depth value and the depth
// test are per sample; the triangle id a sample keeps is what the fragment stage later shades.
void raster_impl(Ctx& cx, int y0, int y1, int samples, uint32_t* vis, float* depth) {
  const OrcScene* sc = cx.sc;
  const int W = cx.W, H = cx.H;
  const bool msaa = samples > 1;
  std::vector<SetupTri> tris;
  for (int inst = 0; inst < sc->n_instances; ++inst)
    for (int t = 0; t < cx.n_tris; ++t) {
      int64_t base = (int64_t)inst * sc->n_verts;
      const int64_t i0 = vertex_index(sc, 3 * (int64_t)t), i1 = vertex_index(sc, 3 * (int64_t)t + 1), i2 = vertex_index(sc, 3 * (int64_t)t + 2);
      setup_triangle((uint32_t)(inst * (int64_t)cx.n_tris + t), load4(&cx.clip[4 * (base + i0)]),
                     load4(&cx.clip[4 * (base + i1)]), load4(&cx.clip[4 * (base + i2)]), W, H, y0, y1, msaa, tris);
    }
  const int rows = y1 - y0;
  for (int64_t i = 0; i < (int64_t)rows * W * samples; ++i) { vis[i] = 0; depth[i] = 1.0f; }   // renderer.ts:1943-1966
  const int BAND = 32;
  const int n_bands = (rows + BAND - 1) / BAND;
#pragma omp parallel for schedule(dynamic, 1)
  for (int band = 0; band < n_bands; ++band) {
    int by0 = y0 + band * BAND, by1 = std::min(by0 + BAND, y1);
    for (const SetupTri& t : tris) {   // draw order: ties keep the earlier triangle ("less")
      int ya = std::max(t.miny, by0), yb = std::min(t.maxy, by1 - 1);
      if (ya > yb) continue;
      int64_t A[3], B[3]; bool tl[3]; int32_t ax[3], ay[3];
      for (int k = 0; k < 3; ++k) {
        int a = (k + 1) % 3, b = (k + 2) % 3;
        int64_t dx = (int64_t)t.X[b] - t.X[a], dy = (int64_t)t.Y[b] - t.Y[a];
        A[k] = -dy; B[k] = dx; ax[k] = t.X[a]; ay[k] = t.Y[a];
        tl[k] = dy < 0 || (dy == 0 && dx > 0);   // left edge, or top edge
      }
      const float inv_area = 1.0f / (float)t.area;
      for (int y = ya; y <= yb; ++y)
        for (int x = t.minx; x <= t.maxx; ++x)
          for (int sidx = 0; sidx < samples; ++sidx) {
            int64_t px = (int64_t)x * 256 + (msaa ? SAMPLE_X[sidx] : 128), py = (int64_t)y * 256 + (msaa ? SAMPLE_Y[sidx] : 128);
            int64_t E[3];
            bool in = true;
            for (int k = 0; k < 3; ++k) {
              E[k] = B[k] * (py - ay[k]) + A[k] * (px - ax[k]);
              in = in && (E[k] > 0 || (E[k] == 0 && tl[k]));
            }
            if (!in) continue;
            float l1 = (float)E[1] * inv_area, l2 = (float)E[2] * inv_area;
            float z = (t.z0 + l1 * t.dz1) + l2 * t.dz2;
            int64_t o = ((int64_t)(y - y0) * W + x) * samples + sidx;
            if (z < depth[o]) { depth[o] = z; vis[o] = t.id + 1; }
          }
    }
  }
}

/* ---------------- fragment stage ---------------- */
struct Varyings { V2 uv; V3 normal, pos; };

struct TriData {
  V4 c[3];
  V2 uv[3];
  V3 n[3], p[3];
  bool projective;         // all w > 0 and snapped coordinates usable
  int32_t X[3], Y[3];
  float inv_w[3];
  int64_t area;            // signed, of the snapped triangle
};

TriData fetch_tri(const Ctx& cx, uint32_t id) {
  const OrcScene* sc = cx.sc;
  int inst = (int)(id / (uint32_t)cx.n_tris), t = (int)(id % (uint32_t)cx.n_tris);
  int64_t base = (int64_t)inst * sc->n_verts;
  TriData d;
  d.projective = true;
  for (int k = 0; k < 3; ++k) {
    int vi = (int)vertex_index(sc, 3 * (int64_t)t + k);
    d.c[k] = load4(&cx.clip[4 * (base + vi)]);
    d.n[k] = load3(&cx.nrm[3 * (base + vi)]);
    d.p[k] = load3(&cx.world[3 * (base + vi)]);
    const float* v = sc->verts + 8 * (int64_t)vi;
    d.uv[k] = {v[3], v[4]};
    if (!(d.c[k].w > 0.0f) || std::fabs(d.c[k].x) > GUARD * d.c[k].w || std::fabs(d.c[k].y) > GUARD * d.c[k].w) d.projective = false;
  }
  if (d.projective) {
    for (int k = 0; k < 3; ++k) {
      Snapped s = snap(d.c[k], cx.W, cx.H);
      d.X[k] = s.X; d.Y[k] = s.Y; d.inv_w[k] = 1.0f / d.c[k].w;
    }
    d.area = (int64_t)(d.X[1] - d.X[0]) * (d.Y[2] - d.Y[0]) - (int64_t)(d.X[2] - d.X[0]) * (d.Y[1] - d.Y[0]);
    if (d.area == 0) d.projective = false;
  }
  return d;
}

// Perspective-correct varyings of triangle d at the centre of pixel (x, y).
Varyings interpolate(const Ctx& cx, const TriData& d, int x, int y) {
  float b0, b1, b2;
  if (d.projective) {
    // screen-space barycentrics from the snapped vertices (exact integers), then 1/w weighting
    int64_t px = (int64_t)x * 256 + 128, py = (int64_t)y * 256 + 128;
    int64_t e0 = (int64_t)(d.X[2] - d.X[1]) * (py - d.Y[1]) - (int64_t)(d.Y[2] - d.Y[1]) * (px - d.X[1]);
    int64_t e1 = (int64_t)(d.X[0] - d.X[2]) * (py - d.Y[2]) - (int64_t)(d.Y[0] - d.Y[2]) * (px - d.X[2]);
    float farea = (float)d.area;
    float l0 = (float)e0 / farea, l1 = (float)e1 / farea;
    float l2 = (1.0f - l0) - l1;
    float q0 = l0 * d.inv_w[0], q1 = l1 * d.inv_w[1], q2 = l2 * d.inv_w[2];
    float qs = (q0 + q1) + q2;
    b0 = q0 / qs; b1 = q1 / qs; b2 = q2 / qs;
  } else {
    // a vertex behind the eye (or beyond the guard band): 2D homogeneous barycentrics
    float xn = ((float)x + 0.5f) / (float)cx.W * 2.0f - 1.0f;
    float yn = 1.0f - ((float)y + 0.5f) / (float)cx.H * 2.0f;
    const V4* c = d.c;
    float h0 = (c[1].y * c[2].w - c[2].y * c[1].w) * xn + (c[2].x * c[1].w - c[1].x * c[2].w) * yn + (c[1].x * c[2].y - c[2].x * c[1].y);
    float h1 = (c[2].y * c[0].w - c[0].y * c[2].w) * xn + (c[0].x * c[2].w - c[2].x * c[0].w) * yn + (c[2].x * c[0].y - c[0].x * c[2].y);
    float h2 = (c[0].y * c[1].w - c[1].y * c[0].w) * xn + (c[1].x * c[0].w - c[0].x * c[1].w) * yn + (c[0].x * c[1].y - c[1].x * c[0].y);
    float hs = (h0 + h1) + h2;
    b0 = h0 / hs; b1 = h1 / hs; b2 = h2 / hs;
  }
  Varyings v;
  v.uv = {(d.uv[0].x * b0 + d.uv[1].x * b1) + d.uv[2].x * b2, (d.uv[0].y * b0 + d.uv[1].y * b1) + d.uv[2].y * b2};
  v.normal = add(add(mul(d.n[0], b0), mul(d.n[1], b1)), mul(d.n[2], b2));
  v.pos = add(add(mul(d.p[0], b0), mul(d.p[1], b1)), mul(d.p[2], b2));
  return v;
}

// shader.wgsl:52-63
float ggx_d(V3 n, V3 h, float roughness) {
  float a = roughness * roughness, a2 = a * a;
  float ndoth = std::max(dot(n, h), 0.0f), ndoth2 = ndoth * ndoth;
  float denom = ndoth2 * (a2 - 1.0f) + 1.0f;
  denom = PI * denom * denom;
  return a2 / denom;
}
// shader.wgsl:83-91
float smith_ggx(float c, float roughness) {
  float r = roughness + 1.0f, k = (r * r) / 8.0f;
  return c / (c * (1.0f - k) + k);
}

struct Material { Tex8 albedo, mr, normal, emissive, occlusion; };

struct QuadShared { V2 duvdx, duvdy; V3 T; };

// shader.wgsl:96,101-107,111,148 for one (possibly helper) invocation: everything the two
// derivative-dependent lookups (prefilter cube, BRDF LUT) take as coordinates.
struct FragCommon { V3 normal, view_dir, R; float ndotv, roughness, metalness; };
FragCommon eval_common(const Ctx& cx, const Material& m, const QuadShared& q, const Varyings& v) {
  const OrcFrameUniforms* u = cx.u;
  FragCommon o;
  V4 ns = sample_2d(m.normal, u->sampler_index, v.uv, q.duvdx, q.duvdy);
  V3 normal = {ns.x * 2.0f - 1.0f, ns.y * 2.0f - 1.0f, ns.z * 2.0f - 1.0f};
  V3 N = normalize(v.normal);
  V3 B = neg(normalize(cross(N, q.T)));
  o.normal = normalize(add(add(mul(q.T, normal.x), mul(B, normal.y)), mul(N, normal.z)));
  V4 mr = sample_2d(m.mr, u->sampler_index, v.uv, q.duvdx, q.duvdy);
  o.metalness = mr.z; o.roughness = mr.y;
  o.view_dir = normalize(sub(V3{u->view_pos[0], u->view_pos[1], u->view_pos[2]}, v.pos));
  V3 I = neg(o.view_dir);
  V3 R = sub(I, mul(o.normal, 2.0f * dot(o.normal, I)));   // reflect(-viewDir, normal)
  o.R = {R.x, R.y, -R.z};
  o.ndotv = std::max(dot(o.normal, o.view_dir), 0.0f);
  return o;
}

void shade_pixel(const Ctx& cx, const Material& m, const CubeTex& irr, const CubeTex& pre, uint32_t id, int x, int y, uint8_t* out) {
  const OrcFrameUniforms* u = cx.u;
  const OrcScene* sc = cx.sc;
  TriData d = fetch_tri(cx, id);
  int qx = x & ~1, qy = y & ~1;
  Varyings h0 = interpolate(cx, d, qx, qy), h1 = interpolate(cx, d, qx + 1, qy), h2 = interpolate(cx, d, qx, qy + 1);
  Varyings f = interpolate(cx, d, x, y);
  QuadShared q;
  V3 q1 = sub(h1.pos, h0.pos), q2 = sub(h2.pos, h0.pos);                       // dpdx/dpdy(pos), shader.wgsl:97-98
  q.duvdx = {h1.uv.x - h0.uv.x, h1.uv.y - h0.uv.y};                           // shader.wgsl:99-100
  q.duvdy = {h2.uv.x - h0.uv.x, h2.uv.y - h0.uv.y};
  q.T = normalize(sub(mul(q1, q.duvdy.y), mul(q2, q.duvdx.y)));               // shader.wgsl:102
  if (t_tangent_variant) {   // (test instrumentation, see t_tangent_variant)
    const float eps = 5.96e-8f;
    const float lt = length(sub(mul(q1, q.duvdy.y), mul(q2, q.duvdx.y)));
    const float err = t_tangent_ulps * eps * (length(f.pos) * (std::fabs(q.duvdy.y) + std::fabs(q.duvdx.y)) + (length(q1) + length(q2)) * std::max(std::fabs(f.uv.y), 1e-3f));
    const float delta = lt > 0.0f ? std::min(err / lt, 0.25f) : 0.0f;
    const V3 Nf = normalize(f.normal);
    const V3 axis = t_tangent_variant <= 2 ? cross(Nf, q.T) : Nf;
    q.T = normalize(add(q.T, mul(axis, (t_tangent_variant & 1) ? delta : -delta)));
  }

  FragCommon c0 = eval_common(cx, m, q, h0), c1 = eval_common(cx, m, q, h1), c2 = eval_common(cx, m, q, h2);
  FragCommon fc0 = eval_common(cx, m, q, f);
  V3 normal = fc0.normal, view_dir = fc0.view_dir, R = fc0.R;
  float metalness = fc0.metalness, roughness = fc0.roughness;

  V4 alb4 = sample_2d(m.albedo, u->sampler_index, f.uv, q.duvdx, q.duvdy);
  V3 albedo = {alb4.x, alb4.y, alb4.z};
  V4 em4 = sample_2d(m.emissive, u->sampler_index, f.uv, q.duvdx, q.duvdy);
  V3 emissive = {em4.x * u->emissive_factor[0], em4.y * u->emissive_factor[1], em4.z * u->emissive_factor[2]};
  float occlusion = sample_2d(m.occlusion, u->sampler_index, f.uv, q.duvdx, q.duvdy).x;

  V3 f0 = {0.04f * (1.0f - metalness) + albedo.x * metalness, 0.04f * (1.0f - metalness) + albedo.y * metalness,
           0.04f * (1.0f - metalness) + albedo.z * metalness};                 // mix, shader.wgsl:113

  V3 Lo = {0, 0, 0};
  for (int i = 0; i < 4; ++i) {                                                // shader.wgsl:118-139
    V3 lp = {u->light_pos[i][0], u->light_pos[i][1], u->light_pos[i][2]};
    V3 lc = {u->light_color[i][0], u->light_color[i][1], u->light_color[i][2]};
    V3 to_light = sub(lp, f.pos);
    V3 light_dir = normalize(to_light);
    V3 halfway = normalize(add(view_dir, light_dir));
    float distance = length(to_light);
    float attenuation = 1.0f / (distance * distance);
    V3 radiance = mul(lc, attenuation);
    float D = ggx_d(normal, halfway, roughness);
    float fc = std::pow(clamp01(1.0f - std::max(dot(halfway, view_dir), 0.0f)), 5.0f);
    V3 F = {f0.x + (1.0f - f0.x) * fc, f0.y + (1.0f - f0.y) * fc, f0.z + (1.0f - f0.z) * fc};
    float ndotv = std::max(dot(normal, view_dir), 0.0f), ndotl = std::max(dot(normal, light_dir), 0.0f);
    float G = smith_ggx(ndotv, roughness) * smith_ggx(ndotl, roughness);
    V3 kD = {(1.0f - F.x) * (1.0f - metalness), (1.0f - F.y) * (1.0f - metalness), (1.0f - F.z) * (1.0f - metalness)};
    float denom = 4.0f * ndotv * ndotl + 0.001f;
    V3 spec = {D * F.x * G / denom, D * F.y * G / denom, D * F.z * G / denom};
    Lo.x += (kD.x * albedo.x / PI + spec.x) * radiance.x * ndotl;
    Lo.y += (kD.y * albedo.y / PI + spec.y) * radiance.y * ndotl;
    Lo.z += (kD.z * albedo.z / PI + spec.z) * radiance.z * ndotl;
  }

  float ndotv = std::max(dot(normal, view_dir), 0.0f);
  float fc = std::pow(clamp01(1.0f - ndotv), 5.0f);                            // shader.wgsl:69-71
  float omr = 1.0f - roughness;
  V3 kS = {f0.x + (std::max(omr, f0.x) - f0.x) * fc, f0.y + (std::max(omr, f0.y) - f0.y) * fc, f0.z + (std::max(omr, f0.z) - f0.z) * fc};
  V3 kD = {(1.0f - kS.x) * (1.0f - metalness), (1.0f - kS.y) * (1.0f - metalness), (1.0f - kS.z) * (1.0f - metalness)};

  V4 irr4 = cube_sample(irr, V3{normal.x, normal.y, -normal.z}, 0.0f);         // shader.wgsl:145 (single level)
  V3 diffuse = {irr4.x * albedo.x, irr4.y * albedo.y, irr4.z * albedo.z};
  float lambda = cube_lambda(pre.size, R, sub(c1.R, c0.R), sub(c2.R, c0.R));   // implicit LOD, shader.wgsl:149
  V4 pf = cube_sample(pre, R, lambda);
  V2 lut_dx = {c1.ndotv - c0.ndotv, c1.roughness - c0.roughness}, lut_dy = {c2.ndotv - c0.ndotv, c2.roughness - c0.roughness};
  V4 brdf = sample_lut(sc->lut_f16, sc->lut_size, u->sampler_index, V2{ndotv, roughness}, lut_dx, lut_dy);   // shader.wgsl:150
  V3 specular = {pf.x * (kS.x * brdf.x + brdf.y), pf.y * (kS.y * brdf.x + brdf.y), pf.z * (kS.z * brdf.x + brdf.y)};
  V3 ambient = {(kD.x * diffuse.x + specular.x) * occlusion, (kD.y * diffuse.y + specular.y) * occlusion, (kD.z * diffuse.z + specular.z) * occlusion};
  V3 color = {ambient.x + Lo.x + emissive.x, ambient.y + Lo.y + emissive.y, ambient.z + Lo.z + emissive.z};
  const float g = 1.0f / 2.2f;
  out[0] = to_unorm8(std::pow(color.x, g)); out[1] = to_unorm8(std::pow(color.y, g)); out[2] = to_unorm8(std::pow(color.z, g));
  out[3] = 255;
}

// skybox.wgsl:24-33 at the centre of pixel (x, y); the varying is clip position (x_ndc, y_ndc, 1, 1).
void shade_sky(const Ctx& cx, const CubeTex& sky, int x, int y, uint8_t* out) {
  const float* m = cx.u->inv_view_proj;
  float xn = ((float)x + 0.5f) / (float)cx.W * 2.0f - 1.0f;
  float yn = 1.0f - ((float)y + 0.5f) / (float)cx.H * 2.0f;
  V4 t = mat_vec(m, xn, yn, 1.0f, 1.0f);
  V3 d = normalize(V3{t.x / t.w, t.y / t.w, t.z / t.w});
  V4 c = cube_sample(sky, V3{d.x, d.y, -d.z}, 0.0f);
  const float g = 1.0f / 2.2f;
  out[0] = to_unorm8(std::pow(c.x, g)); out[1] = to_unorm8(std::pow(c.y, g)); out[2] = to_unorm8(std::pow(c.z, g));
  out[3] = 255;
}

Ctx make_ctx(const OrcScene* sc, const OrcFrameUniforms* u, int W, int H) {
  init_lut();
  Ctx cx;
  cx.sc = sc; cx.u = u; cx.W = W; cx.H = H;
  cx.n_tris = sc->n_indices / 3;
  int64_t nv = (int64_t)sc->n_instances * sc->n_verts;
  cx.clip.resize(4 * nv); cx.world.resize(3 * nv); cx.nrm.resize(3 * nv);
  run_vertex_stage(sc, u, cx.clip.data(), cx.world.data(), cx.nrm.data());
  return cx;
}

// With `samples` > 1 every triangle seen by a pixel's samples is shaded once, at the pixel centre
// ("center" interpolation, the WGSL default), the 8-bit result is what its samples store, and the
// resolve averages the stored values (round half to even; the rounding of a multisample resolve is
// left to the implementation).  The skybox triangle covers the samples no geometry claimed.
void shade_impl(Ctx& cx, int y0, int y1, int samples, const uint32_t* vis, uint8_t* rgba8, float* margin = nullptr, int tangent_variant = 0,
                float tangent_ulps = 4.0f) {
  const OrcScene* sc = cx.sc;
  const OrcFrameUniforms* u = cx.u;
  Material m;
  const bool srgb[5] = {true, false, false, true, false};   // renderer.ts:647,663,681,697,713
  Tex8* slots[5] = {&m.albedo, &m.mr, &m.normal, &m.emissive, &m.occlusion};
  for (int i = 0; i < 5; ++i)
    *slots[i] = {sc->tex[i], sc->tex_w[i], sc->tex_h[i], u->use_mipmaps ? sc->tex_levels[i] : 1, srgb[i]};
  CubeTex irr = {sc->irradiance, sc->irr_size, 1}, pre = {sc->prefilter, sc->pre_size, sc->pre_levels}, sky = {sc->skybox, sc->sky_size, 1};
  const int W = cx.W;
#pragma omp parallel for schedule(dynamic, 4)
  for (int y = y0; y < y1; ++y)
    for (int x = 0; x < W; ++x) {
      int64_t o = (int64_t)(y - y0) * W + x;
      t_margin = 1e30f;
      t_tangent_variant = tangent_variant; t_tangent_ulps = tangent_ulps;
      if (samples == 1) {
        if (vis[o]) shade_pixel(cx, m, irr, pre, vis[o] - 1, x, y, rgba8 + 4 * o);
        else shade_sky(cx, sky, x, y, rgba8 + 4 * o);
        if (margin) margin[o] = t_margin;
        continue;
      }
      const uint32_t* ids = vis + o * samples;
      int sum[3] = {0, 0, 0};
      for (int a = 0; a < samples; ++a) {
        bool seen = false;
        for (int b = 0; b < a; ++b) seen = seen || ids[b] == ids[a];
        if (seen) continue;
        int count = 0;
        for (int b = a; b < samples; ++b) count += ids[b] == ids[a];
        uint8_t c[4];
        if (ids[a]) shade_pixel(cx, m, irr, pre, ids[a] - 1, x, y, c);
        else shade_sky(cx, sky, x, y, c);
        for (int k = 0; k < 3; ++k) sum[k] += count * c[k];
      }
      for (int k = 0; k < 3; ++k) {
        int q = sum[k] / samples, rem = sum[k] % samples;   // samples == 4
        if (2 * rem > samples || (2 * rem == samples && (q & 1))) ++q;
        rgba8[4 * o + k] = (uint8_t)q;
      }
      rgba8[4 * o + 3] = 255;
      if (margin) margin[o] = t_margin;
    }
}

}  // namespace

extern "C" {

void orc_vertex_transform(const OrcScene* sc, const OrcFrameUniforms* u, float* clip, float* world, float* nrm) {
  run_vertex_stage(sc, u, clip, world, nrm);
}

// Bit 16 of the sampler index selects the 4x multisampled pipeline (renderer.ts:1927-1932, :1951-1954);
// vis and depth then hold four entries per pixel.
static int sample_count(const OrcFrameUniforms* u) { return (u->sampler_index & 16u) ? 4 : 1; }

void orc_raster(const OrcScene* sc, const OrcFrameUniforms* u, int W, int H, int y0, int y1, uint32_t* vis, float* depth) {
  Ctx cx = make_ctx(sc, u, W, H);
  const int samples = sample_count(u);
  std::vector<float> d;
  if (!depth) { d.resize((size_t)(y1 - y0) * W * samples); depth = d.data(); }
  raster_impl(cx, y0, y1, samples, vis, depth);
}

void orc_shade(const OrcScene* sc, const OrcFrameUniforms* u, int W, int H, int y0, int y1, const uint32_t* vis, uint8_t* rgba8) {
  Ctx cx = make_ctx(sc, u, W, H);
  shade_impl(cx, y0, y1, sample_count(u), vis, rgba8);
}

void orc_draw_frame(const OrcScene* sc, const OrcFrameUniforms* u, int W, int H, int y0, int y1, uint8_t* rgba8, uint32_t* vis_or_null) {
  Ctx cx = make_ctx(sc, u, W, H);
  const int samples = sample_count(u);
  std::vector<uint32_t> v;
  std::vector<float> d((size_t)(y1 - y0) * W * samples);
  uint32_t* vis = vis_or_null;
  if (!vis) { v.resize((size_t)(y1 - y0) * W * samples); vis = v.data(); }
  raster_impl(cx, y0, y1, samples, vis, d.data());
  shade_impl(cx, y0, y1, samples, vis, rgba8);
}

// orc_draw_frame plus, per pixel, the decision margin described at note_margin (1e30 where no discrete decision was taken).
void orc_draw_frame_margins(const OrcScene* sc, const OrcFrameUniforms* u, int W, int H, int y0, int y1, uint8_t* rgba8, uint32_t* vis, float* margin) {
  Ctx cx = make_ctx(sc, u, W, H);
  const int samples = sample_count(u);
  std::vector<float> d((size_t)(y1 - y0) * W * samples);
  raster_impl(cx, y0, y1, samples, vis, d.data());
  shade_impl(cx, y0, y1, samples, vis, rgba8, margin);
}

// orc_draw_frame_margins plus, per pixel and channel, the lowest and highest 8-bit value the pixel takes when the tangent
// is turned by its own uncertainty (`tangent_ulps` ulps of the positions and texcoords it is built from; see
// t_tangent_variant).  Single-sample pipeline only.
void orc_draw_frame_bounds(const OrcScene* sc, const OrcFrameUniforms* u, int W, int H, int y0, int y1, float tangent_ulps, uint8_t* rgba8, uint32_t* vis,
                           float* margin, uint8_t* lo8, uint8_t* hi8) {
  Ctx cx = make_ctx(sc, u, W, H);
  std::vector<float> d((size_t)(y1 - y0) * W);
  raster_impl(cx, y0, y1, 1, vis, d.data());
  shade_impl(cx, y0, y1, 1, vis, rgba8, margin);
  const size_t n = (size_t)(y1 - y0) * W * 4;
  std::memcpy(lo8, rgba8, n); std::memcpy(hi8, rgba8, n);
  std::vector<uint8_t> other(n);
  for (int variant = 1; variant <= 4; ++variant) {
    shade_impl(cx, y0, y1, 1, vis, other.data(), nullptr, variant, tangent_ulps);
    for (size_t i = 0; i < n; ++i) { lo8[i] = std::min(lo8[i], other[i]); hi8[i] = std::max(hi8[i], other[i]); }
  }
}

void orc_sample_cube(const float* chain, int size, int levels, const float* dir3, float lod, float* rgba_out) {
  CubeTex t = {chain, size, levels};
  V4 c = cube_sample(t, V3{dir3[0], dir3[1], dir3[2]}, lod);
  rgba_out[0] = c.x; rgba_out[1] = c.y; rgba_out[2] = c.z; rgba_out[3] = c.w;
}

void orc_sample_2d_rgba8(const uint8_t* chain, int w, int h, int levels, int is_srgb, uint32_t sampler_index,
                         float u, float v, const float* ddx, const float* ddy, float* rgba_out) {
  init_lut();
  Tex8 t = {chain, w, h, levels, is_srgb != 0};
  V4 c = sample_2d(t, sampler_index, V2{u, v}, V2{ddx[0], ddx[1]}, V2{ddy[0], ddy[1]});
  rgba_out[0] = c.x; rgba_out[1] = c.y; rgba_out[2] = c.z; rgba_out[3] = c.w;
}

}  // extern "C"

// Oracle (test infrastructure): scalar f32 restatement of the IBL precompute shaders
// (src/shaders/equirectangular.wgsl, irradiancemap.wgsl, prefiltermap.wgsl, brdf.wgsl,
// mipmap.wgsl), the Radiance decoder in src/hdr.ts and the mip driver in
// src/renderer.ts:1209-1317.  Statement order follows the WGSL; every vector operation is
// spelled out component-wise so that no FMA contraction or reassociation takes place
// (the file is built with -ffp-contract=off).
#include "oracle.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

struct V3 { float x, y, z; };
inline V3 v3(float x, float y, float z) { return {x, y, z}; }
inline V3 add(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 mul(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline V3 normalize(V3 a) { float l = std::sqrt(dot(a, a)); return {a.x / l, a.y / l, a.z / l}; }

const float PI = 3.14159265359f;                 // prefiltermap.wgsl:1, irradiancemap.wgsl:1, brdf.wgsl:1
const float INV_ATAN_X = 0.1591f, INV_ATAN_Y = 0.3183f;   // equirectangular.wgsl:2

struct Face { V3 forward, up, right; };
// equirectangular.wgsl:20-57 (same table in irradiancemap.wgsl:20-57, prefiltermap.wgsl:77-114)
const Face FACES[6] = {
    {{1, 0, 0}, {0, 1, 0}, {0, 0, -1}},
    {{-1, 0, 0}, {0, 1, 0}, {0, 0, 1}},
    {{0, -1, 0}, {0, 0, 1}, {1, 0, 0}},
    {{0, 1, 0}, {0, 0, -1}, {1, 0, 0}},
    {{0, 0, 1}, {0, 1, 0}, {1, 0, 0}},
    {{0, 0, -1}, {0, 1, 0}, {-1, 0, 0}},
};

// equirectangular.wgsl:60-65: direction of destination texel gid (no half-texel offset).
inline V3 texel_direction(int gx, int gy, int gz, int size) {
  float dim = (float)size;
  float cu = (float)gx / dim * 2.0f - 1.0f;
  float cv = (float)gy / dim * 2.0f - 1.0f;
  const Face& f = FACES[gz];
  return normalize(add(add(f.forward, mul(f.right, cu)), mul(f.up, cv)));
}

// equirectangular.wgsl:68-69: truncating equirect pixel of a direction in a w x h level.
// textureLoad outside the level is implementation-defined in WGSL; it cannot happen for unit
// vectors (max eqUV is below 1) and is clamped here.  A NaN direction converts to pixel 0 (the
// float -> int conversion of NaN is implementation-defined in WGSL; 0 is what the gfx950 conversion
// gives): it occurs at the centre texel of the +-Y irradiance faces of an even-sized map, where
// irradiancemap.wgsl:69-70 normalises cross(up, N) = 0.
inline void equirect_pixel(V3 d, int w, int h, int* px, int* py) {
  float u = std::atan2(d.z, d.x) * INV_ATAN_X + 0.5f;
  float v = std::asin(d.y) * INV_ATAN_Y + 0.5f;
  int x = u == u ? (int)(u * (float)w) : 0, y = v == v ? (int)(v * (float)h) : 0;
  *px = std::min(std::max(x, 0), w - 1);
  *py = std::min(std::max(y, 0), h - 1);
}

inline int level_dim(int d, int level) { return std::max(1, d >> level); }

// prefiltermap.wgsl:17-28
float ggx_d(V3 n, V3 h, float roughness) {
  float a = roughness * roughness;
  float a2 = a * a;
  float ndoth = std::max(dot(n, h), 0.0f);
  float ndoth2 = ndoth * ndoth;
  float denom = ndoth2 * (a2 - 1.0f) + 1.0f;
  denom = PI * denom * denom;
  return a2 / denom;
}

// prefiltermap.wgsl:45-68, brdf.wgsl:22-45
V3 importance_sample_ggx(float xi_x, float xi_y, V3 n, float roughness) {
  float a = roughness * roughness;
  float phi = 2.0f * PI * xi_x;
  float cos_theta = std::sqrt((1.0f - xi_y) / (1.0f + (a * a - 1.0f) * xi_y));
  float sin_theta = std::sqrt(1.0f - cos_theta * cos_theta);
  V3 h = {std::cos(phi) * sin_theta, std::sin(phi) * sin_theta, cos_theta};
  V3 up = std::fabs(n.z) < 0.999f ? v3(0, 0, 1) : v3(1, 0, 0);
  V3 tangent = normalize(cross(up, n));
  V3 bitangent = cross(n, tangent);
  V3 s = add(add(mul(tangent, h.x), mul(bitangent, h.y)), mul(n, h.z));
  return normalize(s);
}

// brdf.wgsl:57-65 (k = r*r/2, not the frame shader's (r+1)^2/8)
float geometry_ggx_ibl(float cos_theta, float roughness) {
  float k = (roughness * roughness) / 2.0f;
  return cos_theta / (cos_theta * (1.0f - k) + k);
}

inline float srgb_to_linear(float c) {
  return c <= 0.04045f ? c / 12.92f : std::pow((c + 0.055f) / 1.055f, 2.4f);
}
inline float linear_to_srgb(float c) {
  return c <= 0.0031308f ? c * 12.92f : 1.055f * std::pow(c, 1.0f / 2.4f) - 0.055f;
}
inline uint8_t to_unorm8(float x) {
  if (!(x > 0.0f)) return 0;        // NaN and negatives store 0
  if (x > 1.0f) x = 1.0f;
  return (uint8_t)std::nearbyint(x * 255.0f);
}

// Store of a linear value into an 8-bit sRGB texel (the render-target write of generateMipmaps into an
// rgba8unorm-srgb level, renderer.ts:1263-1290): the EXACTLY rounded code.  WebGPU / Vulkan leave the encode to the
// implementation within a tolerance, and pow() of two math libraries differs by an ulp, which flips a byte now and then;
// so the code is defined without pow: code k is stored when T[k] <= v < T[k + 1], T[k] being the linear value whose
// encoding is k - 1/2, worked out in double precision and rounded up to the next float.  The float formula only
// proposes a candidate, the table decides.
inline const float* srgb_store_thresholds() {
  static float T[257];
  static bool ready = false;
  if (!ready) {
    T[0] = 0.0f;
    for (int k = 1; k <= 255; ++k) {
      const double c = ((double)k - 0.5) / 255.0;
      const double lin = c <= 0.04045 ? c / 12.92 : std::pow((c + 0.055) / 1.055, 2.4);
      float f = (float)lin;
      if ((double)f < lin) f = std::nextafterf(f, 2.0f);
      T[k] = f;
    }
    T[256] = 3.0e38f;
    ready = true;
  }
  return T;
}
inline uint8_t linear_to_srgb8(float v) {
  const float* T = srgb_store_thresholds();
  int k = (int)to_unorm8(linear_to_srgb(v));
  while (k < 255 && v >= T[k + 1]) ++k;
  while (k > 0 && !(v >= T[k])) --k;
  return (uint8_t)k;
}

}  // namespace

extern "C" {

int orc_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

// prefiltermap.wgsl:32-39
float orc_radical_inverse(uint32_t bits) {
  uint32_t v = (bits << 16) | (bits >> 16);
  v = ((v & 0x55555555u) << 1) | ((v & 0xAAAAAAAAu) >> 1);
  v = ((v & 0x33333333u) << 2) | ((v & 0xCCCCCCCCu) >> 2);
  v = ((v & 0x0F0F0F0Fu) << 4) | ((v & 0xF0F0F0F0u) >> 4);
  v = ((v & 0x00FF00FFu) << 8) | ((v & 0xFF00FF00u) >> 8);
  return (float)v * 2.3283064365386963e-10f;
}

void orc_irradiance_trip_counts(int* n_phi, int* n_theta) {
  int a = 0, b = 0;
  for (float phi = 0.0f; phi < 2.0f * PI; phi += 0.025f) ++a;
  for (float theta = 0.0f; theta < 0.5f * PI; theta += 0.025f) ++b;
  *n_phi = a; *n_theta = b;
}

uint16_t orc_f32_to_f16(float fl) {
  uint32_t x; std::memcpy(&x, &fl, 4);
  uint32_t sign = (x >> 16) & 0x8000u;
  uint32_t exp = (x >> 23) & 0xFFu, man = x & 0x7FFFFFu;
  if (exp == 0xFF) return (uint16_t)(sign | 0x7C00u | (man ? 0x200u | (man >> 13) : 0));
  int e = (int)exp - 127 + 15;
  if (e >= 31) return (uint16_t)(sign | 0x7C00u);
  if (e <= 0) {
    if (e < -10) return (uint16_t)sign;
    man |= 0x800000u;
    int shift = 14 - e;
    uint32_t half = man >> shift, rem = man & ((1u << shift) - 1), mid = 1u << (shift - 1);
    if (rem > mid || (rem == mid && (half & 1))) ++half;
    return (uint16_t)(sign | half);
  }
  uint32_t half = ((uint32_t)e << 10) | (man >> 13), rem = man & 0x1FFFu;
  if (rem > 0x1000u || (rem == 0x1000u && (half & 1))) ++half;   // carries into the exponent correctly
  return (uint16_t)(sign | half);
}

float orc_f16_to_f32(uint16_t h) {
  uint32_t sign = (uint32_t)(h & 0x8000u) << 16, exp = (h >> 10) & 0x1Fu, man = h & 0x3FFu, x;
  if (exp == 0) {
    if (man == 0) x = sign;
    else {
      int e = -1;
      do { ++e; man <<= 1; } while (!(man & 0x400u));
      x = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3FFu) << 13);
    }
  } else if (exp == 31) x = sign | 0x7F800000u | (man << 13);
  else x = sign | ((exp - 15 + 127) << 23) | (man << 13);
  float fl; std::memcpy(&fl, &x, 4); return fl;
}

/* ---------------- hdr.ts:37-118 ---------------- */
static int hdr_header(const uint8_t* b, int64_t n, int64_t* index_out, int* w, int* h) {
  std::string header;
  int64_t index = 0;
  while (index < n) {
    char c = (char)b[index++];
    header.push_back(c);
    if (c == '\n' && header.find("-Y ") != std::string::npos && header.find(" +X ") != std::string::npos) break;
  }
  if (header.find("32-bit_rle_rgbe") == std::string::npos && header.find("32-bit_rle_xyze") == std::string::npos)
    return -1;   // "HDRLoader: Unsupported .hdr format."
  // /-Y (\d+) \+X (\d+)/
  size_t p = 0;
  bool found = false;
  long hh = 0, ww = 0;
  while ((p = header.find("-Y ", p)) != std::string::npos) {
    size_t q = p + 3, q0 = q;
    while (q < header.size() && std::isdigit((unsigned char)header[q])) ++q;
    if (q > q0 && header.compare(q, 4, " +X ") == 0) {
      size_t r = q + 4, r0 = r;
      while (r < header.size() && std::isdigit((unsigned char)header[r])) ++r;
      if (r > r0) {
        hh = std::stol(header.substr(q0, q - q0));
        ww = std::stol(header.substr(r0, r - r0));
        found = true;
        break;
      }
    }
    p += 3;
  }
  if (!found) return -2;   // "HDRLoader: No resolution found."
  *w = (int)ww; *h = (int)hh; *index_out = index;
  return 0;
}

int orc_hdr_dims(const uint8_t* bytes, int64_t n, int* w, int* h) {
  int64_t idx;
  return hdr_header(bytes, n, &idx, w, h);
}

int orc_hdr_decode(const uint8_t* bytes, int64_t n, float* out) {
  int w, h; int64_t index;
  int rc = hdr_header(bytes, n, &index, &w, &h);
  if (rc) return rc;
  const int64_t total = (int64_t)w * h * 4;
  int64_t cur = 0;
  std::vector<uint8_t> scan((size_t)w * 4);
  auto at = [&](int64_t i) -> int { return i < n ? bytes[i] : 0; };   // reads past the end give undefined → 0 in JS arithmetic
  while (index < n) {
    int h0 = at(index), h1 = at(index + 1), h2 = at(index + 2), h3 = at(index + 3);
    index += 4;
    if (h0 != 2 || h1 != 2 || ((h2 << 8) + h3) != w) return -3;   // "HDRLoader: Only support RLE format."
    std::fill(scan.begin(), scan.end(), 0);
    for (int i = 0; i < 4; ++i) {
      int pos = 0;
      while (pos < w) {
        if (index >= n) { pos = w; break; }
        int count = bytes[index++];
        if (count > 128) {
          int run = count - 128, value = at(index++);
          for (int j = 0; j < run; ++j) { if (pos < w) scan[i + pos * 4] = (uint8_t)value; ++pos; }
        } else {
          if (count == 0) { pos = w; break; }   // a zero count would never advance in the reference
          for (int j = 0; j < count; ++j) { int v = at(index++); if (pos < w) scan[i + pos * 4] = (uint8_t)v; ++pos; }
        }
      }
    }
    for (int x = 0; x < w; ++x) {
      int r = scan[x * 4], g = scan[x * 4 + 1], b = scan[x * 4 + 2], e = scan[x * 4 + 3];
      double scale = std::pow(2.0, e - 128) / 255.0;
      if (cur + 4 <= total) {
        out[cur] = (float)(r * scale); out[cur + 1] = (float)(g * scale); out[cur + 2] = (float)(b * scale); out[cur + 3] = 1.0f;
      }
      cur += 4;
    }
  }
  return 0;
}

/* ---------------- mip chains ---------------- */
int orc_mip_level_count(int w, int h) {   // utils.ts:20-24
  int m = std::max(w, h);
  return (int)(1 + std::log2((double)m));
}
int64_t orc_chain_offset(int w, int h, int level) {
  int64_t off = 0;
  for (int l = 0; l < level; ++l) off += (int64_t)level_dim(w, l) * level_dim(h, l);
  return off;
}
int64_t orc_chain_texels(int w, int h, int levels) { return orc_chain_offset(w, h, levels); }

// One bilinear tap at the destination texel centre with the mip sampler of renderer.ts:1116-1118
// (minFilter linear, clamp-to-edge): texel footprint and weights for one axis.
static inline void mip_axis(int dst_i, int dst_n, int src_n, int* i0, int* i1, float* a) {
  float u = ((float)dst_i + 0.5f) / (float)dst_n;
  float t = u * (float)src_n - 0.5f;
  float fl = std::floor(t);
  *a = t - fl;
  int i = (int)fl;
  *i0 = std::min(std::max(i, 0), src_n - 1);
  *i1 = std::min(std::max(i + 1, 0), src_n - 1);
}

void orc_build_mips_rgba32f(float* chain, int w, int h, int levels) {
  for (int l = 1; l < levels; ++l) {
    int sw = level_dim(w, l - 1), sh = level_dim(h, l - 1), dw = level_dim(w, l), dh = level_dim(h, l);
    const float* src = chain + 4 * orc_chain_offset(w, h, l - 1);
    float* dst = chain + 4 * orc_chain_offset(w, h, l);
#pragma omp parallel for schedule(static)
    for (int y = 0; y < dh; ++y) {
      int y0, y1; float b; mip_axis(y, dh, sh, &y0, &y1, &b);
      for (int x = 0; x < dw; ++x) {
        int x0, x1; float a; mip_axis(x, dw, sw, &x0, &x1, &a);
        float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
        for (int c = 0; c < 4; ++c) {
          float t00 = src[4 * ((int64_t)y0 * sw + x0) + c], t10 = src[4 * ((int64_t)y0 * sw + x1) + c];
          float t01 = src[4 * ((int64_t)y1 * sw + x0) + c], t11 = src[4 * ((int64_t)y1 * sw + x1) + c];
          dst[4 * ((int64_t)y * dw + x) + c] = ((t00 * w00 + t10 * w10) + t01 * w01) + t11 * w11;
        }
      }
    }
  }
}

void orc_build_mips_rgba8(uint8_t* chain, int w, int h, int levels, int is_srgb) {
  srgb_store_thresholds();   // built before the parallel loops
  float lut[256];
  for (int i = 0; i < 256; ++i) { float c = (float)i / 255.0f; lut[i] = is_srgb ? srgb_to_linear(c) : c; }
  for (int l = 1; l < levels; ++l) {
    int sw = level_dim(w, l - 1), sh = level_dim(h, l - 1), dw = level_dim(w, l), dh = level_dim(h, l);
    const uint8_t* src = chain + 4 * orc_chain_offset(w, h, l - 1);
    uint8_t* dst = chain + 4 * orc_chain_offset(w, h, l);
#pragma omp parallel for schedule(static)
    for (int y = 0; y < dh; ++y) {
      int y0, y1; float b; mip_axis(y, dh, sh, &y0, &y1, &b);
      for (int x = 0; x < dw; ++x) {
        int x0, x1; float a; mip_axis(x, dw, sw, &x0, &x1, &a);
        float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
        for (int c = 0; c < 4; ++c) {
          const uint8_t s00 = src[4 * ((int64_t)y0 * sw + x0) + c], s10 = src[4 * ((int64_t)y0 * sw + x1) + c];
          const uint8_t s01 = src[4 * ((int64_t)y1 * sw + x0) + c], s11 = src[4 * ((int64_t)y1 * sw + x1) + c];
          float v;
          if (c == 3 || !is_srgb) {   // alpha is never sRGB-encoded
            v = (((float)s00 / 255.0f * w00 + (float)s10 / 255.0f * w10) + (float)s01 / 255.0f * w01) + (float)s11 / 255.0f * w11;
          } else {
            v = ((lut[s00] * w00 + lut[s10] * w10) + lut[s01] * w01) + lut[s11] * w11;
            dst[4 * ((int64_t)y * dw + x) + c] = linear_to_srgb8(v);
            continue;
          }
          dst[4 * ((int64_t)y * dw + x) + c] = to_unorm8(v);
        }
      }
    }
  }
}

/* ---------------- equirectangular.wgsl:13-75 ---------------- */
void orc_equirect_to_cube(const float* hdr, int w, int h, int size, float* out, int64_t first, int64_t count) {
  const int64_t total = 6LL * size * size;
  if (count < 0) count = total - first;
#pragma omp parallel for schedule(static)
  for (int64_t t = first; t < first + count; ++t) {
    int gz = (int)(t / ((int64_t)size * size)), gy = (int)((t / size) % size), gx = (int)(t % size);
    V3 d = texel_direction(gx, gy, gz, size);
    int px, py; equirect_pixel(d, w, h, &px, &py);
    std::memcpy(out + 4 * t, hdr + 4 * ((int64_t)py * w + px), 4 * sizeof(float));
  }
}

/* ---------------- irradiancemap.wgsl:13-91 ---------------- */
void orc_irradiance(const float* hdr, int w, int h, int size, float* out, int64_t first, int64_t count) {
  const int64_t total = 6LL * size * size;
  if (count < 0) count = total - first;
#pragma omp parallel for schedule(dynamic, 8)
  for (int64_t t = first; t < first + count; ++t) {
    int gz = (int)(t / ((int64_t)size * size)), gy = (int)((t / size) % size), gx = (int)(t % size);
    V3 n = texel_direction(gx, gy, gz, size);
    V3 up = {0, 1, 0};
    V3 right = normalize(cross(up, n));
    up = normalize(cross(n, right));
    float ir = 0, ig = 0, ib = 0;
    const float delta = 0.025f;
    int samples = 0;
    for (float phi = 0.0f; phi < 2.0f * PI; phi += delta) {
      for (float theta = 0.0f; theta < 0.5f * PI; theta += delta) {
        V3 ts = {std::sin(theta) * std::cos(phi), std::sin(theta) * std::sin(phi), std::cos(theta)};
        V3 sv = add(add(mul(right, ts.x), mul(up, ts.y)), mul(n, ts.z));
        int px, py; equirect_pixel(sv, w, h, &px, &py);
        const float* texel = hdr + 4 * ((int64_t)py * w + px);
        float ct = std::cos(theta), st = std::sin(theta);
        ir += texel[0] * ct * st; ig += texel[1] * ct * st; ib += texel[2] * ct * st;
        ++samples;
      }
    }
    float inv = 1.0f / (float)samples;
    out[4 * t + 0] = PI * ir * inv; out[4 * t + 1] = PI * ig * inv; out[4 * t + 2] = PI * ib * inv; out[4 * t + 3] = 1.0f;
  }
}

/* ---------------- prefiltermap.wgsl:70-171 ---------------- */
void orc_prefilter(const float* chain, int w, int h, int levels, int size, float roughness, float* out,
                   int64_t first, int64_t count) {
  const int64_t total = 6LL * size * size;
  if (count < 0) count = total - first;
  const uint32_t SAMPLE_COUNT = 1024u;
  std::vector<int64_t> level_off(levels);
  for (int l = 0; l < levels; ++l) level_off[l] = orc_chain_offset(w, h, l);
#pragma omp parallel for schedule(dynamic, 16)
  for (int64_t t = first; t < first + count; ++t) {
    int gz = (int)(t / ((int64_t)size * size)), gy = (int)((t / size) % size), gx = (int)(t % size);
    V3 N = texel_direction(gx, gy, gz, size), R = N, V = R;
    float cr = 0, cg = 0, cb = 0, total_weight = 0;
    for (uint32_t i = 0; i < SAMPLE_COUNT; ++i) {
      float xi_x = (float)i / (float)SAMPLE_COUNT, xi_y = orc_radical_inverse(i);
      V3 H = importance_sample_ggx(xi_x, xi_y, N, roughness);
      V3 L = normalize(sub(mul(H, 2.0f * dot(V, H)), V));
      float ndotl = std::max(dot(N, L), 0.0f);
      if (ndotl > 0.0f) {
        float D = ggx_d(N, H, roughness);
        float ndoth = std::max(dot(N, H), 0.0f), hdotv = std::max(dot(H, V), 0.0f);
        float pdf = D * ndoth / (4.0f * hdotv) + 0.0001f;
        float resolution = 512.0f;
        float sa_texel = 4.0f * PI / (6.0f * resolution * resolution);
        float sa_sample = 1.0f / ((float)SAMPLE_COUNT * pdf + 0.0001f);
        float mip = roughness == 0.0f ? 0.0f : 0.5f * std::log2(sa_sample / sa_texel);
        int m0 = (int)std::floor(mip), m1 = m0 + 1;
        float mix = mip - (float)m0;
        // A level outside the chain is implementation-defined for textureLoad; clamp to the chain.
        int l0 = std::min(std::max(m0, 0), levels - 1), l1 = std::min(std::max(m1, 0), levels - 1);
        int px, py;
        equirect_pixel(L, level_dim(w, l0), level_dim(h, l0), &px, &py);
        const float* t0 = chain + 4 * (level_off[l0] + (int64_t)py * level_dim(w, l0) + px);
        equirect_pixel(L, level_dim(w, l1), level_dim(h, l1), &px, &py);
        const float* t1 = chain + 4 * (level_off[l1] + (int64_t)py * level_dim(w, l1) + px);
        float a0 = t0[0] * ndotl, a1 = t0[1] * ndotl, a2 = t0[2] * ndotl;
        float b0 = t1[0] * ndotl, b1 = t1[1] * ndotl, b2 = t1[2] * ndotl;
        cr += a0 * (1.0f - mix) + b0 * mix; cg += a1 * (1.0f - mix) + b1 * mix; cb += a2 * (1.0f - mix) + b2 * mix;
        total_weight += ndotl;
      }
    }
    out[4 * t + 0] = cr / total_weight; out[4 * t + 1] = cg / total_weight; out[4 * t + 2] = cb / total_weight; out[4 * t + 3] = 1.0f;
  }
}

/* ---------------- test instrumentation: which results are reachable -------------------------------------------------
 * The three cube generators read NEAREST equirect pixels (textureLoad of a truncated coordinate).  A sample whose
 * coordinate lies within rounding distance of a pixel boundary legitimately lands on either side on another
 * implementation (atan2 / asin / the direction itself differ in the last bits), and next to the 5 000-nit sun one such
 * flip moves a texel far beyond any relative tolerance.  These variants compute, per output texel, the interval
 * [lo, hi] of values reachable when every sample within `ang` radians of a boundary may take either pixel.  They are
 * not part of the restated pipeline; the parity tests use them to show that a texel outside the tolerance is such a
 * texel - and by how much the interval had to be used. */
struct PixelRange { int x0, x1, y0, y1; };
inline PixelRange equirect_pixel_range(V3 d, int w, int h, float ang) {
  int px, py; equirect_pixel(d, w, h, &px, &py);
  PixelRange r = {px, px, py, py};
  if (!(d.x == d.x) || !(d.y == d.y) || !(d.z == d.z)) return r;   // NaN direction: pixel 0 by rule
  const float rxz = std::sqrt(d.x * d.x + d.z * d.z), ry = std::sqrt(std::max(1.0f - d.y * d.y, 0.0f));
  const float su = (ang / std::max(rxz, 1e-20f) + ang) * INV_ATAN_X + 1.2e-7f, sv = (ang / std::max(ry, 1e-20f) + ang) * INV_ATAN_Y + 1.2e-7f;
  const float u = std::atan2(d.z, d.x) * INV_ATAN_X + 0.5f, v = std::asin(std::min(std::max(d.y, -1.0f), 1.0f)) * INV_ATAN_Y + 0.5f;
  auto pixel = [](float coord, int n) { return (int)(std::min(std::max(coord, 0.0f), 1.0f) * (float)n) >= n ? n - 1 : (int)(std::min(std::max(coord, 0.0f), 1.0f) * (float)n); };
  if (!(u - su >= 0.0f) || !(u + su <= 0.99983f)) { r.x0 = 0; r.x1 = w - 1; }    // the seam (or a pole): the angle may wrap to the other end of the row
  else { r.x0 = pixel(u - su, w); r.x1 = pixel(u + su, w); }
  r.y0 = pixel(v - sv, h); r.y1 = pixel(v + sv, h);
  r.x0 = std::min(r.x0, px); r.x1 = std::max(r.x1, px); r.y0 = std::min(r.y0, py); r.y1 = std::max(r.y1, py);   // the nominal pixel is always a candidate
  return r;
}
inline void range_min_max(const float* level, int w, const PixelRange& r, float* mn, float* mx) {
  for (int c = 0; c < 3; ++c) { mn[c] = 1e30f; mx[c] = -1e30f; }
  for (int y = r.y0; y <= r.y1; ++y)
    for (int x = r.x0; x <= r.x1; ++x) {
      const float* t = level + 4 * ((int64_t)y * w + x);
      for (int c = 0; c < 3; ++c) { mn[c] = std::min(mn[c], t[c]); mx[c] = std::max(mx[c], t[c]); }
    }
}

void orc_equirect_to_cube_bounds(const float* hdr, int w, int h, int size, float ang, float* lo, float* hi, int64_t first, int64_t count) {
  if (count < 0) count = 6LL * size * size - first;
#pragma omp parallel for schedule(static)
  for (int64_t t = first; t < first + count; ++t) {
    int gz = (int)(t / ((int64_t)size * size)), gy = (int)((t / size) % size), gx = (int)(t % size);
    float mn[3], mx[3];
    range_min_max(hdr, w, equirect_pixel_range(texel_direction(gx, gy, gz, size), w, h, ang), mn, mx);
    for (int c = 0; c < 3; ++c) { lo[4 * t + c] = mn[c]; hi[4 * t + c] = mx[c]; }
    lo[4 * t + 3] = hi[4 * t + 3] = 1.0f;
  }
}

void orc_irradiance_bounds(const float* hdr, int w, int h, int size, float ang, float* lo, float* hi, int64_t first, int64_t count) {
  if (count < 0) count = 6LL * size * size - first;
#pragma omp parallel for schedule(dynamic, 8)
  for (int64_t t = first; t < first + count; ++t) {
    int gz = (int)(t / ((int64_t)size * size)), gy = (int)((t / size) % size), gx = (int)(t % size);
    V3 n = texel_direction(gx, gy, gz, size);
    V3 up = {0, 1, 0};
    V3 right = normalize(cross(up, n));
    up = normalize(cross(n, right));
    double sl[3] = {0, 0, 0}, sh[3] = {0, 0, 0};
    const float delta = 0.025f;
    int samples = 0;
    for (float phi = 0.0f; phi < 2.0f * PI; phi += delta)
      for (float theta = 0.0f; theta < 0.5f * PI; theta += delta) {
        V3 ts = {std::sin(theta) * std::cos(phi), std::sin(theta) * std::sin(phi), std::cos(theta)};
        V3 sv = add(add(mul(right, ts.x), mul(up, ts.y)), mul(n, ts.z));
        float mn[3], mx[3];
        range_min_max(hdr, w, equirect_pixel_range(sv, w, h, ang), mn, mx);
        const float wgt = std::cos(theta) * std::sin(theta);
        for (int c = 0; c < 3; ++c) { sl[c] += (double)mn[c] * wgt; sh[c] += (double)mx[c] * wgt; }
        ++samples;
      }
    for (int c = 0; c < 3; ++c) { lo[4 * t + c] = (float)(PI * sl[c] / samples); hi[4 * t + c] = (float)(PI * sh[c] / samples); }
    lo[4 * t + 3] = hi[4 * t + 3] = 1.0f;
  }
}

void orc_prefilter_bounds(const float* chain, int w, int h, int levels, int size, float roughness, float ang, float* lo, float* hi,
                          int64_t first, int64_t count) {
  if (count < 0) count = 6LL * size * size - first;
  const uint32_t SAMPLE_COUNT = 1024u;
  std::vector<int64_t> level_off(levels);
  for (int l = 0; l < levels; ++l) level_off[l] = orc_chain_offset(w, h, l);
#pragma omp parallel for schedule(dynamic, 16)
  for (int64_t t = first; t < first + count; ++t) {
    int gz = (int)(t / ((int64_t)size * size)), gy = (int)((t / size) % size), gx = (int)(t % size);
    V3 N = texel_direction(gx, gy, gz, size), R = N, V = R;
    double sl[3] = {0, 0, 0}, sh[3] = {0, 0, 0}, total_weight = 0;
    for (uint32_t i = 0; i < SAMPLE_COUNT; ++i) {
      float xi_x = (float)i / (float)SAMPLE_COUNT, xi_y = orc_radical_inverse(i);
      V3 H = importance_sample_ggx(xi_x, xi_y, N, roughness);
      V3 L = normalize(sub(mul(H, 2.0f * dot(V, H)), V));
      float ndotl = std::max(dot(N, L), 0.0f);
      if (!(ndotl > 0.0f)) continue;
      float D = ggx_d(N, H, roughness);
      float ndoth = std::max(dot(N, H), 0.0f), hdotv = std::max(dot(H, V), 0.0f);
      float pdf = D * ndoth / (4.0f * hdotv) + 0.0001f;
      float sa_texel = 4.0f * PI / (6.0f * 512.0f * 512.0f);
      float sa_sample = 1.0f / ((float)SAMPLE_COUNT * pdf + 0.0001f);
      float mip = roughness == 0.0f ? 0.0f : 0.5f * std::log2(sa_sample / sa_texel);
      int m0 = (int)std::floor(mip);
      float mix = mip - (float)m0;
      int l0 = std::min(std::max(m0, 0), levels - 1), l1 = std::min(std::max(m0 + 1, 0), levels - 1);
      float mn0[3], mx0[3], mn1[3], mx1[3];
      range_min_max(chain + 4 * level_off[l0], level_dim(w, l0), equirect_pixel_range(L, level_dim(w, l0), level_dim(h, l0), ang), mn0, mx0);
      range_min_max(chain + 4 * level_off[l1], level_dim(w, l1), equirect_pixel_range(L, level_dim(w, l1), level_dim(h, l1), ang), mn1, mx1);
      for (int c = 0; c < 3; ++c) {
        sl[c] += (double)ndotl * ((double)mn0[c] * (1.0 - mix) + (double)mn1[c] * mix);
        sh[c] += (double)ndotl * ((double)mx0[c] * (1.0 - mix) + (double)mx1[c] * mix);
      }
      total_weight += ndotl;
    }
    for (int c = 0; c < 3; ++c) { lo[4 * t + c] = (float)(sl[c] / total_weight); hi[4 * t + c] = (float)(sh[c] / total_weight); }
    lo[4 * t + 3] = hi[4 * t + 3] = 1.0f;
  }
}

/* ---------------- brdf.wgsl:67-106 ---------------- */
void orc_brdf_lut(int size, float* out_f32, uint16_t* out_f16, int row0, int rows) {
  const uint32_t SAMPLE_COUNT = 1024u;
  const float fsize = (float)size;
  if (rows < 0) rows = size - row0;
#pragma omp parallel for schedule(dynamic, 1)
  for (int y = row0; y < row0 + rows; ++y) {
    for (int x = 0; x < size; ++x) {
      float ndotv = (float)x / (fsize - 1.0f);
      float roughness = (fsize - (float)y) / (fsize - 1.0f);
      V3 V = {std::sqrt(1.0f - ndotv * ndotv), 0.0f, ndotv};
      float A = 0.0f, B = 0.0f;
      V3 N = {0, 0, 1};
      for (uint32_t i = 0; i < SAMPLE_COUNT; ++i) {
        float xi_x = (float)i / (float)SAMPLE_COUNT, xi_y = orc_radical_inverse(i);
        V3 H = importance_sample_ggx(xi_x, xi_y, N, roughness);
        V3 L = normalize(sub(mul(H, 2.0f * dot(V, H)), V));
        float ndotl = std::max(L.z, 0.0f), ndoth = std::max(H.z, 0.0f), vdoth = std::max(dot(V, H), 0.0f);
        if (ndotl > 0.0f) {
          float g1 = geometry_ggx_ibl(std::max(dot(N, V), 0.0f), roughness);
          float g2 = geometry_ggx_ibl(std::max(dot(N, L), 0.0f), roughness);
          float G = g1 * g2;
          float g_vis = (G * vdoth) / (ndoth * ndotv);
          float fc = std::pow(1.0f - vdoth, 5.0f);
          A += (1.0f - fc) * g_vis;
          B += fc * g_vis;
        }
      }
      A /= (float)SAMPLE_COUNT; B /= (float)SAMPLE_COUNT;
      int64_t o = 4 * ((int64_t)(y - row0) * size + x);
      if (out_f32) { out_f32[o] = A; out_f32[o + 1] = B; out_f32[o + 2] = 0.0f; out_f32[o + 3] = 1.0f; }
      if (out_f16) {
        out_f16[o] = orc_f32_to_f16(A); out_f16[o + 1] = orc_f32_to_f16(B);
        out_f16[o + 2] = orc_f32_to_f16(0.0f); out_f16[o + 3] = orc_f32_to_f16(1.0f);
      }
    }
  }
}

}  // extern "C"

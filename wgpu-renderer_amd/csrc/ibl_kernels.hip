// gfx950 kernels for the IBL precompute: mip chains, equirect -> cube, irradiance convolution,
// GGX prefilter and the split-sum BRDF table.  They produce what the reference's compute shaders
// produce (src/shaders/equirectangular.wgsl, irradiancemap.wgsl, prefiltermap.wgsl, brdf.wgsl,
// mipmap.wgsl) but are organised for 64-wide wavefronts: everything that depends only on the
// sample index is hoisted into tables, sample loops are split across lanes and reduced with
// cross-lane shuffles, and texel gathers are 16-byte loads.
#include "pbr_internal.h"
#include "devmath.h"
#include "equirect_math.h"
#include <type_traits>

namespace {

constexpr float PI_F = 3.14159265359f;
constexpr float INV_ATAN_X = 0.1591f, INV_ATAN_Y = 0.3183f;   // equirectangular.wgsl:2

// Face bases of equirectangular.wgsl:20-57: forward, up, right.
__constant__ float FACE_TABLE[6][9] = {
    {1, 0, 0, 0, 1, 0, 0, 0, -1},  {-1, 0, 0, 0, 1, 0, 0, 0, 1}, {0, -1, 0, 0, 0, 1, 1, 0, 0},
    {0, 1, 0, 0, 0, -1, 1, 0, 0},  {0, 0, 1, 0, 1, 0, 1, 0, 0},  {0, 0, -1, 0, 1, 0, -1, 0, 0},
};

// Direction of destination texel (gx, gy) of face gz; the reference applies no half-texel offset
// (equirectangular.wgsl:60-65).
PBR_DEV V3 texel_direction(int gx, int gy, int gz, int size) {
  float dim = (float)size;
  float cu = (float)gx / dim * 2.0f - 1.0f;
  float cv = (float)gy / dim * 2.0f - 1.0f;
  const float* f = FACE_TABLE[gz];
  V3 d = v3(f[0] + f[6] * cu + f[3] * cv, f[1] + f[7] * cu + f[4] * cv, f[2] + f[8] * cu + f[5] * cv);
  return normalize(d);
}

// Truncating equirect texel index of direction d in a w x h level (equirectangular.wgsl:68-69).
PBR_DEV uint32_t equirect_index(V3 d, int w, int h) {
  float u = atan2f(d.z, d.x) * INV_ATAN_X + 0.5f;
  float v = asinf(fminf(fmaxf(d.y, -1.0f), 1.0f)) * INV_ATAN_Y + 0.5f;
  int x = min(max((int)(u * (float)w), 0), w - 1);
  int y = min(max((int)(v * (float)h), 0), h - 1);
  return (uint32_t)y * (uint32_t)w + (uint32_t)x;
}

PBR_DEV float srgb_to_linear(float c) { return c <= 0.04045f ? c / 12.92f : powf((c + 0.055f) / 1.055f, 2.4f); }
PBR_DEV float linear_to_srgb(float c) { return c <= 0.0031308f ? c * 12.92f : 1.055f * powf(c, 1.0f / 2.4f) - 0.055f; }

// Footprint of the one bilinear tap mipmap.wgsl takes at a destination texel centre (clamp-to-edge).
// (every product and sum of the mip filters is rounded on its own, as written: for level sizes that do not halve
// exactly the weights are no powers of two, and a fused multiply-add would move a byte on a rounding boundary)
PBR_DEV void mip_axis(int dst_i, int dst_n, int src_n, int& i0, int& i1, float& a) {
#pragma clang fp contract(off)
  float u = ((float)dst_i + 0.5f) / (float)dst_n;
  float t = u * (float)src_n - 0.5f;
  float fl = floorf(t);
  a = t - fl;
  int i = (int)fl;
  i0 = min(max(i, 0), src_n - 1);
  i1 = min(max(i + 1, 0), src_n - 1);
}

__global__ void __launch_bounds__(256) mip_rgba32f_kernel(const float4* __restrict__ src, float4* __restrict__ dst, int sw, int sh, int dw, int dh) {
#pragma clang fp contract(off)
  int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= dw || y >= dh) return;
  int x0, x1, y0, y1; float a, b;
  mip_axis(x, dw, sw, x0, x1, a);
  mip_axis(y, dh, sh, y0, y1, b);
  float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
  float4 t00 = src[(size_t)y0 * sw + x0], t10 = src[(size_t)y0 * sw + x1], t01 = src[(size_t)y1 * sw + x0], t11 = src[(size_t)y1 * sw + x1];
  float4 o;
  o.x = ((t00.x * w00 + t10.x * w10) + t01.x * w01) + t11.x * w11;
  o.y = ((t00.y * w00 + t10.y * w10) + t01.y * w01) + t11.y * w11;
  o.z = ((t00.z * w00 + t10.z * w10) + t01.z * w01) + t11.z * w11;
  o.w = ((t00.w * w00 + t10.w * w10) + t01.w * w01) + t11.w * w11;
  dst[(size_t)y * dw + x] = o;
}

__global__ void __launch_bounds__(256) mip_rgba8_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, int sw, int sh, int dw, int dh, int srgb,
                                                        const float* __restrict__ srgb_tables) {
#pragma clang fp contract(off)
  // decode table and store thresholds of the 8-bit sRGB codes (host-made: the same numbers on every implementation)
  __shared__ float s_dec[256], s_thr[257];
  for (int i = threadIdx.x; i < 256; i += 256) s_dec[i] = srgb_tables[i];
  for (int i = threadIdx.x; i < 257; i += 256) s_thr[i] = srgb_tables[256 + i];
  __syncthreads();
  int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= dw || y >= dh) return;
  int x0, x1, y0, y1; float a, b;
  mip_axis(x, dw, sw, x0, x1, a);
  mip_axis(y, dh, sh, y0, y1, b);
  float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
  uint32_t t00 = src[tex8_block_offset(x0, y0, sw)], t10 = src[tex8_block_offset(x1, y0, sw)], t01 = src[tex8_block_offset(x0, y1, sw)],
           t11 = src[tex8_block_offset(x1, y1, sw)];
  uint32_t out = 0;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const uint32_t b00 = (t00 >> (8 * c)) & 255u, b10 = (t10 >> (8 * c)) & 255u, b01 = (t01 >> (8 * c)) & 255u, b11 = (t11 >> (8 * c)) & 255u;
    const bool enc = srgb && c < 3;   // the average is taken in linear light and re-encoded at every level
    if (enc) {
      const float v = ((s_dec[b00] * w00 + s_dec[b10] * w10) + s_dec[b01] * w01) + s_dec[b11] * w11;
      // exactly rounded store: the formula proposes a code, the thresholds decide (code k for thr[k] <= v < thr[k + 1])
      int k = (int)to_unorm8(linear_to_srgb(v));
      while (k < 255 && v >= s_thr[k + 1]) ++k;
      while (k > 0 && !(v >= s_thr[k])) --k;
      out |= (uint32_t)k << (8 * c);
    } else {
      const float v = (((float)b00 / 255.0f * w00 + (float)b10 / 255.0f * w10) + (float)b01 / 255.0f * w01) + (float)b11 / 255.0f * w11;
      out |= to_unorm8(v) << (8 * c);
    }
  }
  dst[tex8_block_offset(x, y, dw)] = out;
}

// Row-major image -> the blocked layout of a texture level.
__global__ void __launch_bounds__(256) block_rgba8_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, int w, int h) {
  int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= w || y >= h) return;
  dst[tex8_block_offset(x, y, w)] = src[(size_t)y * w + x];
}

__global__ void __launch_bounds__(256) drop_alpha_kernel(const float4* __restrict__ rgba, float* __restrict__ rgb, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) { const float4 t = rgba[i]; rgb[3 * i] = t.x; rgb[3 * i + 1] = t.y; rgb[3 * i + 2] = t.z; }
}

__global__ void __launch_bounds__(256) equirect_to_cube_kernel(DevEquirect hdr, float4* __restrict__ out, int size) {
  int gx = blockIdx.x * 64 + (threadIdx.x & 63), gy = blockIdx.y * 4 + (threadIdx.x >> 6), gz = blockIdx.z;
  if (gx >= size || gy >= size) return;
  V3 d = texel_direction(gx, gy, gz, size);
  out[((size_t)gz * size + gy) * size + gx] = hdr.base[equirect_index(d, hdr.w, hdr.h)];
}

// One workgroup per output texel: thread p owns azimuth step p and walks the polar steps; the
// partial sums are reduced across the wave by shuffles and across the four waves through LDS.
// tables = cos(phi)[n_phi], sin(phi)[n_phi], sin(theta)[n_theta], cos(theta)[n_theta] for the angles
// the reference's f32 loop counters take (irradiancemap.wgsl:75-76).
__global__ void __launch_bounds__(256) irradiance_kernel(DevEquirect hdr, const float* __restrict__ tables, int n_phi, int n_theta,
                                                         float4* __restrict__ out, int size) {
  __shared__ float s_sin_t[128], s_cos_t[128];
  __shared__ float s_part[4][3];
  const int t = blockIdx.x;
  const int gz = t / (size * size), gy = (t / size) % size, gx = t % size;
  for (int i = threadIdx.x; i < n_theta; i += 256) { s_sin_t[i] = tables[2 * n_phi + i]; s_cos_t[i] = tables[2 * n_phi + n_theta + i]; }
  __syncthreads();
  V3 n = texel_direction(gx, gy, gz, size);
  V3 right = normalize(cross(v3(0, 1, 0), n));
  V3 up = normalize(cross(n, right));
  float r = 0, g = 0, b = 0;
  for (int p = threadIdx.x; p < n_phi; p += 256) {
    const float cp = tables[p], sp = tables[n_phi + p];
    for (int q = 0; q < n_theta; ++q) {
      const float st = s_sin_t[q], ct = s_cos_t[q];
      V3 sv = right * (st * cp) + up * (st * sp) + n * ct;
      float4 tex = hdr.base[equirect_index(sv, hdr.w, hdr.h)];
      r += tex.x * ct * st; g += tex.y * ct * st; b += tex.z * ct * st;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { r += __shfl_xor(r, o); g += __shfl_xor(g, o); b += __shfl_xor(b, o); }
  if ((threadIdx.x & 63) == 0) { s_part[threadIdx.x >> 6][0] = r; s_part[threadIdx.x >> 6][1] = g; s_part[threadIdx.x >> 6][2] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float inv = 1.0f / (float)(n_phi * n_theta);
    float sr = (s_part[0][0] + s_part[1][0]) + (s_part[2][0] + s_part[3][0]);
    float sg = (s_part[0][1] + s_part[1][1]) + (s_part[2][1] + s_part[3][1]);
    float sb = (s_part[0][2] + s_part[1][2]) + (s_part[2][2] + s_part[3][2]);
    out[t] = make_float4(PI_F * sr * inv, PI_F * sg * inv, PI_F * sb * inv, 1.0f);
  }
}

// Light direction of one tabulated sample in world space: three multiply-add chains.
PBR_DEV V3 sample_to_world(const V3& T, const V3& B, const V3& N, float lx, float ly, float lz) {
#pragma clang fp contract(fast)
  return v3(fmaf(lz, N.x, fmaf(ly, B.x, T.x * lx)), fmaf(lz, N.y, fmaf(ly, B.y, T.y * lx)), fmaf(lz, N.z, fmaf(ly, B.z, T.z * lx)));
}

// GGX prefilter of one cube level.  SPLIT lanes share a texel and stride the sample table; the
// table holds, per kept sample, the light direction in the texel's tangent frame, the two source
// levels and their blend — all independent of the texel because N = V (prefiltermap.wgsl:124-126).
//
// The loop is bound by vector-instruction issue, and on this chip an f32 add / multiply / fma whose operands are all
// vector registers or constants issues at twice the rate of one that reads a scalar register.  With one texel per
// lane (SPLIT == 1) the sample is the same in every lane and would arrive in scalar registers; its floating-point
// fields are therefore staged in LDS and read back as a broadcast into vector registers (LDS reads issue beside the
// vector ALU), and only the level numbers, which feed scalar address arithmetic, come through scalar loads.
// Workgroup of the one-texel-per-lane variant: sixteen 8 x 8 texel blocks share one staged table on the large levels;
// a level of under a million texels would be too few such workgroups to fill the chip evenly and takes four blocks each.
constexpr int kPrefilterWideThreads = 1024, kPrefilterNarrowThreads = 256;

// min(max(x, 0), top) as the one median instruction it is
PBR_DEV int clamp_to(int x, int top) {
  int r;
  asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(x), "s"(top));
  return r;
}

// (block: the workgroup's number within its level's launch, or within its level's part of a launch of several levels)
template <int SPLIT, bool POW2, int THREADS>
__device__ __forceinline__ void prefilter_block(const DevEquirect& hdr, const PrefilterSample* __restrict__ samples, int n_samples, int n_single, float total_weight,
                                                float4* __restrict__ out, int size, long long first, long long count, unsigned block) {
  constexpr int kThreads = THREADS;
  // per sample {lx, ly, lz, wa}, {kx, cx, ky, cy}: the pixel coordinates in source level l0 are theta * kx + cx and
  // asin(y) * ky + cy, i.e. (angle * scale + 0.5) * size of equirectangular.wgsl:68-69 with the size multiplied in (for
  // power-of-two sizes the same bits: scaling by a power of two commutes with the rounding); and wc on its own
  __shared__ float4 s_table[SPLIT == 1 ? 2 * PBR_PREFILTER_SAMPLES : 1];
  __shared__ float s_wc[SPLIT == 1 ? PBR_PREFILTER_SAMPLES : 1];
  if constexpr (SPLIT == 1) {
    for (int i = threadIdx.x; i < n_samples; i += kThreads) {
      const PrefilterSample s = samples[i];
      const float w0f = (float)max(1, hdr.w >> s.l0), h0f = (float)max(1, hdr.h >> s.l0);
      s_table[2 * i] = make_float4(s.lx, s.ly, s.lz, s.wa);
      s_table[2 * i + 1] = make_float4(INV_ATAN_X * w0f, 0.5f * w0f, INV_ATAN_Y * h0f, 0.5f * h0f);
      s_wc[i] = s.wc;
    }
    __syncthreads();
  }
  const long long gid = (long long)block * kThreads + threadIdx.x;
  long long t;
  int sub;
  bool live;
  if constexpr (SPLIT == 1) {
    // One texel per lane, and a wavefront is an 8 x 8 block of texels rather than a 64 x 1 strip: its 64 directions
    // stay close together, so the source pixels of one sample fall into few cache lines on every face.  (A strip of
    // a +-Y face fans out over a quarter turn of longitude near the pole rows of the equirect image: those faces
    // took twice as long as the others.)  Rows are counted through the faces (face * size + y); the launch covers
    // the 8-row bands that hold the range [first, first + count), lanes outside it idle.
    const long long wave = gid >> 6;
    const int lane = threadIdx.x & 63;
    const int blocks_x = (size + 7) >> 3;
    const long long row0 = (first / size) & ~7ll;
    const long long row = row0 + (wave / blocks_x) * 8 + (lane >> 3);
    const int x = (int)(wave % blocks_x) * 8 + (lane & 7);
    const long long flat = row * size + x;
    live = x < size && row < 6ll * size && flat >= first && flat < first + count;
    t = live ? flat : first;
    sub = 0;
  } else {
    const long long local = gid / SPLIT;
    sub = (int)(gid % SPLIT);
    live = local < count;
    t = first + (live ? local : 0);
  }
  const int gz = (int)(t / ((long long)size * size)), gy = (int)((t / size) % size), gx = (int)(t % size);
  const V3 N = texel_direction(gx, gy, gz, size);
  const V3 upv = fabsf(N.z) < 0.999f ? v3(0, 0, 1) : v3(1, 0, 0);   // prefiltermap.wgsl:56-61
  const V3 T = normalize(cross(upv, N));
  const V3 B = cross(N, T);
  float r = 0, g = 0, b = 0;
  const int log2w = 31 - __clz(hdr.w);   // used when POW2: width a power of two, at least the height, so every level's width is 1 << (log2w - level)
  // the two scale factors of equirectangular.wgsl:2 held in vector registers (see the note on issue rates above)
  float inv_atan_x = INV_ATAN_X, inv_atan_y = INV_ATAN_Y;
  asm volatile("" : "+v"(inv_atan_x), "+v"(inv_atan_y));
  // Where sample i reads: its source pixel in level l0 and, with TWO, the one in level l1, and their weights.
  // (u, v) of equirectangular.wgsl:68 with the device library's polynomials (equirect_math.h); which pixel a sample
  // within an ulp of a pixel boundary lands on is not pinned - the reference's own f32 arithmetic does not pin it.
  auto locate = [&](auto two, int i, const float*& qa, const float*& qc, float& wa, float& wc) {
    constexpr bool TWO = decltype(two)::value;
    float lx, ly, lz, kx, cx, ky, cy;
    int l0, l1;
    if constexpr (SPLIT == 1) {
      const float4 dir = s_table[2 * i], scale = s_table[2 * i + 1];
      lx = dir.x; ly = dir.y; lz = dir.z; wa = dir.w; kx = scale.x; cx = scale.y; ky = scale.z; cy = scale.w;
      if constexpr (TWO) wc = s_wc[i];
      l0 = samples[i].l0; l1 = TWO ? samples[i].l1 : l0;   // the same in every lane: scalar loads
    } else {
      const PrefilterSample s = samples[i];
      lx = s.lx; ly = s.ly; lz = s.lz; wa = s.wa; wc = s.wc; l0 = s.l0; l1 = s.l1;
      const float w0f = (float)max(1, hdr.w >> l0), h0f = (float)max(1, hdr.h >> l0);
      kx = inv_atan_x * w0f; cx = 0.5f * w0f; ky = inv_atan_y * h0f; cy = 0.5f * h0f;
    }
    const V3 L = sample_to_world(T, B, N, lx, ly, lz);
    const int w0 = max(1, hdr.w >> l0), h0 = max(1, hdr.h >> l0);
    const float yc = __builtin_amdgcn_fmed3f(L.y, -1.0f, 1.0f);
    const float theta = atan2_quick(L.z, L.x), phi = asin_quick(yc);
    const float tu = fmaf(theta, kx, cx), tv = fmaf(phi, ky, cy);
    int x0, y0;
    uint32_t i0;   // pixel index within the level
    // The angles are at most pi and pi / 2 in magnitude and the scale factors are truncated below 1 / (2 pi) and 1 / pi
    // (equirectangular.wgsl:2), so the coordinates lie inside [0, size) as they stand (at most 0.99983 and 0.99998
    // of the size) and a NaN converts to 0: the one-texel-per-lane loop, where every instruction counts, does not clamp.
    if constexpr (SPLIT == 1 && POW2) { x0 = (int)tu; y0 = (int)tv; }
    else { x0 = max(min((int)tu, w0 - 1), 0); y0 = max(min((int)tv, h0 - 1), 0); }
    // (the level's start is added to the pointer, where it is scalar arithmetic, not to the per-lane index)
    if constexpr (POW2) i0 = ((uint32_t)y0 << (log2w - l0)) + (uint32_t)x0;
    else i0 = (uint32_t)y0 * (uint32_t)w0 + (uint32_t)x0;
    qa = hdr.rgb + 3 * (size_t)hdr.off[l0] + 3 * (size_t)i0;
    if constexpr (TWO) {
      const int w1 = max(1, hdr.w >> l1), h1 = max(1, hdr.h >> l1);
      int x1, y1;
      uint32_t i1;
      if constexpr (POW2) {
        // power-of-two source: a level's size is the previous one's half, scaling u by it is exact, and the pixel
        // of the coarser level (l1 >= l0) is the finer level's pixel shifted down - the same integers as below
        // (and rows are a shift apart)
        x1 = x0 >> (l1 - l0); y1 = y0 >> (l1 - l0);
        i1 = ((uint32_t)y1 << (log2w - l1)) + (uint32_t)x1;
      } else {
        const float u = fmaf(theta, inv_atan_x, 0.5f), v = fmaf(phi, inv_atan_y, 0.5f);
        x1 = max(min((int)(u * (float)w1), w1 - 1), 0); y1 = max(min((int)(v * (float)h1), h1 - 1), 0);
        i1 = (uint32_t)y1 * (uint32_t)w1 + (uint32_t)x1;
      }
      qc = hdr.rgb + 3 * (size_t)hdr.off[l1] + 3 * (size_t)i1;
    }
  };
  // Several samples at a time: the samples' source pixels are located first and their 12-byte loads issued together,
  // so that a wavefront has several times the requests in flight; the sums take the samples in the table's order.
  // 2048^2 x 10 levels with two-pixel samples only: 16.9 ms one sample at a time, 15.7 / 15.15 / 15.5 / 17.1 ms in groups
  // of 2 / 3 / 4 / 6 (45 / 60 / 66 / 66 registers: beyond 64 a SIMD holds 7 waves).
#ifndef PBR_PREFILTER_GROUP
#define PBR_PREFILTER_GROUP 3
#endif
  constexpr int kGroup = PBR_PREFILTER_GROUP;
  const std::true_type two_levels;
  const std::false_type one_level;
  int i = sub;
  if constexpr (SPLIT == 1) {
    // the table's leading samples read one source level: one pixel, one weight
    for (; i + (kGroup - 1) < n_single; i += kGroup) {
      const float *qa[kGroup], *qc;
      float wa[kGroup], wc;
      V3 a[kGroup];
#pragma unroll
      for (int k = 0; k < kGroup; ++k) locate(one_level, i + k, qa[k], qc, wa[k], wc);
#pragma unroll
      for (int k = 0; k < kGroup; ++k) a[k] = v3(qa[k][0], qa[k][1], qa[k][2]);   // one 12-byte load each
#pragma unroll
      for (int k = 0; k < kGroup; ++k) { r = fmaf(a[k].x, wa[k], r); g = fmaf(a[k].y, wa[k], g); b = fmaf(a[k].z, wa[k], b); }
    }
    for (; i < n_single; ++i) {
      const float *qa, *qc;
      float wa, wc;
      locate(one_level, i, qa, qc, wa, wc);
      r = fmaf(qa[0], wa, r); g = fmaf(qa[1], wa, g); b = fmaf(qa[2], wa, b);
    }
  }
  for (; i + (kGroup - 1) * SPLIT < n_samples; i += kGroup * SPLIT) {
    const float *qa[kGroup], *qc[kGroup];
    float wa[kGroup], wc[kGroup];
    V3 a[kGroup], c[kGroup];
#pragma unroll
    for (int k = 0; k < kGroup; ++k) locate(two_levels, i + k * SPLIT, qa[k], qc[k], wa[k], wc[k]);
#pragma unroll
    for (int k = 0; k < kGroup; ++k) { a[k] = v3(qa[k][0], qa[k][1], qa[k][2]); c[k] = v3(qc[k][0], qc[k][1], qc[k][2]); }   // one 12-byte load each
#pragma unroll
    for (int k = 0; k < kGroup; ++k) { r += a[k].x * wa[k] + c[k].x * wc[k]; g += a[k].y * wa[k] + c[k].y * wc[k]; b += a[k].z * wa[k] + c[k].z * wc[k]; }
  }
  for (; i < n_samples; i += SPLIT) {
    const float *qa, *qc;
    float wa, wc;
    locate(two_levels, i, qa, qc, wa, wc);
    const V3 a = v3(qa[0], qa[1], qa[2]), c = v3(qc[0], qc[1], qc[2]);
    r += a.x * wa + c.x * wc; g += a.y * wa + c.y * wc; b += a.z * wa + c.z * wc;
  }
#pragma unroll
  for (int o = SPLIT / 2; o > 0; o >>= 1) { r += __shfl_xor(r, o); g += __shfl_xor(g, o); b += __shfl_xor(b, o); }
  if (live && sub == 0) out[t] = make_float4(r / total_weight, g / total_weight, b / total_weight, 1.0f);
}

template <int SPLIT, bool POW2, int THREADS = 256>
__global__ void __launch_bounds__(THREADS)
prefilter_kernel(DevEquirect hdr, const PrefilterSample* __restrict__ samples, int n_samples, int n_single, float total_weight, float4* __restrict__ out, int size,
                 long long first, long long count) {
  prefilter_block<SPLIT, POW2, THREADS>(hdr, samples, n_samples, n_single, total_weight, out, size, first, count, blockIdx.x);
}

// The large levels of a call (262 144 texels and more: the one-texel-per-lane variant, all in its wide shape) in one
// launch.  A launch ends with its last wavefronts' chains of a thousand samples running on a chip that is emptying, about
// a third of a millisecond per large level, and the next level's launch waits for it.  In one launch a level's
// workgroups take the places the level before it frees; the levels come dearest first and roughness 0 (one sample per
// texel) last, where it fills the tail: 13.05 -> 12.2 ms for the 2048^2 x 10 map.  A texel's arithmetic is what it is
// in a launch of its level alone.  (All ten levels in one launch, the small ones through their own lane splits in the
// same kernel, ran as long as ten launches: the kernel that holds all three variants is slower in its main loop, and
// the small levels gain nothing from sharing a launch - the 128^2 x 5 map took 0.39 ms against 0.36.)
struct PrefilterLevelJob {
  const PrefilterSample* samples; int n_samples, n_single; float total_weight;
  float4* out; int size; long long first, count;
  unsigned first_block;   // the launch's workgroups from here on (up to the next job's) are this level's
};
constexpr int kPrefilterMaxJobs = 8;
struct PrefilterLevelJobs { int n; PrefilterLevelJob job[kPrefilterMaxJobs]; };

template <bool POW2>
__global__ void __launch_bounds__(kPrefilterWideThreads) prefilter_levels_kernel(DevEquirect hdr, PrefilterLevelJobs jobs) {
  int k = 0;
  while (k + 1 < jobs.n && blockIdx.x >= jobs.job[k + 1].first_block) ++k;
  const PrefilterLevelJob& j = jobs.job[k];
  prefilter_block<1, POW2, kPrefilterWideThreads>(hdr, j.samples, j.n_samples, j.n_single, j.total_weight, j.out, j.size, j.first, j.count, blockIdx.x - j.first_block);
}

// One workgroup covers 256 texels of one LUT row; the row's roughness fixes the half-vector table,
// which is built once in LDS and then read as a broadcast by every lane.
__global__ void __launch_bounds__(256) brdf_lut_kernel(uint16_t* __restrict__ out, int size) {
  __shared__ float s_h[PBR_PREFILTER_SAMPLES][3];
  const int y = blockIdx.y, x = blockIdx.x * 256 + threadIdx.x;
  const float fsize = (float)size;
  const float roughness = (fsize - (float)y) / (fsize - 1.0f);   // brdf.wgsl:104 (row 0 is above 1)
  const float a = roughness * roughness;
  for (int i = threadIdx.x; i < PBR_PREFILTER_SAMPLES; i += 256) {
    const float xi_x = (float)i / (float)PBR_PREFILTER_SAMPLES;
    const float xi_y = (float)__brev((uint32_t)i) * 2.3283064365386963e-10f;   // brdf.wgsl:9-16
    const float phi = 2.0f * PI_F * xi_x;
    const float ct = sqrtf((1.0f - xi_y) / (1.0f + (a * a - 1.0f) * xi_y));
    const float st = sqrtf(1.0f - ct * ct);
    // N = (0,0,1) makes the tangent frame tangent = (0,-1,0), bitangent = (1,0,0) (brdf.wgsl:33-43)
    V3 h = normalize(v3(sinf(phi) * st, -(cosf(phi) * st), ct));
    s_h[i][0] = h.x; s_h[i][1] = h.y; s_h[i][2] = h.z;
  }
  __syncthreads();
  if (x >= size) return;
  const float ndotv = (float)x / (fsize - 1.0f);
  const V3 V = v3(sqrtf(1.0f - ndotv * ndotv), 0.0f, ndotv);
  const float k = (roughness * roughness) / 2.0f;                 // brdf.wgsl:57-59
  const float g_v = ndotv / (ndotv * (1.0f - k) + k);
  float A = 0.0f, Bsum = 0.0f;
  for (int i = 0; i < PBR_PREFILTER_SAMPLES; ++i) {
    const V3 H = v3(s_h[i][0], s_h[i][1], s_h[i][2]);
    const float vh = dot(V, H);
    const V3 L = normalize(H * (2.0f * vh) - V);
    const float ndotl = fmaxf(L.z, 0.0f), ndoth = fmaxf(H.z, 0.0f), vdoth = fmaxf(vh, 0.0f);
    if (ndotl > 0.0f) {
      const float g_l = ndotl / (ndotl * (1.0f - k) + k);
      const float G = g_v * g_l;
      const float g_vis = (G * vdoth) / (ndoth * ndotv);          // 0/0 = NaN in column 0, kept on purpose
      const float fc = pow5(1.0f - vdoth);
      A += (1.0f - fc) * g_vis;
      Bsum += fc * g_vis;
    }
  }
  A /= (float)PBR_PREFILTER_SAMPLES; Bsum /= (float)PBR_PREFILTER_SAMPLES;
  ushort4 o;
  o.x = __half_as_ushort(__float2half_rn(A));
  o.y = __half_as_ushort(__float2half_rn(Bsum));
  o.z = __half_as_ushort(__float2half_rn(0.0f));
  o.w = __half_as_ushort(__float2half_rn(1.0f));
  reinterpret_cast<ushort4*>(out)[(size_t)y * size + x] = o;
}

}  // namespace

void launch_build_mips_rgba32f(hipStream_t s, float4* chain, const uint32_t* off, int w, int h, int levels) {
  for (int l = 1; l < levels; ++l) {
    int sw = std::max(1, w >> (l - 1)), sh = std::max(1, h >> (l - 1)), dw = std::max(1, w >> l), dh = std::max(1, h >> l);
    dim3 grid((dw + 63) / 64, (dh + 3) / 4);
    hipLaunchKernelGGL(mip_rgba32f_kernel, grid, dim3(256), 0, s, chain + off[l - 1], chain + off[l], sw, sh, dw, dh);
  }
}

void launch_build_mips_rgba8(hipStream_t s, const uint32_t* linear_level0, uint32_t* chain, const uint32_t* off, int w, int h, int levels, int srgb,
                             const float* srgb_tables) {
  hipLaunchKernelGGL(block_rgba8_kernel, dim3((w + 63) / 64, (h + 3) / 4), dim3(256), 0, s, linear_level0, chain + off[0], w, h);
  for (int l = 1; l < levels; ++l) {
    int sw = std::max(1, w >> (l - 1)), sh = std::max(1, h >> (l - 1)), dw = std::max(1, w >> l), dh = std::max(1, h >> l);
    dim3 grid((dw + 63) / 64, (dh + 3) / 4);
    hipLaunchKernelGGL(mip_rgba8_kernel, grid, dim3(256), 0, s, chain + off[l - 1], chain + off[l], sw, sh, dw, dh, srgb, srgb_tables);
  }
}

void launch_drop_alpha(hipStream_t s, const float4* rgba, float* rgb, size_t texels) {
  hipLaunchKernelGGL(drop_alpha_kernel, dim3((unsigned)((texels + 255) / 256)), dim3(256), 0, s, rgba, rgb, texels);
}

void launch_equirect_to_cube(hipStream_t s, DevEquirect hdr, float4* out, int size) {
  dim3 grid((size + 63) / 64, (size + 3) / 4, 6);
  hipLaunchKernelGGL(equirect_to_cube_kernel, grid, dim3(256), 0, s, hdr, out, size);
}

void launch_irradiance(hipStream_t s, DevEquirect hdr, const float* tables, int n_phi, int n_theta, float4* out, int size) {
  hipLaunchKernelGGL(irradiance_kernel, dim3(6 * size * size), dim3(256), 0, s, hdr, tables, n_phi, n_theta, out, size);
}

void launch_prefilter(hipStream_t s, DevEquirect hdr, const PrefilterSample* samples, int n_samples, int n_single, float total_weight,
                      float4* out_level, int level_size, int64_t first, int64_t count) {
  if (count <= 0) return;
  const bool pow2 = (hdr.w & (hdr.w - 1)) == 0 && (hdr.h & (hdr.h - 1)) == 0 && hdr.w >= hdr.h;
  // Few texels: spread each texel's samples over a whole wave so the chip still fills.  The split fixes the order
  // in which a texel's samples are summed, so it follows the size of the level, not of the range: a texel gets the
  // same bits whichever shard computes it.
  const long long level_texels = 6ll * level_size * level_size;
  if (level_texels >= (1 << 18)) {
    // 8-row bands that hold the range, one wavefront per 8 x 8 block of texels
    const long long row0 = (first / level_size) & ~7ll, row1 = (first + count - 1) / level_size;
    const long long threads = ((row1 - row0) / 8 + 1) * ((level_size + 7) / 8) * 64;
    const bool wide = level_texels >= (1 << 20);
    const int per_group = wide ? kPrefilterWideThreads : kPrefilterNarrowThreads;
    const dim3 grid((unsigned)((threads + per_group - 1) / per_group)), block(per_group);
#define PBR_LAUNCH_PREFILTER(POW2_, THREADS_) \
    hipLaunchKernelGGL((prefilter_kernel<1, POW2_, THREADS_>), grid, block, 0, s, hdr, samples, n_samples, n_single, total_weight, out_level, level_size, (long long)first, \
                       (long long)count)
    if (pow2) { if (wide) PBR_LAUNCH_PREFILTER(true, kPrefilterWideThreads); else PBR_LAUNCH_PREFILTER(true, kPrefilterNarrowThreads); }
    else { if (wide) PBR_LAUNCH_PREFILTER(false, kPrefilterWideThreads); else PBR_LAUNCH_PREFILTER(false, kPrefilterNarrowThreads); }
#undef PBR_LAUNCH_PREFILTER
  } else if (level_texels >= (1 << 14)) {
    long long threads = count * 8;
    if (pow2) hipLaunchKernelGGL((prefilter_kernel<8, true>), dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, hdr, samples, n_samples, 0, total_weight,
                                 out_level, level_size, (long long)first, (long long)count);
    else hipLaunchKernelGGL((prefilter_kernel<8, false>), dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, hdr, samples, n_samples, 0, total_weight,
                            out_level, level_size, (long long)first, (long long)count);
  } else {
    long long threads = count * 64;
    if (pow2) hipLaunchKernelGGL((prefilter_kernel<64, true>), dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, hdr, samples, n_samples, 0, total_weight,
                                 out_level, level_size, (long long)first, (long long)count);
    else hipLaunchKernelGGL((prefilter_kernel<64, false>), dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, hdr, samples, n_samples, 0, total_weight,
                            out_level, level_size, (long long)first, (long long)count);
  }
}

void launch_prefilter_levels(hipStream_t s, DevEquirect hdr, const PrefilterLevelRun* runs, int n_runs) {
  const bool pow2 = (hdr.w & (hdr.w - 1)) == 0 && (hdr.h & (hdr.h - 1)) == 0 && hdr.w >= hdr.h;
  // the levels that take the one-texel-per-lane variant share a launch (in its wide shape); every other level has a launch of its own
  std::vector<const PrefilterLevelRun*> wide;
  for (int i = 0; i < n_runs; ++i) {
    const PrefilterLevelRun& r = runs[i];
    if (r.count <= 0) continue;
#ifdef PBR_PREFILTER_NO_MERGE
    const bool shares = false;   // measurement builds: a launch per level
#else
#ifndef PBR_PREFILTER_SHARE_LOG2
#define PBR_PREFILTER_SHARE_LOG2 18   // every level that takes the one-texel-per-lane variant (12.3 -> 12.0 ms against sharing from a million texels)
#endif
    const bool shares = 6ll * r.level_size * r.level_size >= (1ll << PBR_PREFILTER_SHARE_LOG2);
#endif
    if (shares) wide.push_back(&r);
    else launch_prefilter(s, hdr, r.samples, r.n_samples, r.n_single, r.total_weight, r.out_level, r.level_size, r.first, r.count);
  }
  if (wide.size() < 2 || wide.size() > (size_t)kPrefilterMaxJobs) {
    for (const PrefilterLevelRun* r : wide) launch_prefilter(s, hdr, r->samples, r->n_samples, r->n_single, r->total_weight, r->out_level, r->level_size, r->first, r->count);
    return;
  }
  // dearest first; a level of one sample per texel (roughness 0) last
  std::stable_sort(wide.begin(), wide.end(), [](const PrefilterLevelRun* a, const PrefilterLevelRun* b) {
    const bool a_last = a->n_samples <= 1, b_last = b->n_samples <= 1;
    if (a_last != b_last) return b_last;
    return a->count * (double)a->n_samples > b->count * (double)b->n_samples;
  });
  PrefilterLevelJobs jobs = {};
  unsigned blocks = 0;
  for (const PrefilterLevelRun* r : wide) {
    PrefilterLevelJob& j = jobs.job[jobs.n++];
    j.samples = r->samples; j.n_samples = r->n_samples; j.n_single = r->n_single; j.total_weight = r->total_weight;
    j.out = r->out_level; j.size = r->level_size; j.first = r->first; j.count = r->count; j.first_block = blocks;
    const long long row0 = (r->first / r->level_size) & ~7ll, row1 = (r->first + r->count - 1) / r->level_size;
    const long long threads = ((row1 - row0) / 8 + 1) * ((r->level_size + 7) / 8) * 64;
    blocks += (unsigned)((threads + kPrefilterWideThreads - 1) / kPrefilterWideThreads);
  }
  if (pow2) hipLaunchKernelGGL(prefilter_levels_kernel<true>, dim3(blocks), dim3(kPrefilterWideThreads), 0, s, hdr, jobs);
  else hipLaunchKernelGGL(prefilter_levels_kernel<false>, dim3(blocks), dim3(kPrefilterWideThreads), 0, s, hdr, jobs);
}

void launch_brdf_lut(hipStream_t s, uint16_t* out, int size) {
  dim3 grid((size + 255) / 256, size);
  hipLaunchKernelGGL(brdf_lut_kernel, grid, dim3(256), 0, s, out, size);
}

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
  // selects only, no branches: lanes of one wavefront usually look at different faces
  const float ax = fabsf(r.x), ay = fabsf(r.y), az = fabsf(r.z);
  const bool use_z = az >= ax && az >= ay, use_y = !use_z && ay >= ax;
  const float rc = use_z ? r.z : use_y ? r.y : r.x;
  const bool pos = rc >= 0.0f;
  CubeCoord c;
  c.rc = rc;
  c.face = (use_z ? 4 : use_y ? 2 : 0) + (pos ? 0 : 1);
  const float sx = use_z || use_y ? r.x : -r.z;          // sc before the sign of the major axis is applied
  c.sc = (use_y || pos) ? sx : -sx;
  const float tz = pos ? r.z : -r.z;
  c.tc = use_y ? tz : -r.y;
  return c;
}

PBR_DEV uint32_t pack_color(V3 color) {
#pragma clang fp contract(fast)
  // pow(c, 1/2.2) as exp2(log2(c) / 2.2) (shader.wgsl:157, skybox.wgsl:30): 0 -> 0, negative or NaN -> NaN -> 0
  // unorm8 conversion: v_cvt_pk_u8_f32 rounds to nearest even, saturates to [0, 255], turns NaN into 0 and drops
  // the byte into place, which is the whole of "clamp, scale, round, pack" in one instruction per channel
  // the scale to 0..255 rides in the exponent: 255 c^(1/2.2) = exp2(log2(c) / 2.2 + log2(255))
  const float g = 1.0f / 2.2f, k = 7.994353436858858f;
  uint32_t packed = 0xff000000u;
  packed = __builtin_amdgcn_cvt_pk_u8_f32(exp2_fast(fmaf(log2_fast(color.x), g, k)), 0u, packed);
  packed = __builtin_amdgcn_cvt_pk_u8_f32(exp2_fast(fmaf(log2_fast(color.y), g, k)), 1u, packed);
  packed = __builtin_amdgcn_cvt_pk_u8_f32(exp2_fast(fmaf(log2_fast(color.z), g, k)), 2u, packed);
  return packed;
}


struct CubeFoot { uint32_t o; uint32_t m; float w[4]; };   // taps at o, o + 1, o + m, o + m + 1

// Texel offset (from cube.base) and weights of one seamless bilinear lookup on one level of a bordered
// cube: the border row / column of a face holds the adjoining texels of the neighbouring faces and a
// border corner the mean of the three texels around it, which is what a lookup across an edge or a
// corner blends, so the four taps always sit in one face.
PBR_DEV CubeFoot cube_foot(const DevCubeRgb& bordered, int level, const CubeCoord& c) {
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
struct Rgb { float x, y, z; };   // one texel of a DevCubeRgb: 12 bytes, fetched with one three-dword load
PBR_DEV void cube_load4(const float* base, const CubeFoot& f, Rgb t[4]) {
  const Rgb* q = reinterpret_cast<const Rgb*>(base) + f.o;
#ifdef PBR_ABLATE_IBL_TAPS
  // measurement build only (wrong colours): no loads, the address kept alive
  const float fake = __uint_as_float(((uint32_t)(size_t)q & 0x3fffffu) | 0x3e000000u);
  t[0] = Rgb{fake, fake, fake}; t[1] = t[0]; t[2] = t[0]; t[3] = t[0];
#else
  t[0] = q[0]; t[1] = q[1]; t[2] = q[f.m]; t[3] = q[f.m + 1];
#endif
}


PBR_DEV V3 weigh4(const Rgb t[4], const float w[4]) {
#pragma clang fp contract(fast)
  return v3(((t[0].x * w[0] + t[1].x * w[1]) + t[2].x * w[2]) + t[3].x * w[3], ((t[0].y * w[0] + t[1].y * w[1]) + t[2].y * w[2]) + t[3].y * w[3],
            ((t[0].z * w[0] + t[1].z * w[1]) + t[2].z * w[2]) + t[3].z * w[3]);
}


// Face coordinates (sc, tc, rc) of direction d on a given face: the rows of the face-selection table.
PBR_DEV void cube_face_coords(int face, V3 d, float& sc, float& tc, float& rc) {
  const bool fx = face < 2, fy = (face & 6) == 2, odd = (face & 1) != 0;
  rc = fx ? d.x : fy ? d.y : d.z;
  const float s0 = fx ? -d.z : d.x;                       // faces 0, 2, 3, 4; negated on faces 1 and 5
  sc = (odd && !fy) ? -s0 : s0;
  tc = fy ? (odd ? -d.z : d.z) : -d.y;
}

// Two-wide arithmetic: CDNA issues one vector instruction per SIMD every four cycles whether it carries
// one f32 per lane or two (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32), so the skybox works on pairs of
// pixels.  Each element goes through exactly the operations of the one-pixel functions above.
typedef float f2 __attribute__((ext_vector_type(2)));

// x * y + z for x, y below 2^24, spelled out: from __umul24 and an addition the compiler, when it knows a factor small
// or wants the sum as a 64-bit index, makes a v_mad_u64_u32 (or v_mad_i64_i32) into a register pair and clears the
// pair's upper half first; this is one instruction into one register.
PBR_DEV uint32_t mad_u24(uint32_t x, uint32_t y, uint32_t z) {
  uint32_t r;
  asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "v"(z));
  return r;
}

struct CubeFootPair { uint32_t o[2]; uint32_t m; f2 w[4]; };   // element k of each weight belongs to the pixel with offset o[k] (counted from level 0)

// cube_foot on level 0 for two lookups at once (faces and face coordinates given per element).
PBR_DEV CubeFootPair cube_foot_pair(const DevCubeRgb& bordered, int face0, int face1, f2 sc, f2 tc, f2 rc) {
#pragma clang fp contract(fast)
  CubeFootPair f;
  const int n = max(1, bordered.size), m = n + 2;
  const float fn = (float)n;
  const f2 ima = {rcp_fast(fabsf(rc.x)), rcp_fast(fabsf(rc.y))};
  const f2 s = 0.5f * (sc * ima) + 0.5f, t = 0.5f * (tc * ima) + 0.5f;
  // Texel coordinates counted from the border texel: tu + 1 = s n + 1/2 lies in [1/2, n + 1/2], so its integer part
  // (a plain conversion: nothing is negative) is the column in the bordered face and its fraction the blend weight.
  // |sc| and |tc| never exceed |rc| (rc is the component the face was selected by), and the host refuses a frame
  // whose unprojection is not finite, so the coordinates need no clamp; a NaN (a zero direction) converts to column 0.
  const f2 tu = s * fn + 0.5f, tv = t * fn + 0.5f;
  const f2 a = {__builtin_amdgcn_fractf(tu.x), __builtin_amdgcn_fractf(tu.y)}, b = {__builtin_amdgcn_fractf(tv.x), __builtin_amdgcn_fractf(tv.y)};
  const f2 oma = 1.0f - a, omb = 1.0f - b;
  f.w[0] = oma * omb; f.w[1] = a * omb; f.w[2] = oma * b; f.w[3] = a * b;
  const int i0 = (int)tu.x, j0 = (int)tv.x, i1 = (int)tu.y, j1 = (int)tv.y;
  // texel index with 24-bit multiply-adds (face * m + row and m are far below 2^24 for any cube the library accepts)
  f.o[0] = mad_u24(mad_u24((uint32_t)face0, (uint32_t)m, (uint32_t)j0), (uint32_t)m, (uint32_t)i0);
  f.o[1] = mad_u24(mad_u24((uint32_t)face1, (uint32_t)m, (uint32_t)j1), (uint32_t)m, (uint32_t)i1);
  f.m = (uint32_t)m;
  return f;
}

// One bilinear lookup of pixel E of a pair from the paired copy of the cube (DevCubeRgb::pairs): `top` and `bottom`
// are the entries at the footprint's two rows, each holding the row's two texels channel by channel, so a channel is
// two v_dot2_f32_f16 (products and sums in f32) against the rows' weights rounded to halves.  Texels and weights carry
// eleven significant bits: the colour moves by up to a thousandth of itself, a tenth of an 8-bit step after the gamma
// curve, so a few pixels in a hundred land on the neighbouring 8-bit value and none moves further.
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
// texels . w with nothing to add to: the three-operand form of the instruction takes the zero as a constant, where the
// builtin compiles to the accumulating two-operand form and a move that clears its accumulator first
PBR_DEV float dot2_alone(uint32_t texels, h2 w) {
  float r;
  uint32_t wb; __builtin_memcpy(&wb, &w, 4);
  asm("v_dot2_f32_f16 %0, %1, %2, 0" : "=v"(r) : "v"(texels), "v"(wb));
  return r;
}
template <int E>
PBR_DEV V3 weigh_pairs(const uint4& top, const uint4& bottom, const f2 w[4]) {
  const h2 wt = __builtin_convertvector((f2){w[0][E], w[1][E]}, h2), wb = __builtin_convertvector((f2){w[2][E], w[3][E]}, h2);
  auto halves = [](uint32_t v) { h2 r; __builtin_memcpy(&r, &v, 4); return r; };
  return v3(__builtin_amdgcn_fdot2(halves(top.x), wt, dot2_alone(bottom.x, wb), false), __builtin_amdgcn_fdot2(halves(top.y), wt, dot2_alone(bottom.y, wb), false),
            __builtin_amdgcn_fdot2(halves(top.z), wt, dot2_alone(bottom.z, wb), false));
}

// pack_color for two pixels: the two multiplies of each channel run two-wide.
PBR_DEV void pack_color_pair(V3 c0, V3 c1, uint32_t& out0, uint32_t& out1) {
#pragma clang fp contract(fast)
  // 255 c^(1/2.2) = exp2(log2(c) / 2.2 + log2(255)): one multiply-add per channel pair between the two transcendentals
  const float g = 1.0f / 2.2f, k = 7.994353436858858f;
#ifdef PBR_ABLATE_SKY_GAMMA
  {   // measurement build only (wrong colours): what the twelve transcendentals per pixel pair cost the frame - 0.8 us of 49

    uint32_t p0 = 0xff000000u, p1 = 0xff000000u;
    p0 = __builtin_amdgcn_cvt_pk_u8_f32(c0.x * 255.0f, 0u, p0); p1 = __builtin_amdgcn_cvt_pk_u8_f32(c1.x * 255.0f, 0u, p1);
    p0 = __builtin_amdgcn_cvt_pk_u8_f32(c0.y * 255.0f, 1u, p0); p1 = __builtin_amdgcn_cvt_pk_u8_f32(c1.y * 255.0f, 1u, p1);
    p0 = __builtin_amdgcn_cvt_pk_u8_f32(c0.z * 255.0f, 2u, p0); p1 = __builtin_amdgcn_cvt_pk_u8_f32(c1.z * 255.0f, 2u, p1);
    out0 = p0; out1 = p1;
    return;
  }
#endif
  const f2 lx = (f2){log2_fast(c0.x), log2_fast(c1.x)} * g + k, ly = (f2){log2_fast(c0.y), log2_fast(c1.y)} * g + k, lz = (f2){log2_fast(c0.z), log2_fast(c1.z)} * g + k;
  const f2 ex = {exp2_fast(lx.x), exp2_fast(lx.y)}, ey = {exp2_fast(ly.x), exp2_fast(ly.y)}, ez = {exp2_fast(lz.x), exp2_fast(lz.y)};
  uint32_t p0 = 0xff000000u, p1 = 0xff000000u;
  p0 = __builtin_amdgcn_cvt_pk_u8_f32(ex.x, 0u, p0); p1 = __builtin_amdgcn_cvt_pk_u8_f32(ex.y, 0u, p1);
  p0 = __builtin_amdgcn_cvt_pk_u8_f32(ey.x, 1u, p0); p1 = __builtin_amdgcn_cvt_pk_u8_f32(ey.y, 1u, p1);
  p0 = __builtin_amdgcn_cvt_pk_u8_f32(ez.x, 2u, p0); p1 = __builtin_amdgcn_cvt_pk_u8_f32(ez.y, 2u, p1);
  out0 = p0; out1 = p1;
}

// Four pixels' colours in one 16-byte store.  The colour target is 33 MB per 4K frame that no kernel reads back: the
// stores are marked non-temporal so that they stream past the L2s instead of evicting the textures and cube maps
// from them (52.8 -> 51.4 us per 4K frame with four in flight).
PBR_DEV void store_colour4(uint32_t* dst, const uint32_t c[4]) {
  typedef uint32_t u4 __attribute__((ext_vector_type(4)));
  const u4 v = {c[0], c[1], c[2], c[3]};
  __builtin_nontemporal_store(v, reinterpret_cast<u4*>(dst));
}

// Skybox colour of four horizontally adjacent pixels (x0 .. x0+3, y): the clip-space position is
// unprojected with the inverse view-projection matrix and the bordered cube sampled bilinearly from its
// paired copy; all eight loads are issued before any is used.  The unprojected direction is affine in x, so the
// cube face is selected once, for the first pixel, and the face coordinates of the other three are
// that pixel's plus a constant step; when the group straddles a face edge (or w changes sign in
// it) the other three take the full selection again.
PBR_DEV void sky_pixels4(const FrameParams* __restrict__ p, const DevCubeRgb& sky, int x0, int y, uint32_t out[4]) {
#pragma clang fp contract(fast)
  // inverse view-projection applied to the pixel centre's clip position (skybox.wgsl:25-27), as the affine function of
  // the pixel coordinates it is: t(x, y) = a + x bx + y by, the three vectors prepared once per frame on the host
  const float* ta = p->sky_a; const float* tbx = p->sky_bx; const float* tby = p->sky_by;
  const float fx = (float)x0, fy = (float)y;
  const float tx = fmaf(fx, tbx[0], fmaf(fy, tby[0], ta[0])), ty = fmaf(fx, tbx[1], fmaf(fy, tby[1], ta[1]));
  const float tz = fmaf(fx, tbx[2], fmaf(fy, tby[2], ta[2])), tw = fmaf(fx, tbx[3], fmaf(fy, tby[3], ta[3]));
  // normalize(xyz / w) only rescales the lookup direction by 1 / (w |xyz|); the cube lookup depends on
  // the direction alone, so only the sign of w is applied.
  const float sg = tw < 0.0f ? -1.0f : 1.0f;
  const V3 d0 = v3(tx * sg, ty * sg, -(tz * sg));
  const V3 step = v3(tbx[0] * sg, tbx[1] * sg, -(tbx[2] * sg));
  const float w_step = tbx[3];
  const CubeCoord first = cube_select(d0);
  float dsc, dtc, drc;
  cube_face_coords(first.face, step, dsc, dtc, drc);
  // A face is a convex cone of directions and the four directions lie on a segment, so all four pixels are on
  // the first pixel's face (with w of one sign) exactly when the last one is: one test instead of three.
  const float w3 = tw + 3.0f * w_step;
  const f2 k01 = {0.0f, 1.0f}, k23 = {2.0f, 3.0f};
  f2 sc01 = first.sc + k01 * dsc, tc01 = first.tc + k01 * dtc, rc01 = first.rc + k01 * drc;
  f2 sc23 = first.sc + k23 * dsc, tc23 = first.tc + k23 * dtc, rc23 = first.rc + k23 * drc;
  sc01.x = first.sc; tc01.x = first.tc; rc01.x = first.rc;   // pixel 0 exactly as selected (0 * step may be NaN)
  int face[4] = {first.face, first.face, first.face, first.face};
  const bool one_face = fabsf(rc23.y) >= fabsf(sc23.y) && fabsf(rc23.y) >= fabsf(tc23.y) && (rc23.y < 0.0f) == (first.rc < 0.0f) && (w3 < 0.0f) == (tw < 0.0f);
#ifdef SKY_NO_FALLBACK
  if (false) {
#else
  if (!one_face) {
#endif
    CubeCoord c[4];
#pragma unroll
    for (int k = 1; k < 4; ++k) {
      const float wk = tw + (float)k * w_step;
      const float sk = wk < 0.0f ? -1.0f : 1.0f;
      const float ax = fmaf((float)k, tbx[0], tx), ay = fmaf((float)k, tbx[1], ty), az = fmaf((float)k, tbx[2], tz);
      c[k] = cube_select(v3(ax * sk, ay * sk, -(az * sk)));
      face[k] = c[k].face;
    }
    sc01.y = c[1].sc; tc01.y = c[1].tc; rc01.y = c[1].rc;
    sc23 = (f2){c[2].sc, c[3].sc}; tc23 = (f2){c[2].tc, c[3].tc}; rc23 = (f2){c[2].rc, c[3].rc};
  }
  const CubeFootPair fa = cube_foot_pair(sky, face[0], face[1], sc01, tc01, rc01), fb = cube_foot_pair(sky, face[2], face[3], sc23, tc23, rc23);
  uint4 top[4], bottom[4];
  // bytes from an entry to the one below it: the same for every lane; taken through readfirstlane so that it stays one
  // scalar that is added rather than multiplied out again per lane
  const size_t pitch = (size_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(fa.m * (uint32_t)sizeof(uint4)));
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const CubeFootPair& f = k < 2 ? fa : fb;
    const uint4* q = sky.pairs + f.o[k & 1];
    const uint4* below = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(q) + pitch);
#ifdef PBR_ABLATE_SKY_TAPS
    // measurement build only (wrong colours): no loads, the addresses kept alive
    const uint32_t fake = ((uint32_t)(size_t)q & 0x03ff03ffu) | 0x30003000u, fake2 = ((uint32_t)(size_t)below & 0x03ff03ffu) | 0x30003000u;
    top[k] = make_uint4(fake, fake2, fake, 0u); bottom[k] = make_uint4(fake2, fake, fake2, 0u);
#else
    top[k] = *q; bottom[k] = *below;
#endif
  }
  pack_color_pair(weigh_pairs<0>(top[0], bottom[0], fa.w), weigh_pairs<1>(top[1], bottom[1], fa.w), out[0], out[1]);
  pack_color_pair(weigh_pairs<0>(top[2], bottom[2], fb.w), weigh_pairs<1>(top[3], bottom[3], fb.w), out[2], out[3]);
}

// The skybox pass is cut into the same 32 x 32 pixel tiles as the rasteriser.  A tile can be shaded by
// any 256-thread workgroup: the geometry kernels of a frame are short and latency-bound, so a share
// of the tiles rides along in their launches as extra workgroups and fills the otherwise idle chip.
struct SkyShare {
  DevCubeRgb sky;     // bordered skybox cube
  uint32_t* rgba8;    // colour target of the row range
  int first, count;   // tiles [first, first + count) are shaded by the workgroups after the kernel's own
  int own_blocks;     // number of workgroups that do the kernel's own work
};

// Every pixel of tile `tile` gets the skybox colour (covered ones are overwritten by the surface pass).
PBR_DEV void sky_tile(const FrameParams* __restrict__ p, const DevCubeRgb& sky, uint32_t* __restrict__ rgba8, int tile) {
  // (a division by a number only known at run time is some thirty dependent instructions at the head of every workgroup)
  const int tile_y = p->tiles_x == 1 ? tile : (int)__umulhi((uint32_t)tile, p->tiles_x_magic), tile_x = tile - tile_y * p->tiles_x;
  const int x = tile_x * PBR_TILE + (threadIdx.x & 7) * 4, y = p->y0 + tile_y * PBR_TILE + (threadIdx.x >> 3);
  if (x >= p->W || y >= p->y1) return;
  uint32_t c[4];
  sky_pixels4(p, sky, x, y, c);
  const uint32_t o = mad_u24((uint32_t)(y - p->y0), (uint32_t)p->W, (uint32_t)x);   // a row range has fewer than 2^32 pixels
  if (x + 3 < p->W && (o & 3) == 0) { store_colour4(rgba8 + o, c); return; }
#pragma unroll
  for (int k = 0; k < 4; ++k) if (x + k < p->W) rgba8[o + k] = c[k];
}

// The two inverse trigonometric functions of the equirect mapping (equirectangular.wgsl:68, prefiltermap.wgsl:157-163)
// as the device library evaluates them, operation for operation, with the library's special-case handling split off:
// atan2_regular / asin_regular give the bits of atan2f / asinf for every argument equirect_regular accepts, and the
// caller evaluates the library functions themselves for the rest (both arguments zero, infinities, NaN, magnitudes
// where the library's scaled division and a plain multiply by the reciprocal could round differently).  The library
// computes min / max through a division protected against overflow by frexp / ldexp scaling; for magnitudes within
// 2^+-60 that quotient is the product with the hardware reciprocal, bit for bit, because both round the same
// significand product.  scripts/equirect_exact_check.hip compares the bits over 2^33 directions.
#pragma once
#include "devmath.h"

#define EQUIRECT_K(bits) __uint_as_float(bits##u)

// |x| and |y| both ordinary and not both zero: the arguments atan2_regular covers.
PBR_DEV bool equirect_regular(float y, float x) {
  // 2^-60 <= max(|x|, |y|) < 2^60 as one unsigned comparison of the exponent field (infinities and NaN lie above)
  const uint32_t m = __float_as_uint(fmaxf(fabsf(x), fabsf(y)));
  return m - 0x21800000u < 0x3c000000u;
}

PBR_DEV float atan2_regular(float y, float x) {
#pragma clang fp contract(off)
  const float ax = fabsf(x), ay = fabsf(y);
  const float u = fmaxf(ax, ay), v = fminf(ax, ay);
  const float a = v * rcp_fast(u);
  const float s = a * a;
  float p = fmaf(s, EQUIRECT_K(0x3b2d2a58), EQUIRECT_K(0xbc7a590c));
  p = fmaf(s, p, EQUIRECT_K(0x3d29fb3f));
  p = fmaf(s, p, EQUIRECT_K(0xbd97d4d7));
  p = fmaf(s, p, EQUIRECT_K(0x3dd931b2));
  p = fmaf(s, p, EQUIRECT_K(0xbe1160e6));
  p = fmaf(s, p, EQUIRECT_K(0x3e4cb8bf));
  p = fmaf(s, p, EQUIRECT_K(0xbeaaaa62));
  p = s * p;
  float r = fmaf(a, p, a);
  r = ay > ax ? EQUIRECT_K(0x3fc90fdb) - r : r;
  r = x < 0.0f ? EQUIRECT_K(0x40490fdb) - r : r;
  return copysignf(r, y);
}

// |y| <= 1 (the caller clamps).
PBR_DEV float asin_regular(float y) {
#pragma clang fp contract(off)
  const float ay = fabsf(y);
  const bool small = ay < 0.5f;   // (one predicate for both selections; the library tests ay >= 0.5 here, the same for every number)
  const float s = small ? y * y : fmaf(ay, -0.5f, 0.5f);
  float p = fmaf(s, EQUIRECT_K(0x3d1c21a7), EQUIRECT_K(0x3c5fc5da));
  p = fmaf(s, p, EQUIRECT_K(0x3d034c3c));
  p = fmaf(s, p, EQUIRECT_K(0x3d3641b1));
  p = fmaf(s, p, EQUIRECT_K(0x3d999bc8));
  p = fmaf(s, p, EQUIRECT_K(0x3e2aaaac));
  const float q = s * p;
  const float root = __builtin_amdgcn_sqrtf(s);
  const float t = fmaf(root, q, root);
  const float far = EQUIRECT_K(0x3fc90fdb) - (t + t);
  const float near = fmaf(ay, q, ay);
  return copysignf(small ? near : far, y);
}

// The same two functions for callers that need the angle to a few ulp but not the library's bits: the GGX prefilter,
// whose nearest-pixel lookups are compared with the oracle's through the interval of values reachable when a sample
// within 5e-7 rad of a pixel boundary takes either pixel.  Same polynomials, so the error stays the library's (about
// 1e-7 rad); what goes is everything that only exists to reproduce the library's special cases.
// atan2: both arguments zero give 0 (the library's answer for +0, +0); no test for infinities or NaN.
PBR_DEV float atan2_quick(float y, float x) {
#pragma clang fp contract(fast)
  const float ax = fabsf(x), ay = fabsf(y);
  const float u = fmaxf(fmaxf(ax, ay), 1e-30f), v = fminf(ax, ay);   // (one v_max3_f32)
  const float a = v * rcp_fast(u);
  const float s = a * a;
  float p = fmaf(s, EQUIRECT_K(0x3b2d2a58), EQUIRECT_K(0xbc7a590c));
  p = fmaf(s, p, EQUIRECT_K(0x3d29fb3f));
  p = fmaf(s, p, EQUIRECT_K(0xbd97d4d7));
  p = fmaf(s, p, EQUIRECT_K(0x3dd931b2));
  p = fmaf(s, p, EQUIRECT_K(0xbe1160e6));
  p = fmaf(s, p, EQUIRECT_K(0x3e4cb8bf));
  p = fmaf(s, p, EQUIRECT_K(0xbeaaaa62));
  p = s * p;
  float r = fmaf(a, p, a);
  r = ay > ax ? EQUIRECT_K(0x3fc90fdb) - r : r;
  r = x < 0.0f ? EQUIRECT_K(0x40490fdb) - r : r;
  return copysignf(r, y);
}

// asin for |y| <= 1.  The two branches of the library's evaluation (|y| < 1/2: a polynomial in y^2; otherwise through
// sqrt((1 - |y|) / 2)) are taken per wavefront when all its lanes agree, which the 64 neighbouring directions of a
// wavefront nearly always do (the side faces of a cube look at elevations below 30 degrees over most of their area,
// the top and bottom faces above): the transcendental and five more instructions less on the first, three on the second.
PBR_DEV float asin_quick(float y) {
#pragma clang fp contract(fast)
  const float ay = fabsf(y);
  const bool small = ay < 0.5f;
  const unsigned long long all = __builtin_amdgcn_ballot_w64(true), votes = __builtin_amdgcn_ballot_w64(small);
  auto poly = [](float s) {
    float p = fmaf(s, EQUIRECT_K(0x3d1c21a7), EQUIRECT_K(0x3c5fc5da));
    p = fmaf(s, p, EQUIRECT_K(0x3d034c3c));
    p = fmaf(s, p, EQUIRECT_K(0x3d3641b1));
    p = fmaf(s, p, EQUIRECT_K(0x3d999bc8));
    p = fmaf(s, p, EQUIRECT_K(0x3e2aaaac));
    return s * p;
  };
  if (votes == all) return fmaf(y, poly(y * y), y);
  if (votes == 0ull) {
    const float s = fmaf(ay, -0.5f, 0.5f);
    const float root = __builtin_amdgcn_sqrtf(s);
    const float t = fmaf(root, poly(s), root);
    return copysignf(EQUIRECT_K(0x3fc90fdb) - (t + t), y);
  }
  const float s = small ? y * y : fmaf(ay, -0.5f, 0.5f);
  const float q = poly(s);
  const float root = __builtin_amdgcn_sqrtf(s);
  const float t = fmaf(root, q, root);
  return copysignf(small ? fmaf(ay, q, ay) : EQUIRECT_K(0x3fc90fdb) - (t + t), y);
}

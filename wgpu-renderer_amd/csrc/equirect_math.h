// Cheap evaluations of the two inverse trigonometric functions of the equirect mapping
// (equirectangular.wgsl:68, prefiltermap.wgsl:157-163).  The prefilter only needs the PIXEL a direction falls into,
// floor(u * w) and floor(v * h); these polynomials are within a known distance of the library's atan2f / asinf, so
// they decide the pixel whenever the scaled coordinate is farther than that distance from an integer, and the
// library functions are called only for the few samples that are not (see equirect_pixel_fast).  The result is
// the library's pixel in every case.
#pragma once
#include "devmath.h"

// Largest distance, in radians, seen between these functions and the device library's over 2^33 directions (a
// quarter of them hugging the poles, the seam and the axes; scripts/equirect_approx_check.hip) is 2.4e-7 for atan2
// and 1.2e-7 for asin (one unit in the last place of the largest results, in both cases); the guards are twice that.
#define EQUIRECT_ATAN2_GUARD 5.0e-7f
#define EQUIRECT_ASIN_GUARD 2.5e-7f

// atan2(y, x): octant reduction to a = min / max in [0, 1], odd minimax polynomial of degree 15 (1.4e-7 in f32).
PBR_DEV float atan2_fast(float y, float x) {
  const float ax = fabsf(x), ay = fabsf(y);
  const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
  const float a = mn * rcp_fast(mx);   // 0 / 0 gives NaN, which the caller's guard sends to the library path
  const float s = a * a;
  float p = -0.004054485820233822f;
  p = fmaf(p, s, 0.021862655878067017f);
  p = fmaf(p, s, -0.05591187998652458f);
  p = fmaf(p, s, 0.09642163664102554f);
  p = fmaf(p, s, -0.13908615708351135f);
  p = fmaf(p, s, 0.19946563243865967f);
  p = fmaf(p, s, -0.33329859375953674f);
  p = fmaf(p, s, 0.9999993443489075f);
  float r = a * p;
  r = ay > ax ? 1.57079632679489662f - r : r;
  r = x < 0.0f ? 3.14159265358979324f - r : r;
  return copysignf(r, y);
}

// asin(y), |y| <= 1: y + y^3 R(y^2) for |y| <= 0.5, pi/2 - 2 asin(sqrt((1 - |y|) / 2)) beyond.
PBR_DEV float asin_fast(float y) {
  const float ay = fabsf(y);
  const bool big = ay > 0.5f;
  const float s = big ? fmaf(ay, -0.5f, 0.5f) : ay * ay;
  const float r = big ? __builtin_amdgcn_sqrtf(s) : ay;
  float p = 0.043763067573308945f;
  p = fmaf(p, s, 0.023142263293266296f);
  p = fmaf(p, s, 0.04570779576897621f);
  p = fmaf(p, s, 0.07493077218532562f);
  p = fmaf(p, s, 0.16666822135448456f);
  const float t = fmaf(r * s, p, r);
  return copysignf(big ? fmaf(t, -2.0f, 1.57079632679489662f) : t, y);
}

// Pixel coordinates (before truncation) of direction d in an equirect level of w x h pixels, both powers of two:
// tu = u * w, tv = v * h with (u, v) as in equirectangular.wgsl:68.  Returns false when a coordinate is so close to
// an integer that the cheap functions cannot tell on which side the library's result lies (or is NaN); the caller
// then evaluates the library expression.  When it returns true, (int)tu and (int)tv are the library's pixel.
// The guard: the functions' distance to the library's, scaled like the coordinate, plus the rounding of the
// multiply-add that forms u and v on either path (two half-ulps of a number below 1).
PBR_DEV bool equirect_pixel_fast(float dx, float dy, float dz, float w, float h, float& tu, float& tv) {
  const float u = fmaf(atan2_fast(dz, dx), 0.1591f, 0.5f);
  const float v = fmaf(asin_fast(fminf(fmaxf(dy, -1.0f), 1.0f)), 0.3183f, 0.5f);
  tu = u * w; tv = v * h;
  const float gu = w * (EQUIRECT_ATAN2_GUARD * 0.1591f + 1.2e-7f), gv = h * (EQUIRECT_ASIN_GUARD * 0.3183f + 1.2e-7f);
  const float fu = __builtin_amdgcn_fractf(tu), fv = __builtin_amdgcn_fractf(tv);
  // written so that a NaN fails the test
  return fabsf(fu - 0.5f) < 0.5f - gu && fabsf(fv - 0.5f) < 0.5f - gv;
}

// Small f32 vector helpers shared by the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#define PBR_DEV __device__ __forceinline__

struct V3 { float x, y, z; };

PBR_DEV V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
PBR_DEV V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
PBR_DEV V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
PBR_DEV V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
PBR_DEV V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
PBR_DEV V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
PBR_DEV float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
PBR_DEV V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
// v / |v| with IEEE sqrt and divide: normalize(0) must be NaN, as in WGSL.
PBR_DEV V3 normalize(V3 a) {
  float l = sqrtf(dot(a, a));
  return v3(a.x / l, a.y / l, a.z / l);
}
// Hardware approximations (about 1 ulp): used where the result only feeds 8-bit colour, never an
// integer decision.  rsq(0) = +inf, so a zero vector still normalises to NaN (0 * inf).
PBR_DEV float rcp_fast(float x) { return __builtin_amdgcn_rcpf(x); }
PBR_DEV float rsq_fast(float x) { return __builtin_amdgcn_rsqf(x); }
PBR_DEV float log2_fast(float x) { return __builtin_amdgcn_logf(x); }
PBR_DEV float exp2_fast(float x) { return __builtin_amdgcn_exp2f(x); }
PBR_DEV V3 normalize_fast(V3 a) {
  const float r = rsq_fast(dot(a, a));
  return v3(a.x * r, a.y * r, a.z * r);
}
PBR_DEV bool any_nan(V3 a) { return isnan(a.x) || isnan(a.y) || isnan(a.z); }
PBR_DEV float clamp01(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }
PBR_DEV float pow5(float x) { float x2 = x * x; return x2 * x2 * x; }

// Arithmetic that must give the same bits as a host evaluation: one IEEE operation each, never
// contracted into an FMA (HIP's __fmul_rn / __fadd_rn are plain operators and do get contracted).
PBR_DEV float mul_rn(float a, float b) {
#pragma clang fp contract(off)
  return a * b;
}
PBR_DEV float add_rn(float a, float b) {
#pragma clang fp contract(off)
  return a + b;
}
PBR_DEV float sub_rn(float a, float b) {
#pragma clang fp contract(off)
  return a - b;
}
PBR_DEV float div_rn(float a, float b) { return __fdiv_rn(a, b); }

// o = a * b for column-major 4x4 matrices, each element ((p0 + p1) + p2) + p3.
PBR_DEV void mat4_mul_exact(const float* a, const float* b, float* o) {
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      o[c * 4 + r] = add_rn(add_rn(add_rn(mul_rn(a[r], b[c * 4]), mul_rn(a[4 + r], b[c * 4 + 1])), mul_rn(a[8 + r], b[c * 4 + 2])),
                            mul_rn(a[12 + r], b[c * 4 + 3]));
}
PBR_DEV float4 mat4_vec_exact(const float* m, float x, float y, float z, float w) {
  float4 o;
  o.x = add_rn(add_rn(add_rn(mul_rn(m[0], x), mul_rn(m[4], y)), mul_rn(m[8], z)), mul_rn(m[12], w));
  o.y = add_rn(add_rn(add_rn(mul_rn(m[1], x), mul_rn(m[5], y)), mul_rn(m[9], z)), mul_rn(m[13], w));
  o.z = add_rn(add_rn(add_rn(mul_rn(m[2], x), mul_rn(m[6], y)), mul_rn(m[10], z)), mul_rn(m[14], w));
  o.w = add_rn(add_rn(add_rn(mul_rn(m[3], x), mul_rn(m[7], y)), mul_rn(m[11], z)), mul_rn(m[15], w));
  return o;
}

PBR_DEV uint32_t to_unorm8(float x) {
  // clamp to [0,1], NaN -> 0, round to nearest even
  if (!(x > 0.0f)) return 0u;
  x = fminf(x, 1.0f);
  return (uint32_t)__float2int_rn(x * 255.0f);
}

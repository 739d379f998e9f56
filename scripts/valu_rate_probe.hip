// Measures how many vector instructions per cycle one SIMD of the device retires for a few instruction kinds, with
// 1, 2, 4 and 8 resident waves per SIMD: the yardstick for "VALU issue" figures derived from SQ_INSTS_VALU.
// Build: hipcc -O3 --offload-arch=gfx950 scripts/valu_rate_probe.hip -o build/valu_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int kIters = 4096, kChains = 16;

template <int KIND>
__global__ void __launch_bounds__(256) probe(float* out, float seed, long long* cycles) {
  float a[kChains];
  f2 p[kChains / 2];
  uint32_t u[kChains];
  unsigned long long w[kChains / 2];
  const unsigned long long mask = 0x5555555555555555ull;
  for (int i = 0; i < kChains; ++i) { a[i] = seed + (float)(threadIdx.x + i); u[i] = (uint32_t)(threadIdx.x * 7 + i); }
  for (int i = 0; i < kChains / 2; ++i) { p[i] = (f2){a[2 * i], a[2 * i + 1]}; w[i] = u[i]; }
  if constexpr (KIND == 31 || KIND == 3 || KIND == 25) asm volatile("s_mov_b64 vcc, %0" : : "s"(mask) : "vcc");
  const long long t0 = clock64();
  for (int it = 0; it < kIters; ++it) {
#pragma unroll
    for (int i = 0; i < kChains; ++i) {
      if constexpr (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(seed));
      if constexpr (KIND == 1) { if (i < kChains / 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i]) : "v"(p[(i + 1) % (kChains / 2)])); }
      if constexpr (KIND == 2) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
      if constexpr (KIND == 3) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(seed) : );
      if constexpr (KIND == 4) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) % kChains]));
      if constexpr (KIND == 5) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(u[i]) : "v"(u[(i + 1) % kChains]));
      if constexpr (KIND == 6) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a[i]));
      if constexpr (KIND == 7) asm volatile("v_fract_f32 %0, %0" : "+v"(a[i]));
      if constexpr (KIND == 8) asm volatile("v_bfi_b32 %0, %1, %0, %0" : "+v"(u[i]) : "v"(u[(i + 1) % kChains]));
      if constexpr (KIND == 9) asm volatile("v_max_f32 %0, |%0|, |%1|" : "+v"(a[i]) : "v"(seed));
      if constexpr (KIND == 10) asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(seed) : "vcc");
      if constexpr (KIND == 11) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) % kChains]));
      if constexpr (KIND == 12) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(seed), "s"(mask));
      if constexpr (KIND == 13) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(seed));
      if constexpr (KIND == 14) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(seed));
      if constexpr (KIND == 15) asm volatile("v_fmaak_f32 %0, %0, %1, 0x3f7ffff5" : "+v"(a[i]) : "v"(seed));
      if constexpr (KIND == 16) asm volatile("v_mad_u64_u32 %0, vcc, %1, 12, %0" : "+v"(w[i / 2]) : "v"(u[i]) : "vcc");
      if constexpr (KIND == 17) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(u[i]) : "v"(u[(i + 1) % kChains]));
      if constexpr (KIND == 18) asm volatile("v_med3_f32 %0, %0, -1.0, 1.0" : "+v"(a[i]));
      if constexpr (KIND == 19) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
      if constexpr (KIND == 20) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) % kChains]));
      if constexpr (KIND == 21) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) % kChains]));
      if constexpr (KIND == 22) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "s"(seed));
      if constexpr (KIND == 23) { if (i & 1) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(seed)); else asm volatile("v_and_b32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) % kChains])); }
      if constexpr (KIND == 24) { if (i & 3) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(seed)); else asm volatile("v_and_b32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) % kChains])); }
      if constexpr (KIND == 25) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a[i]) : "v"(seed), "v"(a[(i + 1) % kChains]));
      if constexpr (KIND == 26) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(u[i]));
      if constexpr (KIND == 27) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
      if constexpr (KIND == 28) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i / 2]) : "v"(p[(i / 2 + 1) % (kChains / 2)]));
      if constexpr (KIND == 29) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(a[i]) : "v"(u[i]));
      if constexpr (KIND == 31) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(seed));
      if constexpr (KIND == 32) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(seed) : "vcc");
      if constexpr (KIND == 33) asm volatile("v_cmp_gt_f32 %2, %0, %1\n\tv_cndmask_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(seed), "s"(mask));
      if constexpr (KIND == 34) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n\tv_fma_f32 %2, %2, %1, %2\n\tv_fma_f32 %3, %3, %1, %3\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]), "+v"(a[(i + 5) % kChains]), "+v"(a[(i + 9) % kChains]) : "v"(seed) : "vcc");
      if constexpr (KIND == 40) { asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(seed)); }
      if constexpr (KIND == 41) { asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(seed)); }
      if constexpr (KIND == 42) { asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(a[(i + 1) % kChains])); }
      if constexpr (KIND == 43) { asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(seed)); }
      if constexpr (KIND == 44) { asm volatile("v_or_b32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) % kChains])); }
      if constexpr (KIND == 45) { asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) % kChains])); }
      if constexpr (KIND == 46) { asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(u[i])); }
      if constexpr (KIND == 47) { asm volatile("v_ashrrev_i32 %0, 3, %0" : "+v"(u[i])); }
      if constexpr (KIND == 48) { asm volatile("v_min_i32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) % kChains])); }
      if constexpr (KIND == 49) { asm volatile("v_max_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) % kChains])); }
      if constexpr (KIND == 50) { asm volatile("v_sub_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) % kChains])); }
      if constexpr (KIND == 51) { asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(a[i]) : "v"(seed)); }
      if constexpr (KIND == 52) { asm volatile("v_fma_f32 %0, -%0, %1, %0" : "+v"(a[i]) : "v"(seed)); }
      if constexpr (KIND == 53) { asm volatile("v_fma_f32 %0, |%0|, %1, %0" : "+v"(a[i]) : "v"(seed)); }
      if constexpr (KIND == 54) { asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i]) : "s"(seed)); }
      if constexpr (KIND == 55) { asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(a[i])); }
      if constexpr (KIND == 56) { asm volatile("v_mul_f32 %0, 0x3e22f983, %0" : "+v"(a[i])); }
      if constexpr (KIND == 57) { asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(a[i]) : "v"(u[i])); }
      if constexpr (KIND == 58) { asm volatile("v_cvt_u32_f32 %0, %1" : "=v"(u[i]) : "v"(a[i])); }
      if constexpr (KIND == 59) { asm volatile("v_floor_f32 %0, %0" : "+v"(a[i])); }
      if constexpr (KIND == 60) { asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(u[i])); }
      if constexpr (KIND == 61) { asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(u[i]) : "v"(u[(i + 1) % kChains])); }
      if constexpr (KIND == 62) { asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(u[i]) : "v"(u[(i + 1) % kChains])); }
      if constexpr (KIND == 63) { asm volatile("v_cmp_gt_f32 %2, %0, %1" : : "v"(a[i]), "v"(seed), "s"(mask)); }
      if constexpr (KIND == 64) { asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(seed) : "vcc"); }
      if constexpr (KIND == 65) { asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "s"(seed)); }
      if constexpr (KIND == 66) { asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(seed)); }
      if constexpr (KIND == 67) { asm volatile("v_and_b32 %0, 0x00ff00ff, %0" : "+v"(u[i])); }
      if constexpr (KIND == 68) { asm volatile("v_and_b32 %0, 63, %0" : "+v"(u[i])); }
      if constexpr (KIND == 69) { asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(a[i]) : "v"(u[i])); }
      if constexpr (KIND == 70) { asm volatile("v_rsq_f32 %0, %0" : "+v"(a[i])); }
      if constexpr (KIND == 71) { asm volatile("v_log_f32 %0, %0" : "+v"(a[i])); }
      if constexpr (KIND == 72) { asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(a[i]) : "v"(u[i])); }
      if constexpr (KIND == 73) { asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(u[i]) : "v"(u[(i + 1) % kChains])); }
      if constexpr (KIND == 74) { asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i / 2]) : "v"(p[(i / 2 + 1) % (kChains / 2)])); }
      if constexpr (KIND == 90) { asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(seed), "v"(a[(i + 1) % kChains])); }
      if constexpr (KIND == 91) { asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[(i + 1) % kChains]), "v"(seed), "v"(a[(i + 2) % kChains])); }
      if constexpr (KIND == 92) { asm volatile("v_cmp_gt_f32 vcc, %0, %3\n\tv_cndmask_b32 %0, %0, %3, vcc\n\tv_cndmask_b32 %1, %1, %3, vcc\n\tv_cndmask_b32 %2, %2, %3, vcc" : "+v"(a[i]), "+v"(a[(i + 5) % kChains]), "+v"(a[(i + 9) % kChains]) : "v"(seed) : "vcc"); }
      if constexpr (KIND == 93) { asm volatile("v_cmp_gt_f32 %4, %0, %3\n\tv_cndmask_b32 %0, %0, %3, %4\n\tv_cndmask_b32 %1, %1, %3, %4\n\tv_cndmask_b32 %2, %2, %3, %4" : "+v"(a[i]), "+v"(a[(i + 5) % kChains]), "+v"(a[(i + 9) % kChains]) : "v"(seed), "s"(mask)); }
      if constexpr (KIND == 94) { asm volatile("v_cmp_gt_f32 vcc, %0, %2\n\tv_cndmask_b32 %0, %0, %2, vcc\n\tv_cndmask_b32 %1, %1, %2, vcc" : "+v"(a[i]), "+v"(a[(i + 5) % kChains]) : "v"(seed) : "vcc"); }
      if constexpr (KIND == 95) { asm volatile("v_mul_f32 %0, |%0|, %1" : "+v"(a[i]) : "v"(seed)); }
      if constexpr (KIND == 96) { asm volatile("v_add_f32 %0, -%0, %1" : "+v"(a[i]) : "v"(seed)); }
      if constexpr (KIND == 97) { asm volatile("v_fma_f32 %0, %0, %1, 0.5" : "+v"(a[i]) : "v"(seed)); }
      if constexpr (KIND == 98) { asm volatile("v_fmamk_f32 %0, %0, 0x3e22f983, %1" : "+v"(a[i]) : "v"(seed)); }
      if constexpr (KIND == 99) { asm volatile("v_subrev_f32 %0, %0, %1" : "+v"(a[i]) : "v"(seed)); }
      if constexpr (KIND == 100) { asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(a[i]) : "v"(u[i])); }
      if constexpr (KIND == 101) { asm volatile("v_rndne_f32 %0, %0" : "+v"(a[i])); }
      if constexpr (KIND == 102) { asm volatile("v_trunc_f32 %0, %0" : "+v"(a[i])); }
      if constexpr (KIND == 103) { asm volatile("v_mul_f32 %0, %0, %1 clamp" : "+v"(a[i]) : "v"(seed)); }
      if constexpr (KIND == 104) { asm volatile("v_and_b32 %0, %1, %0" : "+v"(u[i]) : "s"((uint32_t)mask)); }
      if constexpr (KIND == 105) { asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "s"(seed)); }
      if constexpr (KIND == 106) { asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(u[i]) : "v"(u[(i + 1) % kChains])); }
      if constexpr (KIND == 107) { asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(u[i]) : "v"(u[(i + 1) % kChains])); }
      if constexpr (KIND == 108) { asm volatile("v_mul_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(seed)); }
      if constexpr (KIND == 109) { asm volatile("v_cndmask_b32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(seed), "s"(mask)); }
      if constexpr (KIND == 110) { asm volatile("v_cvt_i32_f32 %1, %0\n\tv_fma_f32 %0, %0, %2, %0" : "+v"(a[i]), "=v"(u[i]) : "v"(seed)); }
      if constexpr (KIND == 111) { asm volatile("v_rcp_f32 %0, %0\n\tv_fma_f32 %1, %1, %3, %1\n\tv_fma_f32 %2, %2, %3, %2\n\tv_fma_f32 %1, %1, %3, %1" : "+v"(a[i]), "+v"(a[(i + 5) % kChains]), "+v"(a[(i + 9) % kChains]) : "v"(seed)); }
      if constexpr (KIND == 130) { asm volatile("v_cndmask_b32 %0, %0, %3, vcc\n\tv_fma_f32 %1, %1, %3, %1\n\tv_fma_f32 %2, %2, %3, %2\n\tv_fma_f32 %1, %1, %3, %1" : "+v"(a[i]), "+v"(a[(i + 5) % kChains]), "+v"(a[(i + 9) % kChains]) : "v"(seed)); }
      if constexpr (KIND == 131) { asm volatile("v_cndmask_b32 %0, %0, %2, vcc\n\tv_fma_f32 %1, %1, %2, %1" : "+v"(a[i]), "+v"(a[(i + 5) % kChains]) : "v"(seed)); }
      if constexpr (KIND == 132) { asm volatile("v_cmp_gt_f32 vcc, %0, %3\n\tv_cndmask_b32 %0, %0, %3, vcc\n\tv_fma_f32 %1, %1, %3, %1\n\tv_cndmask_b32 %1, %1, %3, vcc\n\tv_fma_f32 %2, %2, %3, %2\n\tv_cndmask_b32 %2, %2, %3, vcc\n\tv_fma_f32 %1, %1, %3, %1" : "+v"(a[i]), "+v"(a[(i + 5) % kChains]), "+v"(a[(i + 9) % kChains]) : "v"(seed) : "vcc"); }
      if constexpr (KIND == 133) { asm volatile("v_cmp_gt_f32 vcc, %0, %3\n\tv_cndmask_b32_e64 %0, %0, %3, vcc\n\tv_cndmask_b32_e64 %1, %1, %3, vcc\n\tv_cndmask_b32_e64 %2, %2, %3, vcc" : "+v"(a[i]), "+v"(a[(i + 5) % kChains]), "+v"(a[(i + 9) % kChains]) : "v"(seed) : "vcc"); }
      if constexpr (KIND == 134) { asm volatile("v_cmp_gt_f32 vcc, %0, %3\n\tv_cndmask_b32 %0, %0, %3, vcc\n\tv_fma_f32 %1, %1, %3, %1\n\tv_fma_f32 %2, %2, %3, %2\n\tv_fma_f32 %1, %1, %3, %1\n\tv_fma_f32 %2, %2, %3, %2\n\tv_fma_f32 %1, %1, %3, %1\n\tv_fma_f32 %2, %2, %3, %2" : "+v"(a[i]), "+v"(a[(i + 5) % kChains]), "+v"(a[(i + 9) % kChains]) : "v"(seed) : "vcc"); }
      if constexpr (KIND == 135) { asm volatile("v_fma_f32 %0, %0, %3, %0\n\tv_fma_f32 %0, %0, %3, %0\n\tv_fma_f32 %1, %1, %3, %1\n\tv_fma_f32 %2, %2, %3, %2\n\tv_fma_f32 %1, %1, %3, %1\n\tv_fma_f32 %2, %2, %3, %2\n\tv_fma_f32 %1, %1, %3, %1\n\tv_fma_f32 %2, %2, %3, %2" : "+v"(a[i]), "+v"(a[(i + 5) % kChains]), "+v"(a[(i + 9) % kChains]) : "v"(seed)); }
      if constexpr (KIND == 30) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf" : "=v"(a[i]) : "v"(a[(i + 1) % kChains]));
    }
  }
  const long long t1 = clock64();
  float s = 0; uint32_t su = 0;
  for (int i = 0; i < kChains; ++i) { s += a[i]; su += u[i]; }
  for (int i = 0; i < kChains / 2; ++i) { s += p[i].x + p[i].y; su += (uint32_t)w[i]; }
  out[blockIdx.x * 256 + threadIdx.x] = s + (float)su;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

template <int KIND>
void run(const char* name, int per_iter, float* out, long long* d_cyc) {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  for (int waves_per_simd : {1, 4, 8}) {
    const int blocks = cus * waves_per_simd;     // 256 threads = 4 waves = one per SIMD of a CU, if the blocks spread evenly
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    probe<KIND><<<blocks, 256>>>(out, 1.0f, d_cyc);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    probe<KIND><<<blocks, 256>>>(out, 1.0f, d_cyc);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    long long cyc; CHECK(hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost));
    const double instr = (double)kIters * per_iter;   // per wave
    std::printf("{\"kind\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.4f, \"wave0_ticks\": %lld, \"instr_per_wave\": %.0f, \"ginstr_per_s_per_simd\": %.3f}\n",
                name, waves_per_simd, ms, cyc, instr, instr * waves_per_simd / (ms * 1e-3) * 1e-9);
  }
}

int main() {
  float* out; long long* d_cyc;
  CHECK(hipMalloc(&out, sizeof(float) * 256 * 256 * 8)); CHECK(hipMalloc(&d_cyc, 8));
  run<0>("v_fma_f32", kChains, out, d_cyc);
  run<1>("v_pk_fma_f32", kChains / 2, out, d_cyc);
  run<2>("v_rcp_f32", kChains, out, d_cyc);
  run<3>("v_cndmask_b32", kChains, out, d_cyc);
  run<4>("v_mul_u32_u24", kChains, out, d_cyc);
  run<5>("v_lshl_add_u32", kChains, out, d_cyc);
  run<6>("v_cvt_i32_f32", kChains, out, d_cyc);
  run<7>("v_fract_f32", kChains, out, d_cyc);
  run<8>("v_bfi_b32", kChains, out, d_cyc);
  run<9>("v_max_f32_abs", kChains, out, d_cyc);
  run<10>("v_cmp_gt_f32", kChains, out, d_cyc);
  run<11>("v_mul_lo_u32", kChains, out, d_cyc);
  run<12>("v_cndmask_b32 sgpr mask", kChains, out, d_cyc);
  run<13>("v_add_f32", kChains, out, d_cyc);
  run<14>("v_mul_f32", kChains, out, d_cyc);
  run<15>("v_fmaak_f32", kChains, out, d_cyc);
  run<16>("v_mad_u64_u32", kChains, out, d_cyc);
  run<17>("v_add3_u32", kChains, out, d_cyc);
  run<18>("v_med3_f32", kChains, out, d_cyc);
  run<19>("v_sqrt_f32", kChains, out, d_cyc);
  run<20>("v_and_b32", kChains, out, d_cyc);
  run<21>("v_add_u32", kChains, out, d_cyc);
  run<22>("v_fma_f32 sgpr operand", kChains, out, d_cyc);
  run<23>("fma and and_b32 alternating", kChains, out, d_cyc);
  run<24>("three fma to one and_b32", kChains, out, d_cyc);
  run<25>("v_cndmask_b32 vcc, separate destination", kChains, out, d_cyc);
  run<26>("v_lshrrev_b32", kChains, out, d_cyc);
  run<27>("v_exp_f32", kChains, out, d_cyc);
  run<28>("v_pk_mul_f32", kChains, out, d_cyc);
  run<29>("v_cvt_f32_ubyte1", kChains, out, d_cyc);
  run<30>("v_mov_b32_dpp quad_perm", kChains, out, d_cyc);
  run<31>("v_cndmask_b32_e64 vcc", kChains, out, d_cyc);
  run<32>("v_cmp vcc + v_cndmask vcc pair", 2 * kChains, out, d_cyc);
  run<33>("v_cmp sgpr + v_cndmask sgpr pair", 2 * kChains, out, d_cyc);
  run<34>("v_cmp vcc, 2 fma, v_cndmask vcc", 4 * kChains, out, d_cyc);
  run<40>("v_max_f32", kChains, out, d_cyc);
  run<41>("v_min_f32", kChains, out, d_cyc);
  run<42>("v_mov_b32", kChains, out, d_cyc);
  run<43>("v_sub_f32", kChains, out, d_cyc);
  run<44>("v_or_b32", kChains, out, d_cyc);
  run<45>("v_xor_b32", kChains, out, d_cyc);
  run<46>("v_lshlrev_b32", kChains, out, d_cyc);
  run<47>("v_ashrrev_i32", kChains, out, d_cyc);
  run<48>("v_min_i32", kChains, out, d_cyc);
  run<49>("v_max_u32", kChains, out, d_cyc);
  run<50>("v_sub_u32", kChains, out, d_cyc);
  run<51>("v_fmac_f32", kChains, out, d_cyc);
  run<52>("v_fma_f32 neg modifier", kChains, out, d_cyc);
  run<53>("v_fma_f32 abs modifier", kChains, out, d_cyc);
  run<54>("v_add_f32 sgpr operand", kChains, out, d_cyc);
  run<55>("v_add_f32 inline constant", kChains, out, d_cyc);
  run<56>("v_mul_f32 literal", kChains, out, d_cyc);
  run<57>("v_cvt_f32_u32", kChains, out, d_cyc);
  run<58>("v_cvt_u32_f32", kChains, out, d_cyc);
  run<59>("v_floor_f32", kChains, out, d_cyc);
  run<60>("v_bfe_u32", kChains, out, d_cyc);
  run<61>("v_and_or_b32", kChains, out, d_cyc);
  run<62>("v_perm_b32", kChains, out, d_cyc);
  run<63>("v_cmp_gt_f32 sgpr destination", kChains, out, d_cyc);
  run<64>("v_cmp_gt_f32_e32 after init", kChains, out, d_cyc);
  run<65>("v_mul_f32 sgpr operand", kChains, out, d_cyc);
  run<66>("v_max3_f32", kChains, out, d_cyc);
  run<67>("v_and_b32 literal", kChains, out, d_cyc);
  run<68>("v_and_b32 inline constant", kChains, out, d_cyc);
  run<69>("v_cvt_f32_ubyte0", kChains, out, d_cyc);
  run<70>("v_rsq_f32", kChains, out, d_cyc);
  run<71>("v_log_f32", kChains, out, d_cyc);
  run<72>("v_ldexp_f32", kChains, out, d_cyc);
  run<73>("v_mad_u32_u24", kChains, out, d_cyc);
  run<74>("v_pk_add_f32", kChains, out, d_cyc);
  run<90>("v_fmac_f32 distinct sources", 1 * kChains, out, d_cyc);
  run<91>("v_fma_f32 four distinct registers", 1 * kChains, out, d_cyc);
  run<92>("v_cmp vcc + 3 v_cndmask vcc", 4 * kChains, out, d_cyc);
  run<93>("v_cmp sgpr + 3 v_cndmask sgpr", 4 * kChains, out, d_cyc);
  run<94>("v_cmp vcc + 2 v_cndmask vcc", 3 * kChains, out, d_cyc);
  run<95>("v_mul_f32 abs modifier", 1 * kChains, out, d_cyc);
  run<96>("v_add_f32 neg modifier", 1 * kChains, out, d_cyc);
  run<97>("v_fma_f32 inline constant", 1 * kChains, out, d_cyc);
  run<98>("v_fmamk_f32", 1 * kChains, out, d_cyc);
  run<99>("v_subrev_f32", 1 * kChains, out, d_cyc);
  run<100>("v_cvt_f32_i32", 1 * kChains, out, d_cyc);
  run<101>("v_rndne_f32", 1 * kChains, out, d_cyc);
  run<102>("v_trunc_f32", 1 * kChains, out, d_cyc);
  run<103>("v_mul_f32 clamp", 1 * kChains, out, d_cyc);
  run<104>("v_and_b32 sgpr operand", 1 * kChains, out, d_cyc);
  run<105>("v_mov_b32 from sgpr", 1 * kChains, out, d_cyc);
  run<106>("v_lshlrev_b32 by register", 1 * kChains, out, d_cyc);
  run<107>("v_lshrrev_b32 by register", 1 * kChains, out, d_cyc);
  run<108>("v_mul_f32 + v_add_f32 pair", 2 * kChains, out, d_cyc);
  run<109>("v_cndmask sgpr mask, fma between", 2 * kChains, out, d_cyc);
  run<110>("v_cvt_i32_f32, fma between", 2 * kChains, out, d_cyc);
  run<111>("v_rcp_f32, 3 fma between", 4 * kChains, out, d_cyc);
  run<130>("v_cndmask vcc, 3 fma between, no compare", 4 * kChains, out, d_cyc);
  run<131>("v_cndmask vcc, 1 fma between, no compare", 2 * kChains, out, d_cyc);
  run<132>("v_cmp vcc; (v_cndmask vcc, fma) x3", 7 * kChains, out, d_cyc);
  run<133>("v_cmp vcc; 3 v_cndmask_e64 vcc", 4 * kChains, out, d_cyc);
  run<134>("v_cmp vcc; v_cndmask vcc; 6 fma", 8 * kChains, out, d_cyc);
  run<135>("8 fma (reference for the mixes)", 8 * kChains, out, d_cyc);
  return 0;
}

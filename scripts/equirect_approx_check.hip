// Measures, on the GPU, how far atan2_fast / asin_fast (csrc/equirect_math.h) are from the device library's
// atan2f / asinf, and checks the property the prefilter relies on: whenever equirect_pixel_fast says a pixel is
// decided, it is the pixel the library expression gives, for every power-of-two level size.
// build: hipcc -O3 --offload-arch=gfx950 scripts/equirect_approx_check.hip -o build/equirect_approx_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "../wgpu-renderer_amd/csrc/equirect_math.h"

__device__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ float unit(uint32_t h) { return (float)(h >> 8) * (1.0f / 8388608.0f) - 1.0f; }   // [-1, 1)

struct Report { float max_atan, max_asin; unsigned long long undecided[13], wrong, total; };

__global__ void check_kernel(Report* rep, unsigned long long n_per_thread, uint32_t seed) {
  const unsigned long long gid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  float max_atan = 0.0f, max_asin = 0.0f;
  unsigned long long undecided[13] = {}, wrong = 0;
  for (unsigned long long k = 0; k < n_per_thread; ++k) {
    const uint32_t base = hash32((uint32_t)(gid * n_per_thread + k) ^ seed);
    float x = unit(hash32(base + 1u)), y = unit(hash32(base + 2u)), z = unit(hash32(base + 3u));
    const uint32_t kind = base & 15u;
    // a share of the directions hugs the places where the mapping is ill-conditioned: the poles, the seam, the axes
    if (kind == 0u) { x *= 1e-4f; z *= 1e-4f; }
    if (kind == 1u) { z *= 1e-5f; x = -fabsf(x); }
    if (kind == 2u) y *= 1e-3f;
    if (kind == 3u) { x *= 1e-6f; }
    const float inv = rsqrtf(x * x + y * y + z * z);
    x *= inv; y *= inv; z *= inv;
    const float ref_a = atan2f(z, x), ref_s = asinf(fminf(fmaxf(y, -1.0f), 1.0f));
    const float da = fabsf(atan2_fast(z, x) - ref_a), ds = fabsf(asin_fast(fminf(fmaxf(y, -1.0f), 1.0f)) - ref_s);
    if (da == da) max_atan = fmaxf(max_atan, da);
    if (ds == ds) max_asin = fmaxf(max_asin, ds);
    const float u = ref_a * 0.1591f + 0.5f, v = ref_s * 0.3183f + 0.5f;
    for (int l = 0; l < 13; ++l) {
      const float w = (float)(4096 >> l), h = fmaxf((float)(2048 >> l), 1.0f);
      float tu, tv;
      if (!equirect_pixel_fast(x, y, z, w, h, tu, tv)) { ++undecided[l]; continue; }
      if ((int)tu != (int)(u * w) || (int)tv != (int)(v * h)) ++wrong;
    }
  }
  atomicMax(reinterpret_cast<int*>(&rep->max_atan), __float_as_int(max_atan));   // non-negative floats order like ints
  atomicMax(reinterpret_cast<int*>(&rep->max_asin), __float_as_int(max_asin));
  for (int l = 0; l < 13; ++l) atomicAdd(&rep->undecided[l], undecided[l]);
  atomicAdd(&rep->wrong, wrong);
  atomicAdd(&rep->total, n_per_thread);
}

int main() {
  Report* d; Report h = {};
  if (hipMalloc(&d, sizeof h) != hipSuccess) { std::printf("no device\n"); return 2; }
  (void)hipMemset(d, 0, sizeof h);
  const int blocks = 256 * 16, threads = 256;
  for (uint32_t pass = 0; pass < 8; ++pass) {
    hipLaunchKernelGGL(check_kernel, dim3(blocks), dim3(threads), 0, 0, d, 1024ull, 0x9e3779b9u * (pass + 1));
    if (hipDeviceSynchronize() != hipSuccess) { std::printf("kernel failed\n"); return 3; }
  }
  (void)hipMemcpy(&h, d, sizeof h, hipMemcpyDeviceToHost);
  std::printf("{\"directions\": %llu, \"max_abs_diff_atan2\": %.4g, \"max_abs_diff_asin\": %.4g, \"wrong_pixels\": %llu, \"undecided_share_by_level\": [",
              h.total, h.max_atan, h.max_asin, h.wrong);
  for (int l = 0; l < 13; ++l) std::printf("%s%.3g", l ? ", " : "", (double)h.undecided[l] / (double)h.total);
  std::printf("]}\n");
  return h.wrong ? 1 : 0;
}

// Compares, on the GPU, atan2_regular / asin_regular (csrc/equirect_math.h) with the device library's atan2f / asinf
// bit for bit over 2^33 directions, a quarter of them hugging the poles, the seam and the axes, plus raw argument
// pairs over many orders of magnitude; counts the arguments equirect_regular hands to the library.
// build: hipcc -O3 --offload-arch=gfx950 scripts/equirect_exact_check.hip -o build/equirect_exact_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "../wgpu-renderer_amd/csrc/equirect_math.h"

__device__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ float unit(uint32_t h) { return (float)(h >> 8) * (1.0f / 8388608.0f) - 1.0f; }   // [-1, 1)

struct Report { unsigned long long atan_differs, asin_differs, irregular, total; uint32_t first_bad[4]; };

__global__ void check_kernel(Report* rep, unsigned long long n_per_thread, uint32_t seed) {
  const unsigned long long gid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long bad_a = 0, bad_s = 0, irregular = 0;
  for (unsigned long long k = 0; k < n_per_thread; ++k) {
    const uint32_t base = hash32((uint32_t)(gid * n_per_thread + k) ^ seed);
    float x = unit(hash32(base + 1u)), y = unit(hash32(base + 2u)), z = unit(hash32(base + 3u));
    const uint32_t kind = base & 15u;
    if (kind == 0u) { x *= 1e-4f; z *= 1e-4f; }
    if (kind == 1u) { z *= 1e-5f; x = -fabsf(x); }
    if (kind == 2u) y *= 1e-3f;
    if (kind == 3u) { x *= 1e-6f; }
    if (kind == 4u) {
      // raw arguments: any exponent within the regular range, zeros and exact ties included
      x = ldexpf(x, (int)(hash32(base + 4u) % 100u) - 50); z = ldexpf(z, (int)(hash32(base + 5u) % 100u) - 50);
      if ((base & 0x300u) == 0u) z = 0.0f;
      if ((base & 0x300u) == 0x100u) x = copysignf(z, x);
      if ((base & 0xf00u) == 0x200u) x = 0.0f;
    } else {
      const float inv = rsqrtf(x * x + y * y + z * z);
      x *= inv; y *= inv; z *= inv;
    }
    const float yc = fminf(fmaxf(y, -1.0f), 1.0f);
    if (!equirect_regular(z, x)) { ++irregular; }
    else if (__float_as_uint(atan2_regular(z, x)) != __float_as_uint(atan2f(z, x))) {
      if (bad_a++ == 0) { rep->first_bad[0] = __float_as_uint(z); rep->first_bad[1] = __float_as_uint(x); }
    }
    if (__float_as_uint(asin_regular(yc)) != __float_as_uint(asinf(yc))) {
      if (bad_s++ == 0) rep->first_bad[2] = __float_as_uint(yc);
    }
  }
  atomicAdd(&rep->atan_differs, bad_a);
  atomicAdd(&rep->asin_differs, bad_s);
  atomicAdd(&rep->irregular, irregular);
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
  std::printf("{\"directions\": %llu, \"atan2_results_that_differ\": %llu, \"asin_results_that_differ\": %llu, \"handed_to_the_library\": %llu, "
              "\"first_differing_arguments_hex\": [\"%08x\", \"%08x\", \"%08x\"]}\n",
              h.total, h.atan_differs, h.asin_differs, h.irregular, h.first_bad[0], h.first_bad[1], h.first_bad[2]);
  return (h.atan_differs || h.asin_differs) ? 1 : 0;
}

// read_bw.hip -- what does this box deliver to a kernel that only READS, 1 KiB per wave and load, the way
// find_kernel's units arrive?  (GPU box: hipcc --offload-arch=gfx950 -O3 -o read_bw read_bw.hip && ./read_bw)
// A buffer of the given size is read `passes` times by 2 workgroups of 1024 threads per CU (find_kernel's
// residency); every wave walks its own pseudo-random sequence of 1 KiB units (one global_load_dwordx4 per lane, `depth`
// loads in flight per wave).  Sizes: 253 MB (the postings of the Geonames-scale image: beside a 256 MiB Infinity
// Cache), 2 GiB (HBM only), 32 MB (L2 / Infinity Cache resident).  The figure roofline.frac_of_achievable in
// bench.py is held against is the guide's 6.29 TB/s COPY; this prints what reads alone reach.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

template <int DEPTH>
__global__ __launch_bounds__(1024, 8) void k(const uint4* __restrict__ buf, uint64_t n_units, uint32_t units_per_wave,
                                             uint32_t* out) {
  const uint32_t lane = threadIdx.x & 63;
  const uint64_t wave = uint64_t(blockIdx.x) * 16 + (threadIdx.x >> 6);
  uint32_t x = __builtin_amdgcn_readfirstlane(uint32_t(wave * 0x9E3779B9u + 12345u));
  uint32_t acc = 0;
  for (uint32_t i = 0; i < units_per_wave; i += DEPTH) {
    uint4 v[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      x = x * 1664525u + 1013904223u;                          // (scalar: one multiply-high per unit, no division)
      const uint64_t u = __umulhi(x, uint32_t(n_units));
      v[d] = buf[u * 64 + lane];
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) acc += v[d].x ^ v[d].y ^ v[d].z ^ v[d].w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}

template <int DEPTH>
static void run(const uint4* d, uint64_t bytes, int n_cus, uint32_t* out) {
  const uint64_t n_units = bytes / 1024;
  const int grid = n_cus * 2;
  const uint32_t upw = 4096;                                   // 4 MiB per wave
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<DEPTH>, dim3(grid), dim3(1024), 0, 0, d, n_units, upw, out);
  (void)hipEventRecord(e0, 0);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k<DEPTH>, dim3(grid), dim3(1024), 0, 0, d, n_units, upw, out);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double moved = 3.0 * double(grid) * 16 * upw * 1024;
  printf("buffer %7.0f MB  %d load(s) in flight per wave  %8.2f ms  %6.2f TB/s\n", bytes / 1e6, DEPTH, ms / 3,
         moved / (ms * 1e-3) / 1e12);
}

int main() {
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  printf("%s, %d CUs; random 1 KiB units, 2 x 1024 threads per CU\n", p.name, p.multiProcessorCount);
  const uint64_t sizes[] = {32ull << 20, 253000000ull, 327000000ull, 2048ull << 20};
  uint32_t* out;
  (void)hipMalloc(&out, 4);
  for (uint64_t bytes : sizes) {
    uint4* d;
    if (hipMalloc(&d, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(d, 1, bytes);
    run<1>(d, bytes, p.multiProcessorCount, out);
    run<2>(d, bytes, p.multiProcessorCount, out);
    run<4>(d, bytes, p.multiProcessorCount, out);
    (void)hipFree(d);
  }
  return 0;
}

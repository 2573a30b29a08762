// lds_atomic_rate.hip -- what does a no-return LDS atomic cost on MI355X?  (GPU box: hipcc --offload-arch=gfx950
// -O3 -o lds_atomic_rate lds_atomic_rate.hip && ./lds_atomic_rate)
// Every wave issues ds_add_u32 (result unused) on a 64 KiB counter array, addresses conflict-free within an
// instruction (lane l -> word base + l, consecutive banks) or random; 2 workgroups of 1024 threads per CU as
// find_kernel runs.  Prints lanes per clock per CU (at the clock rate measured with s_memtime) -- the ceiling
// bench.py's roofline.lds line is held against.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(1024, 8) void k(uint32_t* out, int iters, unsigned long long* clocks) {
  __shared__ uint32_t cnt[16384];
  for (int i = threadIdx.x; i < 16384; i += 1024) cnt[i] = 0;
  __syncthreads();
  const uint32_t lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  uint32_t x = threadIdx.x * 2654435761u + blockIdx.x;
  const unsigned long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      uint32_t w;
      if (MODE == 0) w = (wid * 1024 + it * 64 + j * 8192 + lane) & 16383;        // lane -> consecutive words
      else { x = x * 1664525u + 1013904223u; w = (x >> 10) & 16383; }               // random words
      __hip_atomic_fetch_add(&cnt[w], 1u << ((j & 3) * 8), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  __syncthreads();
  const unsigned long long t1 = clock64();
  if (threadIdx.x == 0) clocks[blockIdx.x] = t1 - t0;
  uint32_t s = 0;
  for (int i = threadIdx.x; i < 16384; i += 1024) s += cnt[i];
  if (s == 0x12345) out[0] = s;
}

template <int MODE>
static void run(const char* what, int n_cus) {
  uint32_t* d; unsigned long long* dc;
  (void)hipMalloc(&d, 4); (void)hipMalloc(&dc, 8 * 4096);
  const int grid = n_cus * 2, iters = 20000;
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(1024), 0, 0, d, 100, dc);
  (void)hipEventRecord(a);
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(1024), 0, 0, d, iters, dc);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  unsigned long long h[4096]; (void)hipMemcpy(h, dc, 8 * grid, hipMemcpyDeviceToHost);
  double clk = 0; for (int i = 0; i < grid; ++i) clk += double(h[i]); clk /= grid;
  const double lanes = double(grid) * 1024.0 * iters * 8;
  printf("%-28s %8.2f ms  %.3g lanes/s  %.2f lanes/clk/CU (kernel clocks %.3g -> %.2f GHz)\n", what, ms, lanes / (ms * 1e-3),
         lanes / n_cus / clk / 1.0, clk, clk / (ms * 1e-3) / 1e9);
}

int main() {
  hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
  printf("%s, %d CUs\n", p.name, p.multiProcessorCount);
  run<0>("ds_add_u32 conflict-free", p.multiProcessorCount);
  run<1>("ds_add_u32 random words", p.multiProcessorCount);
  return 0;
}

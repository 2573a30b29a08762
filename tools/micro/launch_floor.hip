// What a launch costs when the kernel does nothing but answer: host clock from hipLaunchKernelGGL to the kernel's store into
// host-coherent memory being seen, for a grid of 1 and of 129 workgroups of 1024 threads (the single find's shape), with and
// without 2 KB of kernel arguments; and the launch call's own time.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>
struct Big { unsigned pad[576]; };
__global__ __launch_bounds__(1024) void answer(unsigned* out, unsigned seq) {
  __shared__ unsigned lds[16384];
  lds[threadIdx.x] = seq;
  __syncthreads();
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) __hip_atomic_store(out, lds[5], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ __launch_bounds__(1024) void answer_big(unsigned* out, unsigned seq, Big b) {
  __shared__ unsigned lds[16384];
  lds[threadIdx.x] = seq + b.pad[threadIdx.x & 511] * 0u;
  __syncthreads();
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) __hip_atomic_store(out, lds[5], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
int main() {
  unsigned *h = nullptr, *hd = nullptr;
  (void)hipHostMalloc((void**)&h, 4096, hipHostMallocMapped | hipHostMallocCoherent);
  (void)hipHostGetDevicePointer((void**)&hd, h, 0);
  hipStream_t st; (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  Big big{};
  for (int grid : {1, 129}) for (int with_big : {0, 1}) {
    std::vector<double> tot, call;
    volatile unsigned* o = h;
    for (unsigned r = 1; r <= 3000; ++r) {
      auto t0 = std::chrono::steady_clock::now();
      if (with_big) hipLaunchKernelGGL(answer_big, dim3(grid), dim3(1024), 0, st, hd, r, big);
      else hipLaunchKernelGGL(answer, dim3(grid), dim3(1024), 0, st, hd, r);
      auto t1 = std::chrono::steady_clock::now();
      while (*o != r) {}
      auto t2 = std::chrono::steady_clock::now();
      if (r > 100) { tot.push_back(std::chrono::duration<double, std::micro>(t2 - t0).count()); call.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count()); }
    }
    std::sort(tot.begin(), tot.end()); std::sort(call.begin(), call.end());
    printf("grid %3d x 1024 threads, %s: launch to answer p50 %.2f us (p90 %.2f), the launch call itself %.2f us\n", grid,
           with_big ? "2.3 KB of arguments" : "two arguments      ", tot[tot.size() / 2], tot[tot.size() * 9 / 10], call[call.size() / 2]);
  }
  return 0;
}

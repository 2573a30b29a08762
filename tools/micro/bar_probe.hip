// Can the host store into device memory directly (large BAR)?  hipExtMallocWithFlags(hipDeviceMallocFinegrained), a CPU
// store, a kernel that reads it; and how long a host store takes to be seen by a polling kernel, against host-coherent memory.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <csignal>
#include <csetjmp>
static sigjmp_buf jb;
static void on_segv(int) { siglongjmp(jb, 1); }
__global__ void echo(const unsigned* in, unsigned* out, unsigned rounds) {     // out (host-coherent) follows in, `rounds` times
  unsigned seen = 0;
  for (unsigned r = 0; r < rounds; ++r) {
    unsigned v;
    do { v = __hip_atomic_load(in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); } while (v == seen);
    seen = v;
    __hip_atomic_store(out, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
static double ping(volatile unsigned* in_host_view, const unsigned* in_dev, unsigned* out_host, unsigned* out_dev, int rounds) {
  hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  *in_host_view = 0; *out_host = 0;
  hipLaunchKernelGGL(echo, dim3(1), dim3(1), 0, st, in_dev, out_dev, unsigned(rounds));
  volatile unsigned* o = out_host;
  auto t0 = std::chrono::steady_clock::now();
  for (int r = 1; r <= rounds; ++r) { *in_host_view = unsigned(r); while (*o != unsigned(r)) {} }
  auto t1 = std::chrono::steady_clock::now();
  hipStreamSynchronize(st); hipStreamDestroy(st);
  return std::chrono::duration<double, std::micro>(t1 - t0).count() / rounds;
}
int main() {
  unsigned *h = nullptr, *hd = nullptr, *d = nullptr;
  hipHostMalloc((void**)&h, 4096, hipHostMallocMapped | hipHostMallocCoherent);
  hipHostGetDevicePointer((void**)&hd, h, 0);
  std::memset(h, 0, 4096);
  printf("host-coherent doorbell, host-coherent answer: %.2f us a round trip\n", ping(h, hd, h + 64, hd + 64, 2000));
  hipError_t e = hipExtMallocWithFlags((void**)&d, 4096, hipDeviceMallocFinegrained);
  printf("hipExtMallocWithFlags(finegrained): %s\n", hipGetErrorString(e));
  if (e != hipSuccess) return 0;
  hipMemset(d, 0, 4096); hipDeviceSynchronize();
  signal(SIGSEGV, on_segv); signal(SIGBUS, on_segv);
  if (sigsetjmp(jb, 1)) { printf("a host store into device memory faults: no doorbell in device memory here\n"); return 0; }
  volatile unsigned* dv = d;
  *dv = 5;                                                 // the question
  printf("host store into device memory: ok (reads back %u)\n", *dv);
  printf("device-memory doorbell, host-coherent answer: %.2f us a round trip\n", ping(dv, d, h + 64, hd + 64, 2000));
  return 0;
}

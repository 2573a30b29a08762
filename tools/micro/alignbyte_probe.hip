// Which bits of S2 does v_alignbyte_b32 use on gfx950?  (A ds_add_u32 to a misaligned LDS address, tried in
// the first version of this probe, faults the wave.)
// hipcc --offload-arch=gfx950 -O2 -o /tmp/alignbyte_probe tools/micro/alignbyte_probe.hip && /tmp/alignbyte_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void probe(uint32_t* out) {
  __shared__ uint32_t lds[64];
  const uint32_t t = threadIdx.x;
  lds[t] = 0;
  __syncthreads();
  const uint32_t hi = 0u, lo = 0x01000000u;
  const uint32_t sel = (t & 7u) | ((t >> 3) << 16) | 0xABC0u;       // low three bits vary, junk above them
  uint32_t r;
  asm volatile("v_alignbyte_b32 %0, %1, %2, %3" : "=v"(r) : "v"(hi), "v"(lo), "v"(sel));
  out[t] = r;
  __syncthreads();
  out[64 + t] = lds[t];
}

int main() {
  uint32_t* d; uint32_t h[128];
  hipMalloc(&d, sizeof(h));
  probe<<<1, 64>>>(d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int t = 0; t < 8; ++t) printf("alignbyte S2=%d (+junk): %08x\n", t, h[t]);
  printf("junk-independent: %s\n", (h[1] == h[9] && h[2] == h[18] && h[3] == h[27]) ? "yes" : "no");
  return 0;
}

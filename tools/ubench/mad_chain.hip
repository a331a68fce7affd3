// Does a v_mad_u64_u32 that accumulates into the register pair the previous one wrote issue later than an independent one,
// and how many interleaved accumulators hide it?  (The field multipliers of ufe.hpp sum a product column into ONE 64-bit
// accumulator: chains of up to N dependent multiply-adds.)  Patterns: NACC accumulators used round-robin, NACC = 1, 2, 3, 4, 8;
// 2, 3, 4 and 8 waves per SIMD.  Cycles come from s_memtime against s_memrealtime (100 MHz), not from a nominal clock.
//   hipcc -O3 --offload-arch=gfx950 -o mad_chain mad_chain.hip && ./mad_chain > gpurun_out/r03_ubench_mad_chain.jsonl
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int NACC>
__global__ void __launch_bounds__(256) k_chain(uint32_t* out, uint64_t* clocks, int iters, uint32_t seed) {
  const uint32_t lane = threadIdx.x + blockIdx.x * blockDim.x;
  uint32_t a[8], b[8];
  uint64_t w[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    a[c] = lane * 2654435761u + c * 40503u + seed;
    b[c] = (lane ^ 0x9e3779b9u) + c * 7919u + seed;
    w[c] = ((uint64_t)a[c] << 32) | b[c];
  }
  const uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 48; ++u)  // 48 = lcm-friendly for 1, 2, 3, 4, 8 accumulators
      asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[u % NACC]) : "v"(a[u & 7]), "v"(b[(u * 3) & 7]) : "vcc");
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  uint32_t acc = 0;
#pragma unroll
  for (int c = 0; c < 8; ++c) acc ^= (uint32_t)w[c] ^ (uint32_t)(w[c] >> 32);
  if (acc == 0x12345678u) out[lane] = acc;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clocks[0] = t1 - t0; clocks[1] = r1 - r0; }
}

template <int NACC>
int run(uint32_t* dout, uint64_t* dclk, int waves_per_simd, int iters) {
  const int blocks = 256 * waves_per_simd;  // 256-thread blocks = 4 waves, one per SIMD of a CU
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k_chain<NACC>, dim3(blocks), dim3(256), 0, 0, dout, dclk, 16, 1u);
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  uint64_t clk[2] = {0, 0};
  for (int r = 0; r < 3; ++r) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_chain<NACC>, dim3(blocks), dim3(256), 0, 0, dout, dclk, iters, 1u);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) {
      best = ms;
      CK(hipMemcpy(clk, dclk, sizeof(clk), hipMemcpyDeviceToHost));
    }
  }
  const double ghz = clk[1] ? (double)clk[0] / (double)clk[1] * 0.1 : 0.0;
  const double wave_insts = (double)blocks * 4.0 * (double)iters * 48.0;
  const double cyc = (best * 1e-3) * ghz * 1e9 * 1024.0 / wave_insts;
  printf("{\"accumulators\": %d, \"waves_per_simd\": %d, \"ms\": %.3f, \"clock_ghz\": %.3f, \"cycles_per_mad_per_simd\": %.2f}\n", NACC, waves_per_simd, best, ghz, cyc);
  return 0;
}

int main() {
  uint32_t* dout;
  uint64_t* dclk;
  CK(hipMalloc(&dout, 256 * 64 * 256 * sizeof(uint32_t)));
  CK(hipMalloc(&dclk, 2 * sizeof(uint64_t)));
  for (int wps : {2, 3, 4, 8}) {
    if (run<1>(dout, dclk, wps, 600)) return 1;
    if (run<2>(dout, dclk, wps, 600)) return 1;
    if (run<3>(dout, dclk, wps, 600)) return 1;
    if (run<4>(dout, dclk, wps, 600)) return 1;
    if (run<8>(dout, dclk, wps, 600)) return 1;
  }
  return 0;
}

// Field-multiplication microbenchmark for gfx950: saturated 8x32 Montgomery (fe.hpp, what
// the kernels use) against a prototype unsaturated signed 10x28 Montgomery built on
// v_mad_i64_i32 without carry instructions.  Reports cycles per field multiplication per
// wavefront-SIMD slot.  Also times a few instructions the unsaturated form relies on.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I eccoxide_amd/csrc tools/ubench/fe_bench.hip -o tools/ubench/fe_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "curve.hpp"

using namespace eccx;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

// ---- prototype: P-256, radix 2^28, 10 signed limbs, R = 2^280 ----------------------------
constexpr int UB = 28, UN = 10;
constexpr uint32_t UMASK = (1u << UB) - 1;
// p = 2^256 - 2^224 + 2^192 + 2^96 - 1 in radix 2^28
__device__ constexpr int32_t UP[UN] = {0x0fffffff, 0x0fffffff, 0x0fffffff, 0x00000fff, 0, 0, 0x01000000, 0, 0x0fffffff, 0x0000000f};

__device__ __forceinline__ void smad(int64_t& acc, int32_t a, int32_t b) {
  asm("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "vcc");
}
__device__ __forceinline__ void smad_k(int64_t& acc, int32_t a, int32_t k) {
  asm("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "s"(k) : "vcc");
}

struct UFe { int32_t v[UN]; };

template <bool SQR>
__device__ __forceinline__ void umul(UFe& r, const UFe& a, const UFe& b) {
  int32_t m[UN];
  int32_t t[UN];
  int64_t acc = 0;
  int32_t a2[UN];
  if (SQR) {
#pragma unroll
    for (int i = 0; i < UN; ++i) a2[i] = a.v[i] * 2;
  }
#pragma unroll
  for (int k = 0; k < 2 * UN - 1; ++k) {
    if (SQR) {
#pragma unroll
      for (int i = (k < UN ? 0 : k - UN + 1); 2 * i < k; ++i) acc += (int64_t)a.v[i] * a2[k - i];
      if ((k & 1) == 0) acc += (int64_t)a.v[k / 2] * a.v[k / 2];
    } else {
#pragma unroll
      for (int i = (k < UN ? 0 : k - UN + 1); i <= (k < UN ? k : UN - 1); ++i) acc += (int64_t)a.v[i] * b.v[k - i];
    }
#pragma unroll
    for (int i = (k < UN ? 0 : k - UN + 1); i <= (k < UN ? k - 1 : UN - 1); ++i) {
      if (UP[k - i] != 0) acc += (int64_t)m[i] * UP[k - i];
    }
    if (k < UN) {
      m[k] = (int32_t)((uint32_t)acc & UMASK);   // -p^-1 mod 2^28 = 1
      acc += (int64_t)m[k] * UP[0];
    } else {
      t[k - UN] = (int32_t)((uint32_t)acc & UMASK);
    }
    acc >>= UB;
  }
  t[UN - 1] = (int32_t)acc;
#pragma unroll
  for (int i = 0; i < UN; ++i) r.v[i] = t[i];
}

template <int MODE>
__global__ void __launch_bounds__(256) k_fe(uint32_t* out, int iters, uint32_t seed) {
  uint32_t lane = threadIdx.x + blockIdx.x * blockDim.x;
  if (MODE == 0) {
    Fe<8> x, y;
#pragma unroll
    for (int i = 0; i < 8; ++i) { x.v[i] = lane * 2654435761u + i * seed; y.v[i] = (lane ^ 0x9e3779b9u) + i * 77u + seed; }
    x.v[7] &= 0x7fffffffu; y.v[7] &= 0x7fffffffu;
    for (int it = 0; it < iters; ++it) {
      fe_mul<P256>(x, x, y);
      fe_mul<P256>(y, y, x);
    }
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc ^= x.v[i] ^ y.v[i];
    out[lane] = acc;
  } else {
    UFe x, y;
#pragma unroll
    for (int i = 0; i < UN; ++i) { x.v[i] = (lane * 2654435761u + i * seed) & UMASK; y.v[i] = ((lane ^ 0x9e3779b9u) + i * 77u + seed) & UMASK; }
    x.v[UN - 1] &= 0xf; y.v[UN - 1] &= 0xf;
    for (int it = 0; it < iters; ++it) {
      if (MODE == 1) { umul<false>(x, x, y); umul<false>(y, y, x); }
      else { umul<true>(x, x, x); umul<true>(y, y, y); }
    }
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < UN; ++i) acc ^= (uint32_t)x.v[i] ^ (uint32_t)y.v[i];
    out[lane] = acc;
  }
}

template <int MODE>
int run(const char* name, uint32_t* dout, int blocks, int iters) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k_fe<MODE>, dim3(blocks), dim3(256), 0, 0, dout, 4, 1u);
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 3; ++r) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_fe<MODE>, dim3(blocks), dim3(256), 0, 0, dout, iters, 1u);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  double muls_per_wave = 2.0 * iters;
  double waves_per_simd = (double)blocks * 4 / 1024.0;
  double cyc = best * 1e-3 * 2.4e9 / (muls_per_wave * waves_per_simd);
  printf("{\"op\": \"%s\", \"blocks\": %d, \"ms\": %.3f, \"cycles_per_mul_per_simd_at_2p4GHz\": %.1f, \"Gmul_per_s\": %.2f}\n", name, blocks, best, cyc,
         (double)blocks * 256 * muls_per_wave / (best * 1e-3) * 1e-9);
  return 0;
}

int main() {
  uint32_t* dout;
  CK(hipMalloc(&dout, 1024 * 4 * 256 * sizeof(uint32_t)));
  for (int blocks : {256, 512, 768, 1024, 1024 * 2, 1024 * 4}) {  // 1, 2, 3, 4, 8, 16 waves per SIMD
    run<0>("saturated 8x32 fe_mul<P256>", dout, blocks, 2000);
    run<1>("unsaturated 10x28 mul (v_mad_i64_i32)", dout, blocks, 2000);
    run<2>("unsaturated 10x28 sqr (v_mad_i64_i32)", dout, blocks, 2000);
  }
  CK(hipFree(dout));
  return 0;
}

// Prototype: P-256 in unsaturated radix 2^29 (9 limbs, R = 2^261), Montgomery product and
// square with hand-placed v_mad_u64_u32 chunks and NO carry instructions, against the
// saturated fe_mul<P256> of the kernels.  Reports cycles per operation per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "curve.hpp"
using namespace eccx;
#include "umad_chunks.inc"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

constexpr int N = 9;
constexpr uint32_t MASK = (1u << 29) - 1;
// p256 in radix 2^29
__device__ constexpr uint32_t P29[N] = {0x1fffffffu, 0x1fffffffu, 0x1fffffffu, 0x000001ffu, 0u, 0u, 0x00040000u, 0x1fe00000u, 0x00ffffffu};

template <bool K>
struct UQ {
  uint32_t x[9], y[9];
  int n = 0;
  __device__ __forceinline__ void push(uint64_t& acc, uint32_t a, uint32_t b) { x[n] = a; y[n] = b; if (++n == 9) flush(acc); }
  __device__ __forceinline__ void flush(uint64_t& acc) {
#define Q(i) x[i], y[i]
    if constexpr (K) {
      switch (n) {
        case 1: umad1_k(acc, Q(0)); break; case 2: umad2_k(acc, Q(0), Q(1)); break; case 3: umad3_k(acc, Q(0), Q(1), Q(2)); break;
        case 4: umad4_k(acc, Q(0), Q(1), Q(2), Q(3)); break; case 5: umad5_k(acc, Q(0), Q(1), Q(2), Q(3), Q(4)); break;
        case 6: umad6_k(acc, Q(0), Q(1), Q(2), Q(3), Q(4), Q(5)); break; case 7: umad7_k(acc, Q(0), Q(1), Q(2), Q(3), Q(4), Q(5), Q(6)); break;
        case 8: umad8_k(acc, Q(0), Q(1), Q(2), Q(3), Q(4), Q(5), Q(6), Q(7)); break;
        case 9: umad9_k(acc, Q(0), Q(1), Q(2), Q(3), Q(4), Q(5), Q(6), Q(7), Q(8)); break; default: break;
      }
    } else {
      switch (n) {
        case 1: umad1_v(acc, Q(0)); break; case 2: umad2_v(acc, Q(0), Q(1)); break; case 3: umad3_v(acc, Q(0), Q(1), Q(2)); break;
        case 4: umad4_v(acc, Q(0), Q(1), Q(2), Q(3)); break; case 5: umad5_v(acc, Q(0), Q(1), Q(2), Q(3), Q(4)); break;
        case 6: umad6_v(acc, Q(0), Q(1), Q(2), Q(3), Q(4), Q(5)); break; case 7: umad7_v(acc, Q(0), Q(1), Q(2), Q(3), Q(4), Q(5), Q(6)); break;
        case 8: umad8_v(acc, Q(0), Q(1), Q(2), Q(3), Q(4), Q(5), Q(6), Q(7)); break;
        case 9: umad9_v(acc, Q(0), Q(1), Q(2), Q(3), Q(4), Q(5), Q(6), Q(7), Q(8)); break; default: break;
      }
    }
#undef Q
    n = 0;
  }
};

struct U9 { uint32_t v[N]; };

template <bool SQR>
__device__ __forceinline__ void umul(U9& r, const U9& a, const U9& b) {
  uint32_t m[N], t[N], a2[N];
  if (SQR) {
#pragma unroll
    for (int i = 0; i < N; ++i) a2[i] = a.v[i] << 1;
  }
  uint64_t acc = 0;
  UQ<false> qa;
  UQ<true> qm;
#pragma unroll
  for (int k = 0; k < 2 * N - 1; ++k) {
    const int lo = k < N ? 0 : k - N + 1, hi = k < N ? k : N - 1;
    if (SQR) {
#pragma unroll
      for (int i = lo; 2 * i < k; ++i) qa.push(acc, a.v[i], a2[k - i]);
      if ((k & 1) == 0) qa.push(acc, a.v[k / 2], a.v[k / 2]);
    } else {
#pragma unroll
      for (int i = lo; i <= hi; ++i) qa.push(acc, a.v[i], b.v[k - i]);
    }
    qa.flush(acc);
#pragma unroll
    for (int i = lo; i <= (k < N ? k - 1 : N - 1); ++i)
      if (P29[k - i] != 0) qm.push(acc, m[i], P29[k - i]);
    qm.flush(acc);
    if (k < N) {
      m[k] = (uint32_t)acc & MASK;
      umad1_k(acc, m[k], P29[0]);
    } else {
      t[k - N] = (uint32_t)acc & MASK;
    }
    acc >>= 29;
  }
  t[N - 1] = (uint32_t)acc;
#pragma unroll
  for (int i = 0; i < N; ++i) r.v[i] = t[i];
}

template <int MODE>
__global__ void __launch_bounds__(256) k_fe(uint32_t* out, int iters, uint32_t seed) {
  uint32_t lane = threadIdx.x + blockIdx.x * blockDim.x;
  if (MODE == 0) {
    Fe<8> x, y;
#pragma unroll
    for (int i = 0; i < 8; ++i) { x.v[i] = lane * 2654435761u + i * seed; y.v[i] = (lane ^ 0x9e3779b9u) + i * 77u + seed; }
    x.v[7] &= 0x7fffffffu; y.v[7] &= 0x7fffffffu;
    for (int it = 0; it < iters; ++it) { fe_mul<P256>(x, x, y); fe_mul<P256>(y, y, x); }
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc ^= x.v[i] ^ y.v[i];
    out[lane] = acc;
  } else {
    U9 x, y;
#pragma unroll
    for (int i = 0; i < N; ++i) { x.v[i] = (lane * 2654435761u + i * seed) & MASK; y.v[i] = ((lane ^ 0x9e3779b9u) + i * 77u + seed) & MASK; }
    x.v[N - 1] &= 0xffffff; y.v[N - 1] &= 0xffffff;
    for (int it = 0; it < iters; ++it) {
      if (MODE == 1) { umul<false>(x, x, y); umul<false>(y, y, x); }
      else { umul<true>(x, x, x); umul<true>(y, y, y); }
    }
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) acc ^= x.v[i] ^ y.v[i];
    out[lane] = acc;
  }
}

template <int MODE>
int run(const char* name, uint32_t* dout, int blocks, int iters) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k_fe<MODE>, dim3(blocks), dim3(256), 0, 0, dout, 4, 1u);
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 3; ++r) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_fe<MODE>, dim3(blocks), dim3(256), 0, 0, dout, iters, 1u);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  double cyc = best * 1e-3 * 2.4e9 / (2.0 * iters * ((double)blocks * 4 / 1024.0));
  printf("{\"op\": \"%s\", \"waves_per_simd\": %.0f, \"ms\": %.3f, \"cycles_per_op_per_simd_at_2p4GHz\": %.1f}\n", name, blocks * 4 / 1024.0, best, cyc);
  return 0;
}

int main() {
  uint32_t* dout;
  CK(hipMalloc(&dout, 1024 * 4 * 256 * sizeof(uint32_t)));
  for (int blocks : {512, 1024, 1024 * 2, 1024 * 4}) {
    run<0>("saturated 8x32 fe_mul<P256>", dout, blocks, 2000);
    run<1>("unsaturated 9x29 mul, mad-only chunks", dout, blocks, 2000);
    run<2>("unsaturated 9x29 sqr, mad-only chunks", dout, blocks, 2000);
  }
  CK(hipFree(dout));
  return 0;
}

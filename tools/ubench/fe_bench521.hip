// A/B for the P-521 field product (18 limbs of 29 bits, p = 2^521 - 1, 2^522 = 2 mod p):
//   mode 0  the kernels' product: wrapped schoolbook, 324 v_mad_u64_u32 into 18 columns, the carry of
//           a column is the starting value of the next column's chain (ufe.hpp u_mul_core_mers)
//   mode 1  one level of Karatsuba on the Mersenne structure: with y = 2^261 (y^2 = 2),
//           (a0 + a1 y)(b0 + b1 y) = (a0 b0 + 2 a1 b1) + y (a0 b1 + a1 b0) and
//           a0 b1 + a1 b0 = (a0 - a1)(b1 - b0) + a0 b0 + a1 b1: three 9 x 9 products (243 multiplies,
//           one of them signed: v_mad_i64_i32) whose 17 + 17 + 17 column sums must then be COMBINED
//           with 64-bit additions (v_lshl_add_u64) -- the part that decides the outcome on gfx950,
//           where a 64-bit add costs as much as a multiply
//   mode 2 / 3  the same for the square (171 multiplies against 3 x 45)
// Both modes are checked against each other on the same inputs (bit-identical digits) before timing.
//   hipcc -O3 --offload-arch=gfx950 -I ../../eccoxide_amd/csrc -mllvm -pragma-unroll-threshold=1000000 fe_bench521.hip -o fe_bench521
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "ufe.hpp"
#include "curve.hpp"
using namespace eccx;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

constexpr int N = 18, H = 9, B = 29;
constexpr uint32_t MASK = (1u << B) - 1;

__device__ __forceinline__ void add64(uint64_t& d, uint64_t a) {  // d += a, one instruction
  asm("v_lshl_add_u64 %0, %1, 0, %0" : "+v"(d) : "v"(a));
}
__device__ __forceinline__ void add64x2(uint64_t& d, uint64_t a) {  // d += 2a, one instruction
  asm("v_lshl_add_u64 %0, %1, 1, %0" : "+v"(d) : "v"(a));
}
__device__ __forceinline__ void smad(int64_t& acc, int32_t a, int32_t b) {
  asm("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "vcc");
}

template <bool SQR>
__device__ __forceinline__ void kmul(uint32_t (&r)[N], const uint32_t (&a)[N], const uint32_t (&b)[N]) {
  uint64_t p0[2 * H - 1], p1[2 * H - 1];
  int64_t pd[2 * H - 1];
  uint32_t a2lo[H], a2hi[H];
  if (SQR) {
#pragma unroll
    for (int i = 0; i < H; ++i) { a2lo[i] = a[i] << 1; a2hi[i] = a[H + i] << 1; }
  }
  int32_t da[H], db[H];
#pragma unroll
  for (int i = 0; i < H; ++i) {
    da[i] = (int32_t)a[i] - (int32_t)a[H + i];
    db[i] = (int32_t)b[H + i] - (int32_t)b[i];
  }
#pragma unroll
  for (int k = 0; k < 2 * H - 1; ++k) {
    const int lo = k < H ? 0 : k - H + 1, hi = k < H ? k : H - 1;
    uint64_t s0 = 0, s1 = 0;
    int64_t sd = 0;
    if (SQR) {
      // a0^2, a1^2 with halved cross terms; (a0 - a1)(a1 - a0) = -(a0 - a1)^2
#pragma unroll
      for (int i = lo; 2 * i < k; ++i) { umad1_v(s0, a[i], a2lo[k - i]); umad1_v(s1, a[H + i], a2hi[k - i]); }
      if ((k & 1) == 0) { umad1_v(s0, a[k / 2], a[k / 2]); umad1_v(s1, a[H + k / 2], a[H + k / 2]); }
#pragma unroll
      for (int i = lo; i <= hi; ++i) smad(sd, da[i], db[k - i]);
    } else {
#pragma unroll
      for (int i = lo; i <= hi; ++i) {
        umad1_v(s0, a[i], b[k - i]);
        umad1_v(s1, a[H + i], b[H + k - i]);
        smad(sd, da[i], db[k - i]);
      }
    }
    p0[k] = s0; p1[k] = s1; pd[k] = sd;
  }
  // mid_j = pd_j + p0_j + p1_j  (= column j of a0 b1 + a1 b0)
  uint64_t mid[2 * H - 1];
#pragma unroll
  for (int j = 0; j < 2 * H - 1; ++j) {
    uint64_t m = (uint64_t)pd[j];
    add64(m, p0[j]);
    add64(m, p1[j]);
    mid[j] = m;
  }
  // column k of the result: p0_k + 2 p1_k + [mid_{k-9} | 2 mid_{k+9}], then one carry chain
  uint32_t t[N];
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    if (k < 2 * H - 1) { add64(acc, p0[k]); add64x2(acc, p1[k]); }
    if (k >= H) add64(acc, mid[k - H]);
    if (k + H < 2 * H - 1) add64x2(acc, mid[k + H]);
    t[k] = (uint32_t)acc & MASK;
    acc >>= B;
  }
  acc = (acc << 1) + t[0];  // weight 2^522 = 2
  t[0] = (uint32_t)acc & MASK;
  t[1] += (uint32_t)(acc >> B);
#pragma unroll
  for (int i = 0; i < N; ++i) r[i] = t[i];
}

template <int MODE>
__global__ void __launch_bounds__(256, 2) k_fe(uint32_t* out, int iters, uint32_t seed, int dump) {
  const uint32_t lane = threadIdx.x + blockIdx.x * blockDim.x;
  uint32_t x[N], y[N];
#pragma unroll
  for (int i = 0; i < N; ++i) { x[i] = (lane * 2654435761u + i * seed) & MASK; y[i] = ((lane ^ 0x9e3779b9u) * 40503u + i * 77u + seed) & MASK; }
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) { u_mul_core<P521U, false, false>(x, x, y); u_mul_core<P521U, false, false>(y, y, x); }
    if (MODE == 1) { kmul<false>(x, x, y); kmul<false>(y, y, x); }
    if (MODE == 2) { u_mul_core<P521U, true, false>(x, x, x); u_mul_core<P521U, true, false>(y, y, y); }
    if (MODE == 3) { kmul<true>(x, x, x); kmul<true>(y, y, y); }
#pragma unroll
    for (int i = 0; i < N; ++i) { x[i] &= MASK; y[i] &= MASK; }  // digit 1 may carry one unit over: keep the inputs tight
  }
  if (dump) {
#pragma unroll
    for (int i = 0; i < N; ++i) { out[(size_t)lane * 2 * N + i] = x[i]; out[(size_t)lane * 2 * N + N + i] = y[i]; }
  } else {
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) acc ^= x[i] ^ y[i];
    out[lane] = acc;
  }
}

template <int MODE>
int run(const char* name, uint32_t* dout, int blocks, int iters) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k_fe<MODE>, dim3(blocks), dim3(256), 0, 0, dout, 4, 1u, 0);
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 3; ++r) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_fe<MODE>, dim3(blocks), dim3(256), 0, 0, dout, iters, 1u, 0);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  hipFuncAttributes at;
  CK(hipFuncGetAttributes(&at, reinterpret_cast<const void*>(&k_fe<MODE>)));
  const double cyc = best * 1e-3 * 2.4e9 / (2.0 * iters * ((double)blocks * 4 / 1024.0));
  printf("{\"op\": \"%s\", \"waves_per_simd\": %.0f, \"ms\": %.3f, \"cycles_per_op_per_simd_at_2p4GHz\": %.1f, \"vgprs\": %d, \"scratch_bytes\": %d}\n",
         name, blocks * 4 / 1024.0, best, cyc, at.numRegs, (int)at.localSizeBytes);
  return 0;
}

template <int A, int Bm>
int same(uint32_t* d0, uint32_t* d1, const char* what) {
  const int blocks = 8, lanes = blocks * 256;
  hipLaunchKernelGGL(k_fe<A>, dim3(blocks), dim3(256), 0, 0, d0, 3, 12345u, 1);
  hipLaunchKernelGGL(k_fe<Bm>, dim3(blocks), dim3(256), 0, 0, d1, 3, 12345u, 1);
  CK(hipDeviceSynchronize());
  static uint32_t h0[8 * 256 * 2 * N], h1[8 * 256 * 2 * N];
  CK(hipMemcpy(h0, d0, sizeof(h0), hipMemcpyDeviceToHost));
  CK(hipMemcpy(h1, d1, sizeof(h1), hipMemcpyDeviceToHost));
  int bad = 0;
  for (int i = 0; i < lanes * 2 * N; ++i) bad += h0[i] != h1[i];
  printf("{\"check\": \"%s\", \"digits_compared\": %d, \"mismatches\": %d}\n", what, lanes * 2 * N, bad);
  return bad ? 1 : 0;
}

int main() {
  uint32_t *d0, *d1;
  CK(hipMalloc(&d0, 1024 * 2 * 256 * 2 * N * sizeof(uint32_t)));
  CK(hipMalloc(&d1, 1024 * 2 * 256 * 2 * N * sizeof(uint32_t)));
  if (same<0, 1>(d0, d1, "karatsuba product == wrapped schoolbook product")) return 1;
  if (same<2, 3>(d0, d1, "karatsuba square == wrapped schoolbook square")) return 1;
  for (int blocks : {512, 1024}) {  // 2 and 4 waves per SIMD asked for; the register file decides what runs
    run<0>("p521 18x29 product, wrapped schoolbook (324 mads)", d0, blocks, 1000);
    run<1>("p521 18x29 product, one-level Karatsuba (243 mads + 64-bit recombination)", d0, blocks, 1000);
    run<2>("p521 18x29 square, wrapped schoolbook (171 mads)", d0, blocks, 1000);
    run<3>("p521 18x29 square, one-level Karatsuba (135 + 81 mads + recombination)", d0, blocks, 1000);
  }
  CK(hipFree(d0)); CK(hipFree(d1));
  return 0;
}

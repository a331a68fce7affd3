// A/B microbenchmark (round 3): are FP64 FMAs a cheaper multiplier for the field layer than
// v_mad_u64_u32?  v_fma_f64 issues every 4.4 cycles per wave and SIMD (profiles/r01_valu_rates.jsonl) and delivers
// 53 x 53 bits where the integer multiplier gives 32 x 32 in 5.
//
// Limb product in doubles (the construction of Emmart, Zheng and Weems for modular exponentiation on GPUs): for
// integers 0 <= a, b < 2^52 held exactly in doubles and round-toward-zero arithmetic,
//     hi = fma(a, b, 2^104)             the top 52 bits of a*b ride in the mantissa of a double with exponent 2^104
//     lo = fma(a, b, (2^104 + 2^52) - hi)   the low 52 bits, in the mantissa of a double with exponent 2^52
// and the mantissas are summed as INTEGERS (the bit patterns, whose exponent fields are equal and known, so that the
// constant is taken off once per column).  Per 52 x 52 product: 2 FMA + 1 FP64 subtraction + 2 64-bit integer additions.
// A 256-bit product is 25 of them (5 limbs of 52 bits); a Montgomery reduction at this radix is 5 low products for the
// factors and 25 more for m * p -- P-256's p + 1 has no digit structure in radix 2^52 (96 = 52 + 44).
//
// The kernel checks the construction bit for bit against 128-bit integer arithmetic on every lane, then times the
// schoolbook 5 x 5 product with its ten column accumulators.  Kill criterion: a P-256 product + reduction below 600
// cycles per wave and SIMD (the 9 x 29-bit integer form: 676, profiles/r02_ubench_lanes.jsonl).
//   build: make -C tools/ubench fp64_limb_bench      run on the GPU box: tools/ubench/fp64_limb_bench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

constexpr int L = 5;
__device__ __forceinline__ double d_of(uint64_t bits) { return __longlong_as_double((long long)bits); }
__device__ __forceinline__ uint64_t b_of(double d) { return (uint64_t)__double_as_longlong(d); }

// FP64 rounding mode of this wave: round toward zero (MODE register bits 3:2 = 3)
__device__ __forceinline__ void set_fp64_rz() { asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 3"); }

__device__ __forceinline__ void dmul52(uint64_t& hi_acc, uint64_t& lo_acc, double a, double b) {
  const double C1 = 0x1p104, C2 = 0x1p104 + 0x1p52;
  double hi, t, lo;
  asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(hi) : "v"(a), "v"(b), "v"(C1));
  asm volatile("v_add_f64 %0, %1, -%2" : "=v"(t) : "v"(C2), "v"(hi));
  asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(lo) : "v"(a), "v"(b), "v"(t));
  hi_acc += b_of(hi);  // mantissa = floor(a*b / 2^52) under exponent 2^104
  lo_acc += b_of(lo);  // mantissa = a*b mod 2^52 under exponent 2^52
}

__global__ void __launch_bounds__(256) k_check(const uint64_t* a, const uint64_t* b, uint64_t* out) {
  set_fp64_rz();
  const int i = threadIdx.x + blockIdx.x * blockDim.x;
  uint64_t hi = 0, lo = 0;
  dmul52(hi, lo, (double)a[i], (double)b[i]);
  out[2 * i] = hi - b_of(0x1p104);
  out[2 * i + 1] = lo - b_of(0x1p52);
}

__global__ void __launch_bounds__(256) k_prod(uint64_t* out, uint64_t* clocks, int iters, uint64_t seed) {
  set_fp64_rz();
  const uint32_t lane = threadIdx.x + blockIdx.x * blockDim.x;
  double a[L], b[L];
#pragma unroll
  for (int i = 0; i < L; ++i) {
    a[i] = (double)(((uint64_t)lane * 0x9e3779b97f4a7c15ull + seed + i * 77) & ((1ull << 52) - 1));
    b[i] = (double)(((uint64_t)lane * 0xc2b2ae3d27d4eb4full + seed * 3 + i * 131) & ((1ull << 52) - 1));
  }
  uint64_t sink = 0;
  const uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    uint64_t col[2 * L];
#pragma unroll
    for (int k = 0; k < 2 * L; ++k) col[k] = 0;
#pragma unroll
    for (int i = 0; i < L; ++i)
#pragma unroll
      for (int j = 0; j < L; ++j) dmul52(col[i + j + 1], col[i + j], a[i], b[j]);
#pragma unroll
    for (int k = 0; k < 2 * L; ++k) sink ^= col[k];
    a[0] = d_of((b_of(a[0]) ^ (sink & 1)));  // keep the loop from being hoisted
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (sink == 0x123456789abcdefull) out[lane] = sink;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clocks[0] = t1 - t0; clocks[1] = r1 - r0; }
}

int main() {
  const int N = 1 << 16;
  uint64_t *da, *db, *dout, *dclk;
  CK(hipMalloc(&da, N * 8)); CK(hipMalloc(&db, N * 8)); CK(hipMalloc(&dout, 2 * N * 8 + 256 * 1024 * 8)); CK(hipMalloc(&dclk, 16));
  uint64_t* ha = new uint64_t[N]; uint64_t* hb = new uint64_t[N]; uint64_t* ho = new uint64_t[2 * N];
  uint64_t x = 88172645463325252ull;
  for (int i = 0; i < N; ++i) {
    x ^= x << 13; x ^= x >> 7; x ^= x << 17; ha[i] = x & ((1ull << 52) - 1);
    x ^= x << 13; x ^= x >> 7; x ^= x << 17; hb[i] = x & ((1ull << 52) - 1);
  }
  ha[0] = hb[0] = (1ull << 52) - 1; ha[1] = 0; hb[2] = 1; ha[2] = (1ull << 52) - 1;
  CK(hipMemcpy(da, ha, N * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(db, hb, N * 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_check, dim3(N / 256), dim3(256), 0, 0, da, db, dout);
  CK(hipMemcpy(ho, dout, 2 * N * 8, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int i = 0; i < N; ++i) {
    unsigned __int128 p = (unsigned __int128)ha[i] * hb[i];
    if (ho[2 * i] != (uint64_t)(p >> 52) || ho[2 * i + 1] != (uint64_t)(p & ((1ull << 52) - 1))) ++bad;
  }
  printf("{\"check\": \"52x52 product as two round-toward-zero FMAs vs 128-bit integers\", \"lanes\": %d, \"mismatches\": %d}\n", N, bad);
  for (int blocks : {256 * 4, 256 * 8}) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_prod, dim3(blocks), dim3(256), 0, 0, dout, dclk, 16, 1ull);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    uint64_t clk[2] = {0, 0};
    for (int r = 0; r < 3; ++r) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_prod, dim3(blocks), dim3(256), 0, 0, dout, dclk, iters, 7ull);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) { best = ms; CK(hipMemcpy(clk, dclk, 16, hipMemcpyDeviceToHost)); }
    }
    const double ghz = clk[1] ? (double)clk[0] / (double)clk[1] * 0.1 : 0.0;
    const double waves = (double)blocks * 4.0;
    const double cyc = (best * 1e-3) * ghz * 1e9 * 1024.0 / (waves * iters);
    printf("{\"kernel\": \"5x5 limb product, 52-bit limbs in doubles (25 x [2 v_fma_f64 + v_add_f64 + 2 64-bit integer adds])\", "
           "\"waves_per_simd\": %d, \"ms\": %.3f, \"clock_ghz\": %.3f, \"cycles_per_product_per_simd\": %.1f, "
           "\"cycles_per_limb_product\": %.2f, \"projected_product_plus_montgomery_reduction\": %.0f, "
           "\"integer_9x29_product_plus_reduction\": 676, \"kill_criterion\": \"< 600\"}\n",
           blocks / 256, best, ghz, cyc, cyc / 25.0, cyc / 25.0 * 55.0);
  }
  return bad != 0;
}

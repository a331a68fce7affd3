// A/B for the layout BASELINE.json's north star suggests -- field limbs spread ACROSS lanes with
// cross-lane carry handling -- against the layout the kernels use (one field element per lane).
// Workload: P-256 Montgomery products, both variants bit-checked against each other.
//   mode 0  one element per lane: 8 x 32-bit saturated limbs in registers, fe_mul<P256> (fe.hpp)
//   mode 1  one element per lane: 9 x 29-bit unsaturated limbs (ufe.hpp), the default kernels' product
//   mode 2  one LIMB per lane, 8 lanes per element (8 elements per wavefront), limb-parallel CIOS:
//           per step i the group broadcasts b_i (ds_swizzle), every lane multiplies its a_j by it into a
//           64-bit carry-save accumulator, lane 0's low word is the Montgomery factor (N0 = 1 for P-256),
//           broadcast again, every lane adds q p_j, and the accumulators move one lane down
//           (v_mov_b32_dpp row_shr); carries are resolved once at the end with a ballot-free
//           propagate/generate scan over the 8 lanes (7 dependent DPP steps at most)
// Reported: cycles per PRODUCT per SIMD (a wavefront instruction of mode 2 serves 8 products, of modes
// 0 / 1 it serves 64).
//   hipcc -O3 --offload-arch=gfx950 -I ../../eccoxide_amd/csrc -mllvm -pragma-unroll-threshold=1000000 lanes_bench.hip -o lanes_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "ufe.hpp"
#include "curve.hpp"
using namespace eccx;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

// lane' = (lane & 0x18) | i inside each group of 32 lanes: broadcast limb i of every 8-lane group
template <int I>
__device__ __forceinline__ uint32_t bcast8(uint32_t v) {
  // ds_swizzle bitmask mode: offset = xor_mask << 10 | or_mask << 5 | and_mask
  return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, (0 << 10) | (I << 5) | 0x18);
}
// value of lane + 1 (next limb up) inside a row of 16 lanes; lanes 7 / 15 read garbage that is masked out
__device__ __forceinline__ uint32_t from_up(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x101 /* row_shl:1 */, 0xf, 0xf, false);
}
// value of lane - 1 (next limb down)
__device__ __forceinline__ uint32_t from_down(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111 /* row_shr:1 */, 0xf, 0xf, false);
}

// a, b: this lane's limb (lane j of a group holds limb j); returns limb j of a b R^-1 mod p (R = 2^256),
// fully reduced.  p limbs in pj.
__device__ __forceinline__ uint32_t lanes_mont_mul(uint32_t a, uint32_t b, uint32_t pj, int j) {
  uint64_t acc = 0;  // carry-save: column value of this lane, up to 64 bits + a spill word
  uint32_t spill = 0;  // bit 64 and above of the column (at most a few units)
#define STEP(I)                                                                                     \
  {                                                                                                 \
    const uint32_t bi = bcast8<I>(b);                                                               \
    uint64_t t = acc + (uint64_t)a * bi;                                                            \
    spill += t < acc;                                                                               \
    acc = t;                                                                                        \
    const uint32_t q = bcast8<0>((uint32_t)acc); /* N0 = 1: the factor is lane 0's low word */      \
    t = acc + (uint64_t)q * pj;                                                                     \
    spill += t < acc;                                                                               \
    acc = t;                                                                                        \
    /* divide by 2^32: lane j keeps its high word (+ spill) and takes the low word of lane j + 1 */ \
    const uint32_t up = from_up((uint32_t)acc); /* every lane executes the move: a DPP read of a */ \
    const uint32_t lo_up = (j == 7) ? 0u : up;  /* lane masked off by EXEC returns nothing       */ \
    const uint64_t hi = (acc >> 32) | ((uint64_t)spill << 32);                                      \
    acc = hi + lo_up;                                                                               \
    spill = 0;                                                                                      \
    /* lane 0's low word was zero by construction; its high part stays as this lane's carry-in */    \
  }
  STEP(0) STEP(1) STEP(2) STEP(3) STEP(4) STEP(5) STEP(6) STEP(7)
#undef STEP
  // resolve the carry-save form: every lane sends what it holds above 32 bits one lane up, 8 + 1 rounds
  // (a carry born in lane 0 needs 7 rounds to reach lane 7); what leaves lane 7 is the result's bit 256
  uint32_t limb = (uint32_t)acc;
  uint32_t carry = (uint32_t)(acc >> 32);
  uint32_t top = 0;
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    if (j == 7) top += carry;
    uint32_t cin = from_down(carry);
    if (j == 0) cin = 0;
    const uint32_t sum = limb + cin;
    carry = sum < limb;
    limb = sum;
  }
  if (j == 7) top += carry;
  // conditional subtraction of p: d = limbs - p with a borrow ripple of the same shape
  uint32_t d = limb - pj, bw = limb < pj, bout = 0;
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    if (j == 7) bout += bw;
    uint32_t bin = from_down(bw);
    if (j == 0) bin = 0;
    const uint32_t nd = d - bin;
    bw = d < bin;
    d = nd;
  }
  if (j == 7) bout += bw;
  const uint32_t topv = bcast8<7>(top), boutv = bcast8<7>(bout);
  const bool ge = (topv != 0) || (boutv == 0);  // (top : limbs) >= p
  return ge ? d : limb;
}

template <int MODE>
__global__ void __launch_bounds__(256) k_fe(uint32_t* out, int iters, uint32_t seed, int dump) {
  const uint32_t lane = threadIdx.x + blockIdx.x * blockDim.x;
  if (MODE == 0) {
    Fe<8> x, y;
#pragma unroll
    for (int i = 0; i < 8; ++i) { x.v[i] = lane * 2654435761u + i * seed; y.v[i] = (lane ^ 0x9e3779b9u) * 40503u + i * 77u + seed; }
    x.v[7] &= 0x7fffffffu; y.v[7] &= 0x7fffffffu;
    for (int it = 0; it < iters; ++it) { fe_mul<P256>(x, x, y); fe_mul<P256>(y, y, x); }
    if (dump) {
#pragma unroll
      for (int i = 0; i < 8; ++i) { out[(size_t)lane * 16 + i] = x.v[i]; out[(size_t)lane * 16 + 8 + i] = y.v[i]; }
    } else {
      uint32_t acc = 0;
#pragma unroll
      for (int i = 0; i < 8; ++i) acc ^= x.v[i] ^ y.v[i];
      out[lane] = acc;
    }
  } else if (MODE == 1) {
    uint32_t x[9], y[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) { x[i] = (lane * 2654435761u + i * seed) & P256U::MASK; y[i] = ((lane ^ 0x9e3779b9u) * 40503u + i * 77u + seed) & P256U::MASK; }
    x[8] &= 0xffffff; y[8] &= 0xffffff;
    for (int it = 0; it < iters; ++it) { u_mul_core<P256U, false, false>(x, x, y); u_mul_core<P256U, false, false>(y, y, x); }
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) acc ^= x[i] ^ y[i];
    out[lane] = acc;
  } else {
    // element e = lane / 8 of the grid, limb j = lane % 8: the same inputs as element e of mode 0
    const uint32_t e = lane >> 3;
    const int j = (int)(lane & 7);
    uint32_t x = e * 2654435761u + j * seed, y = (e ^ 0x9e3779b9u) * 40503u + j * 77u + seed;
    if (j == 7) { x &= 0x7fffffffu; y &= 0x7fffffffu; }
    const uint32_t pj = P256::P[j];
    for (int it = 0; it < iters; ++it) { x = lanes_mont_mul(x, y, pj, j); y = lanes_mont_mul(y, x, pj, j); }
    if (dump) {
      out[(size_t)e * 16 + j] = x;
      out[(size_t)e * 16 + 8 + j] = y;
    } else {
      out[lane] = x ^ y;
    }
  }
}

template <int MODE>
int run(const char* name, uint32_t* dout, int blocks, int iters, int products_per_wave_instruction) {
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
  const double waves_per_simd = blocks * 4 / 1024.0;
  const double cyc_per_wave_op = best * 1e-3 * 2.4e9 / (2.0 * iters * waves_per_simd);
  printf("{\"layout\": \"%s\", \"waves_per_simd\": %.0f, \"ms\": %.3f, \"cycles_per_wave_operation_per_simd\": %.1f, "
         "\"products_per_wave_operation\": %d, \"cycles_per_product_per_simd\": %.2f}\n",
         name, waves_per_simd, best, cyc_per_wave_op, products_per_wave_instruction, cyc_per_wave_op / products_per_wave_instruction);
  return 0;
}

int main() {
  uint32_t *d0, *d1;
  const int blocks = 8;
  CK(hipMalloc(&d0, 4096 * 256 * 16 * sizeof(uint32_t)));
  CK(hipMalloc(&d1, 4096 * 256 * 16 * sizeof(uint32_t)));
  // the limb-per-lane product computes the same residues as fe_mul<P256> (elements 0 .. blocks*256/8 - 1)
  hipLaunchKernelGGL(k_fe<0>, dim3(blocks), dim3(256), 0, 0, d0, 3, 12345u, 1);
  hipLaunchKernelGGL(k_fe<2>, dim3(blocks), dim3(256), 0, 0, d1, 3, 12345u, 1);
  CK(hipDeviceSynchronize());
  const int elems = blocks * 256 / 8;
  static uint32_t h0[8 * 256 * 16], h1[8 * 256 * 16];
  CK(hipMemcpy(h0, d0, sizeof(h0), hipMemcpyDeviceToHost));
  CK(hipMemcpy(h1, d1, sizeof(h1), hipMemcpyDeviceToHost));
  int bad = 0;
  for (int i = 0; i < elems * 16; ++i) bad += h0[i] != h1[i];
  printf("{\"check\": \"limb-per-lane product == element-per-lane fe_mul<P256>\", \"words_compared\": %d, \"mismatches\": %d}\n", elems * 16, bad);
  for (int blk : {1024, 2048, 4096}) {
    run<0>("element per lane, 8 x 32 saturated (fe.hpp)", d0, blk, 2000, 64);
    run<1>("element per lane, 9 x 29 unsaturated (ufe.hpp, default kernels)", d0, blk, 2000, 64);
    run<2>("limb per lane, 8 lanes per element, swizzle broadcast + DPP carries", d0, blk, 2000, 8);
  }
  CK(hipFree(d0)); CK(hipFree(d1));
  return bad ? 1 : 0;
}

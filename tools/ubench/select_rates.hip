// What does a per-lane select of table words cost on gfx950?  (round 3: the secret-scalar kernels read every
// table entry and keep one by v_cndmask; the first build moved the lane mask to VCC with s_mov_b64 and ran
// four times slower than its VALU count.)  Each pattern selects 4 words per "op"; eight independent
// accumulator groups per lane, so dependencies do not bind.  Cycles are real: the shader clock is measured in
// the kernel (s_memtime against the 100 MHz s_memrealtime), not assumed.
//   build: make -C tools/ubench select_rates      run on the GPU box: tools/ubench/select_rates
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

constexpr int GROUPS = 8;
enum Pat { SMOV_CND4, VCMP_CND4, VCMP_CND8, CND4_E64, BFI4, VCMP_CND1, BPERM_ID, BPERM_RANDOM, BPERM_SAME, BPERM_FOLD,
           EXEC_MOV32_DIGIT, EXEC_MOV32_FULL, EXEC_MOV32_LOWER, EXEC_MOV32_ZERO, EXEC_MOV64_DIGIT, EXEC_MOV64_FULL, EXEC_MOV64_LOWER, EXEC_MOV64_ZERO,
           PK_FMA_F32, FMA_F64, AND_OR4 };

template <int P>
__global__ void __launch_bounds__(256) k_sel(uint32_t* out, uint64_t* clocks, int iters, uint32_t seed) {
  const uint32_t lane = threadIdx.x + blockIdx.x * blockDim.x;
  uint32_t o[GROUPS][8], a[GROUPS][8];
#pragma unroll
  for (int g = 0; g < GROUPS; ++g)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      o[g][i] = lane * 2654435761u + g * 97u + i;
      a[g][i] = (lane ^ seed) + g * 31u + i * 7u;
    }
  const uint32_t d = (lane * 7u + seed) & 31u;  // the lane's "digit"
  uint64_t o64[GROUPS][2], a64[GROUPS][2];  // the floating-point forms: words tagged once, outside the timed loop; accumulators start at +0
#pragma unroll
  for (int g = 0; g < GROUPS; ++g)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      o64[g][h] = 0;
      a64[g][h] = ((uint64_t)((a[g][2 * h + 1] & 0x3fffffffu) | 0x40000000u) << 32) | ((a[g][2 * h] & 0x3fffffffu) | 0x40000000u);
    }
  const uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    const uint32_t j = (uint32_t)it & 31u;  // wave-uniform "entry index"
#pragma unroll
    for (int g = 0; g < GROUPS; ++g) {
      if constexpr (P == SMOV_CND4) {
        const uint64_t m = __builtin_amdgcn_uicmp(d, j, 32);
        asm volatile("s_mov_b64 vcc, %8\n\tv_cndmask_b32_e32 %0, %0, %4, vcc\n\tv_cndmask_b32_e32 %1, %1, %5, vcc\n\t"
                     "v_cndmask_b32_e32 %2, %2, %6, vcc\n\tv_cndmask_b32_e32 %3, %3, %7, vcc"
                     : "+v"(o[g][0]), "+v"(o[g][1]), "+v"(o[g][2]), "+v"(o[g][3])
                     : "v"(a[g][0]), "v"(a[g][1]), "v"(a[g][2]), "v"(a[g][3]), "s"(m) : "vcc");
      } else if constexpr (P == VCMP_CND4) {
        asm volatile("v_cmp_eq_u32_e32 vcc, %9, %8\n\tv_cndmask_b32_e32 %0, %0, %4, vcc\n\tv_cndmask_b32_e32 %1, %1, %5, vcc\n\t"
                     "v_cndmask_b32_e32 %2, %2, %6, vcc\n\tv_cndmask_b32_e32 %3, %3, %7, vcc"
                     : "+v"(o[g][0]), "+v"(o[g][1]), "+v"(o[g][2]), "+v"(o[g][3])
                     : "v"(a[g][0]), "v"(a[g][1]), "v"(a[g][2]), "v"(a[g][3]), "v"(d), "s"(j) : "vcc");
      } else if constexpr (P == VCMP_CND8) {  // counts as two ops
        asm volatile("v_cmp_eq_u32_e32 vcc, %17, %16\n\tv_cndmask_b32_e32 %0, %0, %8, vcc\n\tv_cndmask_b32_e32 %1, %1, %9, vcc\n\t"
                     "v_cndmask_b32_e32 %2, %2, %10, vcc\n\tv_cndmask_b32_e32 %3, %3, %11, vcc\n\tv_cndmask_b32_e32 %4, %4, %12, vcc\n\t"
                     "v_cndmask_b32_e32 %5, %5, %13, vcc\n\tv_cndmask_b32_e32 %6, %6, %14, vcc\n\tv_cndmask_b32_e32 %7, %7, %15, vcc"
                     : "+v"(o[g][0]), "+v"(o[g][1]), "+v"(o[g][2]), "+v"(o[g][3]), "+v"(o[g][4]), "+v"(o[g][5]), "+v"(o[g][6]), "+v"(o[g][7])
                     : "v"(a[g][0]), "v"(a[g][1]), "v"(a[g][2]), "v"(a[g][3]), "v"(a[g][4]), "v"(a[g][5]), "v"(a[g][6]), "v"(a[g][7]),
                       "v"(d), "s"(j) : "vcc");
      } else if constexpr (P == CND4_E64) {
        const uint64_t m = __builtin_amdgcn_uicmp(d, j, 32);
        asm volatile("v_cndmask_b32_e64 %0, %0, %4, %8\n\tv_cndmask_b32_e64 %1, %1, %5, %8\n\t"
                     "v_cndmask_b32_e64 %2, %2, %6, %8\n\tv_cndmask_b32_e64 %3, %3, %7, %8"
                     : "+v"(o[g][0]), "+v"(o[g][1]), "+v"(o[g][2]), "+v"(o[g][3])
                     : "v"(a[g][0]), "v"(a[g][1]), "v"(a[g][2]), "v"(a[g][3]), "s"(m));
      } else if constexpr (P == BFI4) {
        const uint32_t m = 0u - (uint32_t)(d == j);
        asm volatile("v_bfi_b32 %0, %8, %4, %0\n\tv_bfi_b32 %1, %8, %5, %1\n\tv_bfi_b32 %2, %8, %6, %2\n\tv_bfi_b32 %3, %8, %7, %3"
                     : "+v"(o[g][0]), "+v"(o[g][1]), "+v"(o[g][2]), "+v"(o[g][3])
                     : "v"(a[g][0]), "v"(a[g][1]), "v"(a[g][2]), "v"(a[g][3]), "v"(m));
      } else if constexpr (P == VCMP_CND1) {  // a quarter of an op
        asm volatile("v_cmp_eq_u32_e32 vcc, %3, %2\n\tv_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(o[g][0]) : "v"(a[g][0]), "v"(d), "s"(j) : "vcc");
      } else if constexpr (P == PK_FMA_F32 || P == FMA_F64) {
        // the select as an exact floating-point multiply-add: words tagged with bit 30 are normal binary32 numbers (pairs
        // of them, tagged in the upper word, normal binary64 numbers), e * 1.0 + 0 = e and e * 0.0 + acc = acc exactly:
        // two words per instruction (kernels_ct.hpp ct_scan_lds_pk)
        uint64_t &o0 = o64[g][0], &o1 = o64[g][1];
        const uint64_t a0 = a64[g][0], a1 = a64[g][1];
        if constexpr (P == PK_FMA_F32) {
          const uint32_t m = (d == j) ? 0x3f800000u : 0u;
          const uint64_t m2 = ((uint64_t)m << 32) | m;
          // acc = a * m + acc * (1 - m) would need two instructions; the scan's form is acc = a * m + acc with acc = 0
          // until the one entry whose m is 1: same instruction, same operands' classes
          asm volatile("v_pk_fma_f32 %0, %2, %4, %0\n\tv_pk_fma_f32 %1, %3, %4, %1" : "+v"(o0), "+v"(o1) : "v"(a0), "v"(a1), "v"(m2));
        } else {
          const uint64_t m2 = (d == j) ? 0x3ff0000000000000ull : 0ull;
          asm volatile("v_fma_f64 %0, %2, %4, %0\n\tv_fma_f64 %1, %3, %4, %1" : "+v"(o0), "+v"(o1) : "v"(a0), "v"(a1), "v"(m2));
        }
      } else if constexpr (P == AND_OR4) {
        const uint32_t m = 0u - (uint32_t)(d == j);
        asm volatile("v_and_or_b32 %0, %4, %8, %0\n\tv_and_or_b32 %1, %5, %8, %1\n\tv_and_or_b32 %2, %6, %8, %2\n\tv_and_or_b32 %3, %7, %8, %3"
                     : "+v"(o[g][0]), "+v"(o[g][1]), "+v"(o[g][2]), "+v"(o[g][3])
                     : "v"(a[g][0]), "v"(a[g][1]), "v"(a[g][2]), "v"(a[g][3]), "v"(m));
      } else if constexpr (P >= EXEC_MOV32_DIGIT) {
        // the select as an EXEC-masked move: exec = lane mask, v_mov, exec = all.  A VALU instruction issues whatever EXEC
        // holds -- if the hardware skipped a half-wave whose lanes are all masked off, the time would depend on the
        // mask: four masks (the digit's, all lanes, the lower half only, none) must cost the same for this to be usable.
        uint64_t m;
        if constexpr (P == EXEC_MOV32_DIGIT || P == EXEC_MOV64_DIGIT) m = __builtin_amdgcn_uicmp(d, j, 32);
        else if constexpr (P == EXEC_MOV32_FULL || P == EXEC_MOV64_FULL) m = ~0ull;
        else if constexpr (P == EXEC_MOV32_LOWER || P == EXEC_MOV64_LOWER) m = 0xffffffffull;
        else m = (uint64_t)(j >> 6);  // zero, but not a compile-time constant
        if constexpr (P < EXEC_MOV64_DIGIT) {
          asm volatile("s_mov_b64 exec, %8\n\tv_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7\n\ts_mov_b64 exec, -1"
                       : "+v"(o[g][0]), "+v"(o[g][1]), "+v"(o[g][2]), "+v"(o[g][3])
                       : "v"(a[g][0]), "v"(a[g][1]), "v"(a[g][2]), "v"(a[g][3]), "s"(m));
        } else {
          uint64_t o0 = ((uint64_t)o[g][1] << 32) | o[g][0], o1 = ((uint64_t)o[g][3] << 32) | o[g][2];
          const uint64_t a0 = ((uint64_t)a[g][1] << 32) | a[g][0], a1 = ((uint64_t)a[g][3] << 32) | a[g][2];
          asm volatile("s_mov_b64 exec, %4\n\tv_mov_b64 %0, %2\n\tv_mov_b64 %1, %3\n\ts_mov_b64 exec, -1"
                       : "+v"(o0), "+v"(o1) : "v"(a0), "v"(a1), "s"(m));
          o[g][0] = (uint32_t)o0; o[g][1] = (uint32_t)(o0 >> 32); o[g][2] = (uint32_t)o1; o[g][3] = (uint32_t)(o1 >> 32);
        }
      } else {
        // ds_bpermute_b32 with four index patterns (one op = 4 permutes of one word): is its time the same whatever
        // the lanes ask for?  identity; a random source lane in 0..31; every lane the same source; sources j and
        // j + 32 alternating (two sources on one "bank" if the crossbar is banked like ds_read_b32)
        uint32_t src;
        if constexpr (P == BPERM_ID) src = lane & 63u;
        else if constexpr (P == BPERM_RANDOM) src = (d + j) & 31u;
        else if constexpr (P == BPERM_SAME) src = j;
        else src = ((lane & 1u) ? 32u : 0u) + (j & 31u);
        const uint32_t addr = src << 2;
        asm volatile("ds_bpermute_b32 %0, %4, %0\n\tds_bpermute_b32 %1, %4, %1\n\tds_bpermute_b32 %2, %4, %2\n\tds_bpermute_b32 %3, %4, %3\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "+v"(o[g][0]), "+v"(o[g][1]), "+v"(o[g][2]), "+v"(o[g][3]) : "v"(addr));
      }
    }
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  uint32_t acc = 0;
#pragma unroll
  for (int g = 0; g < GROUPS; ++g)
#pragma unroll
    for (int i = 0; i < 8; ++i) acc ^= o[g][i];
#pragma unroll
  for (int g = 0; g < GROUPS; ++g) acc ^= (uint32_t)(o64[g][0] ^ (o64[g][0] >> 32) ^ o64[g][1] ^ (o64[g][1] >> 32));
  if (acc == 0x12345678u) out[lane] = acc;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clocks[0] = t1 - t0; clocks[1] = r1 - r0; }
}

template <int P>
int run(const char* name, double ops_per_group, uint32_t* dout, uint64_t* dclk, int blocks, int iters) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k_sel<P>, dim3(blocks), dim3(256), 0, 0, dout, dclk, 16, 1u);
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  uint64_t clk[2] = {0, 0};
  for (int r = 0; r < 3; ++r) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_sel<P>, dim3(blocks), dim3(256), 0, 0, dout, dclk, iters, 1u);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) {
      best = ms;
      CK(hipMemcpy(clk, dclk, sizeof(clk), hipMemcpyDeviceToHost));
    }
  }
  const double ghz = clk[1] ? (double)clk[0] / (double)clk[1] * 0.1 : 0.0;  // memrealtime ticks at 100 MHz
  const double wave_ops = (double)blocks * 4.0 * (double)iters * GROUPS * ops_per_group;  // 4 waves per block
  const double cyc = (best * 1e-3) * ghz * 1e9 * 1024.0 / wave_ops;                       // 1024 SIMDs
  printf("{\"pattern\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.3f, \"clock_ghz\": %.3f, \"cycles_per_4word_select_per_simd\": %.2f}\n", name,
         blocks / 256, best, ghz, cyc);
  return 0;
}

int main() {
  uint32_t* dout;
  uint64_t* dclk;
  CK(hipMalloc(&dout, 256 * 64 * 256 * sizeof(uint32_t)));
  CK(hipMalloc(&dclk, 2 * sizeof(uint64_t)));
  const int iters = 4000;
  for (int blocks : {256 * 4, 256 * 8}) {  // 4 or 8 waves per SIMD... 256-thread blocks = 4 waves, one per SIMD
    run<SMOV_CND4>("s_mov_b64 vcc + 4 v_cndmask_e32", 1, dout, dclk, blocks, iters);
    run<VCMP_CND4>("v_cmp_e32 + 4 v_cndmask_e32", 1, dout, dclk, blocks, iters);
    run<VCMP_CND8>("v_cmp_e32 + 8 v_cndmask_e32 (per 4 words)", 2, dout, dclk, blocks, iters);
    run<CND4_E64>("4 v_cndmask_e64, SGPR-pair mask", 1, dout, dclk, blocks, iters);
    run<BFI4>("4 v_bfi_b32, VGPR mask", 1, dout, dclk, blocks, iters);
    run<VCMP_CND1>("v_cmp_e32 + 1 v_cndmask_e32 (per 4 words)", 0.25, dout, dclk, blocks, iters);
    run<EXEC_MOV32_DIGIT>("exec = digit mask, 4 v_mov_b32", 1, dout, dclk, blocks, iters);
    run<EXEC_MOV32_FULL>("exec = all lanes, 4 v_mov_b32", 1, dout, dclk, blocks, iters);
    run<EXEC_MOV32_LOWER>("exec = lower half only, 4 v_mov_b32", 1, dout, dclk, blocks, iters);
    run<EXEC_MOV32_ZERO>("exec = no lane, 4 v_mov_b32", 1, dout, dclk, blocks, iters);
    run<EXEC_MOV64_DIGIT>("exec = digit mask, 2 v_mov_b64", 1, dout, dclk, blocks, iters);
    run<EXEC_MOV64_FULL>("exec = all lanes, 2 v_mov_b64", 1, dout, dclk, blocks, iters);
    run<EXEC_MOV64_LOWER>("exec = lower half only, 2 v_mov_b64", 1, dout, dclk, blocks, iters);
    run<EXEC_MOV64_ZERO>("exec = no lane, 2 v_mov_b64", 1, dout, dclk, blocks, iters);
    run<PK_FMA_F32>("2 v_pk_fma_f32 against (1.0, 1.0) / (0.0, 0.0)", 1, dout, dclk, blocks, iters);
    run<FMA_F64>("2 v_fma_f64 against 1.0 / 0.0", 1, dout, dclk, blocks, iters);
    run<AND_OR4>("4 v_and_or_b32, VGPR mask", 1, dout, dclk, blocks, iters);
    run<BPERM_ID>("4 ds_bpermute_b32, identity", 1, dout, dclk, blocks, iters / 4);
    run<BPERM_RANDOM>("4 ds_bpermute_b32, random source in 0..31", 1, dout, dclk, blocks, iters / 4);
    run<BPERM_SAME>("4 ds_bpermute_b32, every lane the same source", 1, dout, dclk, blocks, iters / 4);
    run<BPERM_FOLD>("4 ds_bpermute_b32, sources j and j+32 alternating", 1, dout, dclk, blocks, iters / 4);
  }
  CK(hipFree(dout));
  CK(hipFree(dclk));
  return 0;
}

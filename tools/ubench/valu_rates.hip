// VALU issue-rate microbenchmark for gfx950 (MI355X).
// Measures lane-ops/s of the integer / fp64 instructions a multi-limb modular
// multiplication can be built from, relative to v_add_u32 (full rate).
// Build: hipcc -O3 --offload-arch=gfx950 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

constexpr int CHAINS = 8;     // independent dependency chains per lane
constexpr int UNROLL = 8;     // ops per chain per loop iteration

enum Op { ADD_U32, ADDC_CHAIN, MAD_U64_U32, MUL_LO_U32, MUL_HI_U32, MAD_U32_U24, MUL_HI_U32_U24, FMA_F64, FMA_F32, ADD3_U32, LSHL_ADD_U64, CNDMASK, MAD_U64_U32_DEP, DOT4_U32_U8, MUL_LO_U16_PK, ALIGNBIT, ADD_LSHL, MAC_PAIR, CMP_CNDMASK, DPP_SHR, BPERMUTE, SUBB_CHAIN, MAD_I64_I32, LSHR64, MOV };

template <int OP>
__global__ void __launch_bounds__(256) k_rate(uint32_t* out, int iters, uint32_t seed) {
  uint32_t lane = threadIdx.x + blockIdx.x * blockDim.x;
  uint32_t a[CHAINS], b[CHAINS];
  uint64_t w[CHAINS];
  double d[CHAINS];
  double dm = 1.0000001 + seed * 1e-9, da = 1e-9;
  float f[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) {
    a[c] = lane * 2654435761u + c * 40503u + seed;
    b[c] = (lane ^ 0x9e3779b9u) + c * 7919u + seed;
    w[c] = ((uint64_t)a[c] << 32) | b[c];
    d[c] = 1.0 + c * 0.125 + (lane & 7) * 0.01;
    f[c] = 1.0f + c * 0.125f;
  }
  uint32_t m = seed | 1u;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) {
        if constexpr (OP == ADD_U32) {
          asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[c]) : "v"(b[c]));
        } else if constexpr (OP == ADDC_CHAIN) {
          asm volatile("v_add_co_u32 %0, vcc, %0, %1\n\tv_addc_co_u32 %2, vcc, %2, %1, vcc" : "+v"(a[c]), "+v"(b[c]) : "v"(m) : "vcc");
        } else if constexpr (OP == MAD_U64_U32) {
          asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[c]) : "v"(a[c]), "v"(b[c]) : "vcc");
        } else if constexpr (OP == MAD_U64_U32_DEP) {
          asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[0]) : "v"(a[c]), "v"(b[c]) : "vcc");
        } else if constexpr (OP == MUL_LO_U32) {
          asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[c]) : "v"(b[c]));
        } else if constexpr (OP == MUL_HI_U32) {
          asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[c]) : "v"(b[c]));
        } else if constexpr (OP == MAD_U32_U24) {
          asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[c]) : "v"(b[c]), "v"(m));
        } else if constexpr (OP == MUL_HI_U32_U24) {
          asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a[c]) : "v"(b[c]));
        } else if constexpr (OP == FMA_F64) {
          asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[c]) : "v"(dm), "v"(da));
        } else if constexpr (OP == FMA_F32) {
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[c]) : "v"(f[(c + 1) % CHAINS]), "v"(f[(c + 2) % CHAINS]));
        } else if constexpr (OP == ADD3_U32) {
          asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[c]) : "v"(b[c]), "v"(m));
        } else if constexpr (OP == LSHL_ADD_U64) {
          asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(w[c]) : "v"(w[(c + 1) % CHAINS]));
        } else if constexpr (OP == CNDMASK) {
          asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[c]) : "v"(b[c]) : );
        } else if constexpr (OP == DOT4_U32_U8) {
          asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(a[c]) : "v"(b[c]), "v"(m));
        } else if constexpr (OP == MUL_LO_U16_PK) {
          asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(a[c]) : "v"(b[c]));
        } else if constexpr (OP == ALIGNBIT) {
          asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(a[c]) : "v"(b[c]));
        } else if constexpr (OP == ADD_LSHL) {
          asm volatile("v_add_lshl_u32 %0, %0, %1, 1" : "+v"(a[c]) : "v"(b[c]));
        } else if constexpr (OP == MAC_PAIR) {   // the field multiplier's unit of work
          asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(w[c]), "+v"(a[(c + 1) % CHAINS]) : "v"(a[c]), "v"(b[c]) : "vcc");
        } else if constexpr (OP == CMP_CNDMASK) {
          asm volatile("v_cmp_lt_u32 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[c]) : "v"(b[c]), "v"(m) : "vcc");
        } else if constexpr (OP == DPP_SHR) {    // cross-lane move inside a row of 16 lanes
          asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[c]) : "v"(b[c]));
        } else if constexpr (OP == BPERMUTE) {   // arbitrary cross-lane gather through the LDS crossbar
          asm volatile("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(a[c]) : "v"(b[c]));
        } else if constexpr (OP == SUBB_CHAIN) {
          asm volatile("v_sub_co_u32 %0, vcc, %0, %1\n\tv_subb_co_u32 %2, vcc, %2, %1, vcc" : "+v"(a[c]), "+v"(b[c]) : "v"(m) : "vcc");
        } else if constexpr (OP == MAD_I64_I32) {
          asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(w[c]) : "v"(a[c]), "v"(b[c]) : "vcc");
        } else if constexpr (OP == LSHR64) {
          asm volatile("v_lshrrev_b64 %0, 29, %0" : "+v"(w[c]));
        } else if constexpr (OP == MOV) {
          asm volatile("v_mov_b32 %0, %1" : "+v"(a[c]) : "v"(b[c]));
        }
      }
    }
  }
  uint32_t acc = 0;
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) {
    acc ^= a[c] ^ b[c] ^ (uint32_t)w[c] ^ (uint32_t)(w[c] >> 32) ^ (uint32_t)d[c] ^ (uint32_t)f[c];
  }
  if (acc == 0x12345678u) out[lane] = acc;  // keep results live
}

template <int OP>
int run(const char* name, int insts_per_op, uint32_t* dout, int blocks, int iters) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(256), 0, 0, dout, 16, 1u);
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 3; ++r) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(256), 0, 0, dout, iters, 1u);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  double lane_ops = (double)blocks * 256 * (double)iters * UNROLL * CHAINS * insts_per_op;
  double rate = lane_ops / (best * 1e-3);
  // per-SIMD cycles per wave-instruction assuming 2.4 GHz and 1024 SIMDs
  double wave_inst = lane_ops / 64.0;
  double cyc = (best * 1e-3) * 2.4e9 * 1024.0 / wave_inst;
  printf("{\"op\": \"%s\", \"blocks\": %d, \"ms\": %.3f, \"lane_Gops\": %.1f, \"cyc_per_waveinst_per_simd_at_2p4GHz\": %.2f}\n",
         name, blocks, best, rate * 1e-9, cyc);
  return 0;
}

int main(int argc, char** argv) {
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  printf("{\"device\": \"%s\", \"arch\": \"%s\", \"cus\": %d, \"clock_khz\": %d, \"lds_per_block\": %zu}\n", p.name, p.gcnArchName, p.multiProcessorCount, p.clockRate, p.sharedMemPerBlock);
  uint32_t* dout;
  CK(hipMalloc(&dout, 256 * 64 * 256 * sizeof(uint32_t)));
  int iters = 4000;
  for (int blocks : {256 * 2, 256 * 8}) {   // 2 or 8 waves per SIMD
    run<ADD_U32>("v_add_u32", 1, dout, blocks, iters);
    run<ADD3_U32>("v_add3_u32", 1, dout, blocks, iters);
    run<ADD_LSHL>("v_add_lshl_u32", 1, dout, blocks, iters);
    run<ALIGNBIT>("v_alignbit_b32", 1, dout, blocks, iters);
    run<CNDMASK>("v_cndmask_b32", 1, dout, blocks, iters);
    run<ADDC_CHAIN>("v_add_co+v_addc_co pair", 2, dout, blocks, iters);
    run<LSHL_ADD_U64>("v_lshl_add_u64", 1, dout, blocks, iters);
    run<MAD_U64_U32>("v_mad_u64_u32", 1, dout, blocks, iters);
    run<MAD_U64_U32_DEP>("v_mad_u64_u32 (single dep chain)", 1, dout, blocks, iters / 4);
    run<MUL_LO_U32>("v_mul_lo_u32", 1, dout, blocks, iters);
    run<MUL_HI_U32>("v_mul_hi_u32", 1, dout, blocks, iters);
    run<MAD_U32_U24>("v_mad_u32_u24", 1, dout, blocks, iters);
    run<MUL_HI_U32_U24>("v_mul_hi_u32_u24", 1, dout, blocks, iters);
    run<DOT4_U32_U8>("v_dot4_u32_u8", 1, dout, blocks, iters);
    run<MUL_LO_U16_PK>("v_pk_mul_lo_u16", 1, dout, blocks, iters);
    run<MAC_PAIR>("MAC = v_mad_u64_u32 + v_addc_co_u32 (per pair)", 1, dout, blocks, iters);
    run<CMP_CNDMASK>("v_cmp_lt_u32 + v_cndmask_b32 (per pair)", 1, dout, blocks, iters);
    run<SUBB_CHAIN>("v_sub_co+v_subb_co pair", 2, dout, blocks, iters);
    run<MAD_I64_I32>("v_mad_i64_i32", 1, dout, blocks, iters);
    run<LSHR64>("v_lshrrev_b64", 1, dout, blocks, iters);
    run<MOV>("v_mov_b32", 1, dout, blocks, iters);
    run<DPP_SHR>("v_mov_b32_dpp row_shr:1", 1, dout, blocks, iters);
    run<BPERMUTE>("ds_bpermute_b32 (+wait)", 1, dout, blocks, iters / 4);
    run<FMA_F32>("v_fma_f32", 1, dout, blocks, iters);
    run<FMA_F64>("v_fma_f64", 1, dout, blocks, iters);
  }
  CK(hipFree(dout));
  return 0;
}

// Test-only library: runs single operations of the unsaturated field layer
// (eccoxide_amd/csrc/ufe.hpp) on raw limb arrays so that tests/test_field_layer.py can feed
// them the worst limbs their type bounds admit and compare with Python big-int arithmetic.
// Not part of the product; built by __graft_entry__.build() into tests/hip/libfieldcheck.so.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "curve.hpp"
#include "inv_gcd.hpp"
#include "ufe.hpp"

namespace eccx {

// operand bounds used by the checks: the laziest the multiplier admits without reducing
template <class C>
struct Chk {
  static constexpr int KK = UB<C>::KKMAX;
  static constexpr int KA = KK >= 16 ? 4 : (KK >= 6 ? 3 : 1);  // product operand a
  static constexpr int KB = KK >= 16 ? 4 : (KK >= 6 ? 2 : 1);  // product operand b
  static constexpr int KS = UB<C>::ksq_ok(4) ? 4 : (UB<C>::ksq_ok(2) ? 2 : 1);
  static constexpr int VA = 7, VB = 5, VS = 4;
  static_assert(UB<C>::kk_ok(KA, KB), "check bounds");
};

enum : int { OP_MUL_TIGHT = 0, OP_MUL_LAZY = 1, OP_SQR_LAZY = 2, OP_SUB_CHAIN = 3, OP_REDUCE_MAX = 4,
             OP_CANONICAL = 5, OP_MUL_AUTO = 6, OP_ADD_AUTO = 7, OP_REDUCE_LAZY = 8, OP_INVERT = 9,
             // merged products, four operands (a, b, c, d): one Montgomery reduction for two products
             OP_MUL_ADD = 10,       // general Montgomery (BLS12-381): a*b + (4p - c)*d, tight c
             OP_MUL_ADD_MAX = 11,   //   a*b + c*d at the column budget (K 3*3 + 2*4 = 17)
             OP_MUL_SUB = 12,       // signed columns (P-384 sparse, BLS12-381 general): a*b - c*d (+ p)
             OP_MUL_SUB_MAX = 13,   //   both products at the column budget (K 2*4 each)
             OP_MUL_SUB_2SQR = 14 };//   a*b - 2*c^2 (+ p), the doubling's Y3

template <class C, int K, int V>
__device__ U<C, K, V> load_u(const uint32_t* p) {
  U<C, K, V> r;
#pragma unroll
  for (int i = 0; i < C::N; ++i) r.v[i] = p[i];
  return r;
}
template <class C, int K, int V>
__device__ void store_u(uint32_t* p, const U<C, K, V>& a) {
#pragma unroll
  for (int i = 0; i < C::N; ++i) p[i] = a.v[i];
}

template <class C>
__global__ void k_field_check(int op, const uint32_t* __restrict__ a, const uint32_t* __restrict__ b,
                              uint32_t* __restrict__ out, size_t n, const uint32_t* __restrict__ c = nullptr,
                              const uint32_t* __restrict__ d = nullptr) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  constexpr int N = C::N;
  constexpr int KM = UB<C>::KMAX;
  const uint32_t* pa = a + i * N;
  const uint32_t* pb = b + i * N;
  const uint32_t* pc = c ? c + i * N : pa;
  const uint32_t* pd = d ? d + i * N : pb;
  uint32_t* po = out + i * N;
  switch (op) {
    case OP_MUL_ADD:
      if constexpr (C::KIND == UK_MONT)
        store_u(po, u_mul_add(load_u<C, 1, 3>(pa), load_u<C, 3, 6>(pb), u_neg(load_u<C, 1, 3>(pc)), load_u<C, 1, 3>(pd)));
      break;
    case OP_MUL_ADD_MAX:
      if constexpr (C::KIND == UK_MONT)
        store_u(po, u_mul_add(load_u<C, 3, 6>(pa), load_u<C, 3, 6>(pb), load_u<C, 2, 4>(pc), load_u<C, 4, 8>(pd)));
      break;
    case OP_MUL_SUB:
      if constexpr (UBS<C>::OK)
        store_u(po, u_mul_sub(load_u<C, 1, 3>(pa), load_u<C, 3, 6>(pb), load_u<C, 1, 3>(pc), load_u<C, 1, 3>(pd)));
      break;
    case OP_MUL_SUB_MAX:
      if constexpr (UBS<C>::OK)
        store_u(po, u_mul_sub(load_u<C, 2, 4>(pa), load_u<C, 4, 8>(pb), load_u<C, 2, 4>(pc), load_u<C, 4, 8>(pd)));
      break;
    case OP_MUL_SUB_2SQR:
      if constexpr (UBS<C>::OK)
        store_u(po, u_mul_sub_2sqr(load_u<C, 2, 4>(pa), load_u<C, 4, 8>(pb), load_u<C, 2, 4>(pc)));
      break;
    case OP_MUL_TIGHT: store_u(po, u_mul(load_u<C, 1, 3>(pa), load_u<C, 1, 3>(pb))); break;
    case OP_MUL_LAZY:
      store_u(po, u_mul(load_u<C, Chk<C>::KA, Chk<C>::VA>(pa), load_u<C, Chk<C>::KB, Chk<C>::VB>(pb)));
      break;
    case OP_SQR_LAZY: store_u(po, u_sqr(load_u<C, Chk<C>::KS, Chk<C>::VS>(pa))); break;
    case OP_SUB_CHAIN: {  // ((a - b) - b) - b on tight operands, then the weak reduction
      auto x = load_u<C, 1, 3>(pa);
      auto y = load_u<C, 1, 3>(pb);
      store_u(po, u_reduce(u_sub(u_sub(u_sub(x, y), y), y)));
      break;
    }
    case OP_REDUCE_MAX: store_u(po, u_reduce(load_u<C, KM, 64>(pa))); break;
    case OP_CANONICAL: {  // out = canonical plain integer, saturated limbs, zero padded to N words
      Fe<C::Sat::L> s;
      u_to_canonical<C>(s, load_u<C, KM, 64>(pa));
#pragma unroll
      for (int k = 0; k < N; ++k) po[k] = k < C::Sat::L ? s.v[k] : 0u;
      break;
    }
    case OP_MUL_AUTO:  // operands far beyond what the columns take: the wrapper must reduce them
      store_u(po, u_mul(load_u<C, KM, 64>(pa), load_u<C, KM, 64>(pb)));
      break;
    case OP_ADD_AUTO: {  // sums that would overflow 32 bits without the automatic reduction
      auto x = load_u<C, KM, 64>(pa);
      auto y = load_u<C, KM, 64>(pb);
      store_u(po, u_reduce(u_add(u_add(x, y), u_add(x, y))));
      break;
    }
    case OP_INVERT: {  // a: saturated limbs of a canonical plain integer (zero padded to N words); out likewise
      Fe<C::Sat::L> x, y;
#pragma unroll
      for (int k = 0; k < C::Sat::L; ++k) x.v[k] = pa[k];
      fe_inv_gcd<typename C::Sat>(y, x);
#pragma unroll
      for (int k = 0; k < N; ++k) po[k] = k < C::Sat::L ? y.v[k] : 0u;
      break;
    }
    case OP_REDUCE_LAZY:  // the bounds under which the Solinas primes reduce with shifts of q
      store_u(po, u_reduce(load_u<C, KM - 1, 48>(pa)));
      break;
    default: break;
  }
}

}  // namespace eccx

extern "C" {

// bounds the Python side needs to build its inputs: fills info[0..7] = N, B, KMAX, KKMAX, KA, KB, KS, L
int fieldcheck_info(int curve, int* info) {
  using namespace eccx;
#define ECCX_INFO(C)                                                                               \
  {                                                                                                \
    info[0] = C::N; info[1] = C::B; info[2] = UB<C>::KMAX; info[3] = UB<C>::KKMAX;                 \
    info[4] = Chk<C>::KA; info[5] = Chk<C>::KB; info[6] = Chk<C>::KS; info[7] = C::Sat::L;         \
    return 0;                                                                                      \
  }
  switch (curve) {
    case 0: ECCX_INFO(P256U)
    case 1: ECCX_INFO(P384U)
    case 2: ECCX_INFO(P521U)
    case 3: ECCX_INFO(BLS12_381U)
    case 4: ECCX_INFO(ED25519U)
    default: return -1;
  }
#undef ECCX_INFO
}

int fieldcheck_run4(int curve, int op, const uint32_t* a, const uint32_t* b, const uint32_t* c, const uint32_t* d,
                    uint32_t* out, size_t n);
// runs one operation over n rows of N limbs (host pointers); returns 0 or a hipError_t
int fieldcheck_run(int curve, int op, const uint32_t* a, const uint32_t* b, uint32_t* out, size_t n) {
  return fieldcheck_run4(curve, op, a, b, nullptr, nullptr, out, n);
}
// the same with two more operand arrays (merged products); c / d may be null
int fieldcheck_run4(int curve, int op, const uint32_t* a, const uint32_t* b, const uint32_t* c, const uint32_t* d,
                    uint32_t* out, size_t n) {
  using namespace eccx;
  int info[8];
  if (fieldcheck_info(curve, info)) return -1;
  const size_t bytes = n * (size_t)info[0] * sizeof(uint32_t);
  uint32_t *da = nullptr, *db = nullptr, *dout = nullptr, *dc = nullptr, *dd = nullptr;
  hipError_t e;
  if ((e = hipMalloc(&da, bytes)) != hipSuccess) return (int)e;
  if ((e = hipMalloc(&db, bytes)) != hipSuccess) { (void)hipFree(da); return (int)e; }
  if ((e = hipMalloc(&dout, bytes)) != hipSuccess) { (void)hipFree(da); (void)hipFree(db); return (int)e; }
  if (c && d) {
    if ((e = hipMalloc(&dc, bytes)) != hipSuccess || (e = hipMalloc(&dd, bytes)) != hipSuccess) {
      (void)hipFree(da); (void)hipFree(db); (void)hipFree(dout); if (dc) (void)hipFree(dc);
      return (int)e;
    }
    (void)hipMemcpy(dc, c, bytes, hipMemcpyHostToDevice);
    (void)hipMemcpy(dd, d, bytes, hipMemcpyHostToDevice);
  }
  (void)hipMemcpy(da, a, bytes, hipMemcpyHostToDevice);
  (void)hipMemcpy(db, b, bytes, hipMemcpyHostToDevice);
  (void)hipMemset(dout, 0, bytes);
  const int wg = 128;
  const int grid = (int)((n + wg - 1) / wg);
  switch (curve) {
    case 0: hipLaunchKernelGGL(k_field_check<P256U>, dim3(grid), dim3(wg), 0, 0, op, da, db, dout, n, dc, dd); break;
    case 1: hipLaunchKernelGGL(k_field_check<P384U>, dim3(grid), dim3(wg), 0, 0, op, da, db, dout, n, dc, dd); break;
    case 2: hipLaunchKernelGGL(k_field_check<P521U>, dim3(grid), dim3(wg), 0, 0, op, da, db, dout, n, dc, dd); break;
    case 3: hipLaunchKernelGGL(k_field_check<BLS12_381U>, dim3(grid), dim3(wg), 0, 0, op, da, db, dout, n, dc, dd); break;
    case 4: hipLaunchKernelGGL(k_field_check<ED25519U>, dim3(grid), dim3(wg), 0, 0, op, da, db, dout, n, dc, dd); break;
    default: break;
  }
  e = hipGetLastError();
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpy(out, dout, bytes, hipMemcpyDeviceToHost);
  (void)hipFree(da);
  (void)hipFree(db);
  (void)hipFree(dout);
  if (dc) (void)hipFree(dc);
  if (dd) (void)hipFree(dd);
  return (int)e;
}
}

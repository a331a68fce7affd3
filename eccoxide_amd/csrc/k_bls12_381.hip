// Kernel instantiations for BLS12_381 (see kernels.hpp).
#include "kernels_fast.hpp"
#include "launch.hpp"

namespace eccx {
namespace {
hipError_t var_(int grid, hipStream_t s, size_t n, const uint8_t* scalars, const uint8_t* points, uint8_t* out,
                uint8_t* flags, uint8_t* proj, uint32_t* scratch, uint32_t opts) {
  hipLaunchKernelGGL(k_scalarmul_var<BLS12_381>, dim3(grid), dim3(WG), 0, s, n, scalars, points, out, flags, proj, scratch, opts);
  return hipGetLastError();
}
hipError_t base_(int grid, hipStream_t s, size_t n, const uint8_t* scalars, const uint32_t* table, uint8_t* out,
                 uint8_t* flags, uint8_t* proj, uint32_t opts) {
  hipLaunchKernelGGL(k_scalarmul_base<BLS12_381>, dim3(grid), dim3(WG), 0, s, n, scalars, table, out, flags, proj, opts);
  return hipGetLastError();
}
hipError_t var_fast_(int grid, hipStream_t s, size_t n, const uint8_t* scalars, const uint8_t* points, uint32_t* jac,
                     uint8_t* flags, uint32_t* scratch, uint32_t opts) {
  hipLaunchKernelGGL(k_scalarmul_var_fast<BLS12_381>, dim3(grid), dim3(WG), 0, s, n, scalars, points, jac, flags, scratch, opts);
  return hipGetLastError();
}
hipError_t to_affine_jac_(int grid, hipStream_t s, size_t n, const uint32_t* jac, uint8_t* out, uint8_t* flags) {
  hipLaunchKernelGGL((k_batch_to_affine<BLS12_381, NORM_JACOBIAN, TO_AFFINE_U>), dim3(grid), dim3(WG), 0, s, n, jac, out, flags);
  return hipGetLastError();
}
hipError_t base_fast_(int grid, hipStream_t s, size_t n, const uint8_t* scalars, const uint32_t* table, uint32_t* jac,
                      uint8_t* flags) {
  hipLaunchKernelGGL(k_scalarmul_base_fast<BLS12_381>, dim3(grid), dim3(WG), 0, s, n, scalars, table, jac, flags);
  return hipGetLastError();
}
hipError_t to_affine_hom_(int grid, hipStream_t s, size_t n, const uint32_t* rows, uint8_t* out, uint8_t* flags) {
  hipLaunchKernelGGL((k_batch_to_affine<BLS12_381, NORM_HOMOGENEOUS, TO_AFFINE_U>), dim3(grid), dim3(WG), 0, s, n, rows, out, flags);
  return hipGetLastError();
}
int var_grid_(int cus, size_t n) {
  static const int occ = occupancy_per_cu(k_scalarmul_var<BLS12_381>);
  return persistent_grid(occ, cus, n);
}
int var_fast_grid_(int cus, size_t n) {
  static const int occ = occupancy_per_cu(k_scalarmul_var_fast<BLS12_381>);
  return persistent_grid(occ, cus, n);
}
hipError_t point_add_(int grid, hipStream_t s, size_t n, const uint8_t* a, const uint8_t* a_inf, const uint8_t* b,
                      const uint8_t* b_inf, uint32_t* rows, uint8_t* flags, uint32_t opts) {
  hipLaunchKernelGGL(k_point_add<BLS12_381>, dim3(grid), dim3(WG), 0, s, n, a, a_inf, b, b_inf, rows, flags, opts);
  return hipGetLastError();
}
}  // namespace
const CurveOps& ops_BLS12_381() {
  static const CurveOps o = {{BLS12_381::FB, BLS12_381::SB, BLS12_381::L, 2 * BLS12_381::L, row_words<BLS12_381::L>(), 0, row5_words<BLS12_381::L>(), row_words<BLS12_381::L>()}, var_, base_, var_fast_, to_affine_jac_, base_fast_, nullptr, to_affine_hom_, var_grid_, var_fast_grid_, to_affine_jac_, point_add_};
  return o;
}
}  // namespace eccx

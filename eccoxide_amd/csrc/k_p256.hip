// Kernel instantiations for P256 (see kernels.hpp).
#include "kernels_u29.hpp"
#include "launch.hpp"

namespace eccx {
namespace {
hipError_t var_(int grid, hipStream_t s, size_t n, const uint8_t* scalars, const uint8_t* points, uint8_t* out,
                uint8_t* flags, uint8_t* proj, uint32_t* scratch, uint32_t opts) {
  hipLaunchKernelGGL(k_scalarmul_var<P256>, dim3(grid), dim3(WG), 0, s, n, scalars, points, out, flags, proj, scratch, opts);
  return hipGetLastError();
}
hipError_t base_(int grid, hipStream_t s, size_t n, const uint8_t* scalars, const uint32_t* table, uint8_t* out,
                 uint8_t* flags, uint8_t* proj, uint32_t opts) {
  hipLaunchKernelGGL(k_scalarmul_base<P256>, dim3(grid), dim3(WG), 0, s, n, scalars, table, out, flags, proj, opts);
  return hipGetLastError();
}
hipError_t var_fast_(int grid, hipStream_t s, size_t n, const uint8_t* scalars, const uint8_t* points, uint32_t* jac,
                     uint8_t* flags, uint32_t* scratch, uint32_t opts) {
  // P-256 runs the unsaturated-field ladder (kernels_u29.hpp); rows leave as plain integers
  hipLaunchKernelGGL(k_scalarmul_var_u29<P256U>, dim3(grid), dim3(WG), 0, s, n, scalars, points, jac, flags, scratch, opts);
  return hipGetLastError();
}
hipError_t to_affine_var_(int grid, hipStream_t s, size_t n, const uint32_t* jac, uint8_t* out, uint8_t* flags) {
  hipLaunchKernelGGL((k_batch_to_affine<P256, NORM_JACOBIAN, TO_AFFINE_U, true>), dim3(grid), dim3(WG), 0, s, n, jac, out, flags);
  return hipGetLastError();
}
hipError_t to_affine_jac_(int grid, hipStream_t s, size_t n, const uint32_t* jac, uint8_t* out, uint8_t* flags) {
  hipLaunchKernelGGL((k_batch_to_affine<P256, NORM_JACOBIAN, TO_AFFINE_U>), dim3(grid), dim3(WG), 0, s, n, jac, out, flags);
  return hipGetLastError();
}
hipError_t base_fast_(int grid, hipStream_t s, size_t n, const uint8_t* scalars, const uint32_t* table, uint32_t* jac,
                      uint8_t* flags) {
  hipLaunchKernelGGL(k_scalarmul_base_fast<P256>, dim3(grid), dim3(WG), 0, s, n, scalars, table, jac, flags);
  return hipGetLastError();
}
hipError_t to_affine_hom_(int grid, hipStream_t s, size_t n, const uint32_t* rows, uint8_t* out, uint8_t* flags) {
  hipLaunchKernelGGL((k_batch_to_affine<P256, NORM_HOMOGENEOUS, TO_AFFINE_U>), dim3(grid), dim3(WG), 0, s, n, rows, out, flags);
  return hipGetLastError();
}
int var_grid_(int cus, size_t n) {
  static const int occ = occupancy_per_cu(k_scalarmul_var<P256>);
  return persistent_grid(occ, cus, n);
}
int var_fast_grid_(int cus, size_t n) {
  static const int occ = occupancy_per_cu(k_scalarmul_var_u29<P256U>);
  return persistent_grid(occ, cus, n);
}
hipError_t point_add_(int grid, hipStream_t s, size_t n, const uint8_t* a, const uint8_t* a_inf, const uint8_t* b,
                      const uint8_t* b_inf, uint32_t* rows, uint8_t* flags, uint32_t opts) {
  hipLaunchKernelGGL(k_point_add<P256>, dim3(grid), dim3(WG), 0, s, n, a, a_inf, b, b_inf, rows, flags, opts);
  return hipGetLastError();
}
}  // namespace
const CurveOps& ops_P256() {
  static const CurveOps o = {{P256::FB, P256::SB, P256::L, 2 * P256::L, row_words<P256::L>(), 0, urow_words<P256U>(), row_words<P256::L>()}, var_, base_, var_fast_, to_affine_jac_, base_fast_, nullptr, to_affine_hom_, var_grid_, var_fast_grid_, to_affine_var_, point_add_};
  return o;
}
}  // namespace eccx

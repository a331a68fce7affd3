// Kernel instantiations for BLS12_381 (see k_weierstrass.inc) + the a = 0 / endomorphism kernels (kernels_bls.hpp).
#define ECCX_CURVE BLS12_381
#define ECCX_CURVE_U BLS12_381U
#define ECCX_OPS_NAME ops_BLS12_381
#define ECCX_CODEC_FORMAT FORMAT_ZCASH
#include "kernels_bls.hpp"
#include "launch.hpp"

namespace eccx {
namespace {
hipError_t var_coz_(int grid, hipStream_t s, size_t n, const uint8_t* scalars, const uint8_t* points, uint32_t* rows,
                    uint8_t* flags, uint32_t* scratch, uint32_t opts, int glv) {
  if (glv) {
    hipLaunchKernelGGL((k_scalarmul_coz_unsat<BLS12_381U, BLS12_381_GLV, true>), dim3(grid), dim3(WG), 0, s, n, scalars, points, rows,
                       flags, scratch, opts);
  } else {
    hipLaunchKernelGGL((k_scalarmul_coz_unsat<BLS12_381U, BLS12_381_GLV, false>), dim3(grid), dim3(WG), 0, s, n, scalars, points, rows,
                       flags, scratch, opts);
  }
  return hipGetLastError();
}
int var_coz_grid_(int cus, size_t n, int glv) {
  static const int occ_glv = occupancy_per_cu(k_scalarmul_coz_unsat<BLS12_381U, BLS12_381_GLV, true>);
  static const int occ = occupancy_per_cu(k_scalarmul_coz_unsat<BLS12_381U, BLS12_381_GLV, false>);
  return persistent_grid(glv ? occ_glv : occ, cus, n);
}
hipError_t subgroup_check_(int grid, hipStream_t s, size_t n, uint8_t* xy, uint8_t* flags) {
  hipLaunchKernelGGL((k_bls_subgroup_check<BLS12_381U, BLS12_381_GLV>), dim3(grid), dim3(WG), 0, s, n, xy, flags);
  return hipGetLastError();
}
}  // namespace
}  // namespace eccx
#define ECCX_EXTRA_OPS(t)                                   \
  do {                                                      \
    (t).var_coz = var_coz_;                                 \
    (t).var_coz_grid = var_coz_grid_;                       \
    (t).coz_row_words = urowc_words<BLS12_381U>();          \
    (t).subgroup_check = subgroup_check_;                   \
  } while (0)
#include "k_weierstrass.inc"

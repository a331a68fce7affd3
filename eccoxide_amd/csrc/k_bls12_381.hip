// Kernel instantiations for BLS12_381 (see k_weierstrass.inc) + the a = 0 / endomorphism kernels (kernels_bls.hpp).
#define ECCX_CURVE BLS12_381
#define ECCX_CURVE_U BLS12_381U
#define ECCX_OPS_NAME ops_BLS12_381
#define ECCX_CODEC_FORMAT FORMAT_ZCASH
#define ECCX_GLV_PARAMS BLS12_381_GLV
#include "kernels_bls.hpp"
#include "launch.hpp"

namespace eccx {
namespace {
hipError_t subgroup_check_(int grid, hipStream_t s, size_t n, uint8_t* xy, uint8_t* flags) {
  hipLaunchKernelGGL((k_bls_subgroup_check<BLS12_381U, BLS12_381_GLV>), dim3(grid), dim3(WG), 0, s, n, xy, flags);
  return hipGetLastError();
}
}  // namespace
}  // namespace eccx
#define ECCX_EXTRA_OPS(t)                                   \
  do {                                                      \
    (t).subgroup_check = subgroup_check_;                   \
  } while (0)
#include "k_weierstrass.inc"

// Kernel instantiations for edwards25519 (see kernels.hpp).
#include "kernels.hpp"
#include "launch.hpp"

namespace eccx {
namespace {
hipError_t var_(int grid, hipStream_t s, size_t n, const uint8_t* scalars, const uint8_t* points, uint8_t* out,
                uint8_t* flags, uint8_t* proj, uint32_t* /*scratch*/, uint32_t opts) {
  hipLaunchKernelGGL(k_ed_scalarmul_var<ED25519>, dim3(grid), dim3(WG), 0, s, n, scalars, points, out, flags, proj, opts);
  return hipGetLastError();
}
hipError_t base_(int grid, hipStream_t s, size_t n, const uint8_t* scalars, const uint32_t* table, uint8_t* out,
                 uint8_t* flags, uint8_t* proj, uint32_t opts) {
  hipLaunchKernelGGL(k_ed_scalarmul_base<ED25519>, dim3(grid), dim3(WG), 0, s, n, scalars, table, out, flags, proj, opts);
  return hipGetLastError();
}
}  // namespace
const CurveOps& ops_ED25519() {
  static const CurveOps o = {{ED25519::FB, ED25519::SB, ED25519::L, 3 * ED25519::L, 0, 1, 0, 0}, var_, base_, nullptr, nullptr};
  return o;
}
}  // namespace eccx

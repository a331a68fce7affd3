// Kernel instantiations for edwards25519 (see kernels.hpp).
#include "kernels_codec.hpp"
#include "kernels_ct.hpp"
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
hipError_t base_lds_(int cus, hipStream_t s, size_t n, const uint8_t* scalars, const uint32_t* table, uint32_t* rows,
                     uint8_t* flags) {
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ed_scalarmul_base_lds6<ED25519U>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)ED_LDS_BYTES);
  if (attr != hipSuccess) return attr;
  size_t need = (n + ED_LDS_BLOCK - 1) / ED_LDS_BLOCK;
  int grid = (int)(need < (size_t)cus ? (need ? need : 1) : (size_t)cus);  // one workgroup per CU
  hipLaunchKernelGGL(k_ed_scalarmul_base_lds6<ED25519U>, dim3(grid), dim3(ED_LDS_BLOCK), ED_LDS_BYTES, s, n, scalars, table,
                     rows, flags);
  return hipGetLastError();
}
hipError_t lds_convert_(hipStream_t s, size_t entries, const uint8_t* affine, uint32_t* table) {
  hipLaunchKernelGGL(k_ed_affine_to_niels_unsat<ED25519U>, dim3((unsigned)((entries + 127) / 128)), dim3(128), 0, s, entries, affine,
                     table, ED_LDS_ENTRY_WORDS);
  return hipGetLastError();
}
hipError_t to_affine_var_(int grid, hipStream_t s, size_t n, const uint32_t* rows, uint8_t* out, uint8_t* flags) {
  hipLaunchKernelGGL((k_batch_to_affine_unsat<ED25519U, NORM_EDWARDS, to_affine_u(ED25519::L)>), dim3(grid), dim3(WG), 0, s, n, rows, out,
                     flags);
  return hipGetLastError();
}
hipError_t to_affine_hom_(int grid, hipStream_t s, size_t n, const uint32_t* rows, uint8_t* out, uint8_t* flags) {
  hipLaunchKernelGGL((k_batch_to_affine<ED25519, NORM_EDWARDS, to_affine_u(ED25519::L)>), dim3(grid), dim3(WG), 0, s, n, rows, out, flags);
  return hipGetLastError();
}
hipError_t comb_convert_(hipStream_t s, size_t entries, const uint8_t* affine, uint32_t* table) {
  hipLaunchKernelGGL(k_ed_affine_to_niels_unsat<ED25519U>, dim3((unsigned)((entries + 127) / 128)), dim3(128), 0, s, entries, affine, table, ED_U_ENTRY_WORDS);
  return hipGetLastError();
}
hipError_t base_w8_(int grid, hipStream_t s, size_t n, const uint8_t* scalars, const uint32_t* table, uint32_t* rows,
                    uint8_t* flags) {
  hipLaunchKernelGGL(k_ed_scalarmul_base_unsat<ED25519U>, dim3(grid), dim3(WG), 0, s, n, scalars, table, rows, flags);
  return hipGetLastError();
}
hipError_t var_fast_(int grid, hipStream_t s, size_t n, const uint8_t* scalars, const uint8_t* points, uint32_t* rows,
                     uint8_t* flags, uint32_t* scratch, uint32_t opts) {
  hipLaunchKernelGGL((k_ed_scalarmul_var_unsat<ED25519U, false>), dim3(grid), dim3(WG), 0, s, n, scalars, points, rows, flags, scratch,
                     opts, nullptr, nullptr);
  return hipGetLastError();
}
hipError_t var_fused_(int grid, hipStream_t s, size_t n, const uint8_t* u2, const uint8_t* q, uint32_t* rows, uint8_t* flags,
                      uint32_t* scratch, uint32_t opts, const uint8_t* u1, const uint32_t* utable) {
  hipLaunchKernelGGL((k_ed_scalarmul_var_unsat<ED25519U, true>), dim3(grid), dim3(WG), 0, s, n, u2, q, rows, flags, scratch, opts,
                     u1, utable);
  return hipGetLastError();
}
int var_fast_grid_(int cus, size_t n) {
  static const int occ = occupancy_per_cu(k_ed_scalarmul_var_unsat<ED25519U, false>);
  return persistent_grid(occ, cus, n);
}
int var_grid_(int cus, size_t n) {
  static const int occ = occupancy_per_cu(k_ed_scalarmul_var<ED25519>);
  return persistent_grid(occ, cus, n);
}
hipError_t point_add_(int grid, hipStream_t s, size_t n, const uint8_t* a, const uint8_t* a_fl, const uint8_t* b,
                      const uint8_t* b_fl, uint32_t* rows, uint8_t* flags, uint32_t opts) {
  hipLaunchKernelGGL(k_ed_point_add<ED25519>, dim3(grid), dim3(WG), 0, s, n, a, a_fl, b, b_fl, rows, flags, opts);
  return hipGetLastError();
}
hipError_t to_affine_add_u_(int grid, hipStream_t s, size_t n, const uint32_t* rows, uint8_t* out, uint8_t* flags) {
  hipLaunchKernelGGL((k_batch_to_affine_unsat<ED25519U, NORM_EDWARDS, to_affine_add_u(ED25519::L)>), dim3(grid), dim3(WG), 0, s, n, rows,
                     out, flags);
  return hipGetLastError();
}
hipError_t point_add_u_(int grid, hipStream_t s, size_t n, const uint8_t* a, const uint8_t* a_fl, const uint8_t* b,
                        const uint8_t* b_fl, uint32_t* rows, uint8_t* flags, uint32_t opts) {
  hipLaunchKernelGGL(k_ed_point_add_unsat<ED25519U>, dim3(grid), dim3(WG), 0, s, n, a, a_fl, b, b_fl, rows, flags, opts);
  return hipGetLastError();
}
hipError_t decompress_(int grid, hipStream_t s, size_t n, const uint8_t* enc, uint8_t* out, uint8_t* flags) {
  hipLaunchKernelGGL(k_ed_point_decompress<ED25519U>, dim3(grid), dim3(WG), 0, s, n, enc, out, flags);
  return hipGetLastError();
}
hipError_t compress_(int grid, hipStream_t s, size_t n, const uint8_t* xy, const uint8_t* /*inf: the identity is (0, 1)*/, uint8_t* out) {
  hipLaunchKernelGGL((k_point_compress<ED25519, FORMAT_RFC8032>), dim3(grid), dim3(WG), 0, s, n, xy, nullptr, out);
  return hipGetLastError();
}
// secret scalars (ECCX_CT_SCAN), variable base: the windowed ladder with every table row read at every lookup
hipError_t var_ct_(int grid, hipStream_t s, size_t n, const uint8_t* scalars, const uint8_t* points, uint32_t* rows,
                   uint8_t* flags, uint32_t* scratch, uint32_t opts) {
  hipLaunchKernelGGL((k_ed_scalarmul_var_unsat<ED25519U, false, ECCX_CT_ED_VAR_BITS, true>), dim3(grid), dim3(WG), 0, s, n, scalars, points,
                     rows, flags, scratch, opts, nullptr, nullptr);
  return hipGetLastError();
}
int var_ct_grid_(int cus, size_t n) {
  static const int occ = occupancy_per_cu(k_ed_scalarmul_var_unsat<ED25519U, false, ECCX_CT_ED_VAR_BITS, true>);
  return persistent_grid(occ, cus, n);
}
// secret scalars (ECCX_CT_SCAN): signed windows, every entry of the window read (kernels_ct.hpp)
hipError_t ct_convert_(hipStream_t s, size_t entries, const uint8_t* affine, uint32_t* table) {
  hipLaunchKernelGGL(k_ed_affine_to_niels_unsat<ED25519U>, dim3((unsigned)((entries + 127) / 128)), dim3(128), 0, s, entries, affine,
                     table, ct_entry_words<ED25519U>());
  return hipGetLastError();
}
hipError_t base_ct_(int grid, hipStream_t s, size_t n, const uint8_t* scalars, const uint32_t* table, uint32_t* rows,
                    uint8_t* flags) {
  hipLaunchKernelGGL((k_ed_scalarmul_base_ct<ED25519U, false>), dim3(grid), dim3(WG), 0, s, n, scalars, table, rows, flags);
  return hipGetLastError();
}
hipError_t base_ctg_(int grid, hipStream_t s, size_t n, const uint8_t* scalars, const uint32_t* table, uint32_t* rows,
                     uint8_t* flags) {
  hipLaunchKernelGGL((k_ed_scalarmul_base_ct<ED25519U, true>), dim3(grid), dim3(WG), 0, s, n, scalars, table, rows, flags);
  return hipGetLastError();
}
}  // namespace
hipError_t launch_x25519_ladder(int grid, hipStream_t s, size_t n, const uint8_t* scalars, const uint8_t* u, uint32_t* rows,
                                uint8_t* flags, uint32_t opts) {
  hipLaunchKernelGGL(k_x25519_ladder_unsat<ED25519U>, dim3(grid), dim3(WG), 0, s, n, scalars, u, rows, flags, opts);
  return hipGetLastError();
}
hipError_t launch_x25519_to_u(int grid, hipStream_t s, size_t n, const uint32_t* rows, uint8_t* out, uint8_t* flags) {
  hipLaunchKernelGGL((k_batch_to_affine_unsat<ED25519U, NORM_MONTGOMERY_U, to_affine_u(ED25519::L)>), dim3(grid), dim3(WG), 0, s, n, rows,
                     out, flags);
  return hipGetLastError();
}
const CurveOps& ops_ED25519() {
  static const CurveOps o = [] {
    CurveOps t = {{ED25519::FB, ED25519::SB, ED25519::L, 4 * ED25519::L, 0, 1, ED_VAR_ROW_WORDS,
                              (urow3_words<ED25519U>() > row_words<ED25519::L>() ? urow3_words<ED25519U>() : row_words<ED25519::L>())}, var_, base_, var_fast_, base_lds_, to_affine_hom_, var_grid_, var_fast_grid_, to_affine_var_, point_add_, ED_U_ENTRY_WORDS, comb_bits<ED25519U>(), comb_convert_, base_w8_, ED_LDS_BITS, ED_LDS_WINDOWS, ED_LDS_DIGITS, ED_LDS_ENTRY_WORDS, lds_convert_, var_fused_, 32, decompress_, compress_, nullptr, nullptr,
                              point_add_u_, to_affine_add_u_};
    t.ct_bits = ct_base_bits<ED25519U>();
    t.ct_windows = ct_base_windows<ED25519U>();
    t.ct_entries = ct_base_entries<ED25519U>();
    t.ct_entry_words = ct_entry_words<ED25519U>();
    t.ct_convert = ct_convert_;
    t.base_ct = base_ct_;
    t.ctg_bits = ct_base_bits<ED25519U, true>();
    t.ctg_windows = ct_base_windows<ED25519U, true>();
    t.ctg_entries = ct_base_entries<ED25519U, true>();
    t.base_ctg = base_ctg_;
    t.var_ct = var_ct_;
    t.var_ct_grid = var_ct_grid_;
    return t;
  }();
  return o;
}
}  // namespace eccx

// Host-side launch interface between the C ABI (eccx_api.cpp) and the per-curve
// kernel translation units (k_<curve>.hip, one per curve so they compile in parallel).
#pragma once
#ifndef ECCX_NORM_U8
#define ECCX_NORM_U8 16
#endif
#ifndef ECCX_NORM_UBIG
#define ECCX_NORM_UBIG 8
#endif
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace eccx {

struct CurveInfo {
  int fb;           // field bytes
  int sb;           // scalar bytes
  int limbs;        // 32-bit limbs per field element
  int table_words;  // words per comb-table entry (2L Weierstrass, 3L Edwards)
  int row_words;    // words per variable-base scratch row, mirror kernels (0: no scratch)
  int edwards;
  int row5_words;   // words per scratch row of the fast (Jacobian) kernel; 0: no fast path
  int jac_words;    // words per un-normalised result row (X, Y, Z)
};

struct CurveOps {
  CurveInfo info;
  // variable base; internal opts are the OPT_* bits of kernels.hpp
  hipError_t (*var)(int grid, hipStream_t s, size_t n, const uint8_t* scalars, const uint8_t* points, uint8_t* out,
                    uint8_t* flags, uint8_t* proj, uint32_t* scratch, uint32_t opts);
  hipError_t (*base)(int grid, hipStream_t s, size_t n, const uint8_t* scalars, const uint32_t* table, uint8_t* out,
                     uint8_t* flags, uint8_t* proj, uint32_t opts);
  // default path (may be null): windowed ladder on the unsaturated field into `jac` rows, then
  // the batched normalisation to_affine_var
  hipError_t (*var_fast)(int grid, hipStream_t s, size_t n, const uint8_t* scalars, const uint8_t* points,
                         uint32_t* jac, uint8_t* flags, uint32_t* scratch, uint32_t opts);
  // fixed-base variant with the comb table staged in LDS (may be null); picks its own grid
  hipError_t (*base_lds)(int cus, hipStream_t s, size_t n, const uint8_t* scalars, const uint32_t* table,
                         uint32_t* rows, uint8_t* flags);
  // batched normalisation of homogeneous rows (mirror / comb kernels run with OPT_OUT_ROWS)
  hipError_t (*to_affine_hom)(int grid, hipStream_t s, size_t n, const uint32_t* rows, uint8_t* out, uint8_t* flags);
  // persistent-grid sizes of the variable-base kernels: min(workgroups needed, CUs x resident
  // workgroups per CU from the occupancy query), so no workgroup waits for a slot
  int (*var_grid)(int cus, size_t n);
  int (*var_fast_grid)(int cus, size_t n);
  // normalisation of the rows var_fast / base_unsat / var_fused write (canonical plain integers:
  // Jacobian X, Y, Z for the Weierstrass curves, X, Y, Z for edwards25519); may be null
  hipError_t (*to_affine_var)(int grid, hipStream_t s, size_t n, const uint32_t* rows, uint8_t* out, uint8_t* flags);
  // batched group law a + b (or a - b) on affine inputs into un-normalised rows
  hipError_t (*point_add)(int grid, hipStream_t s, size_t n, const uint8_t* a, const uint8_t* a_inf, const uint8_t* b,
                          const uint8_t* b_inf, uint32_t* rows, uint8_t* flags, uint32_t opts);
  // fixed-base comb on the unsaturated field with wide windows (may be null): its table
  // (windows x 2^W entries of utable_words) is made by comb_convert from the affine bytes of
  // d * 2^(W*w) * G; rows go to to_affine_var
  int utable_words;
  int comb_bits;  // window width W of that table: ceil(8*SB / W) windows x 2^W entries
  hipError_t (*comb_convert)(hipStream_t s, size_t entries, const uint8_t* affine, uint32_t* utable);
  hipError_t (*base_unsat)(int grid, hipStream_t s, size_t n, const uint8_t* scalars, const uint32_t* utable,
                           uint32_t* rows, uint8_t* flags);
  // fused double-scalar u1*G + u2*Q (may be null): var_fast's ladder on (u2, q) followed by the
  // comb of u1 over utable, rows for to_affine_var; same grid and scratch as var_fast
  // table of the LDS-resident fixed-base variant (base_lds; may be 0 / null): lds_windows x lds_digits
  // entries of lds_entry_words, entry (w, d) = d * 2^(lds_bits * w) * G, made by lds_convert from
  // affine bytes
  int lds_bits, lds_windows, lds_digits, lds_entry_words;
  hipError_t (*lds_convert)(hipStream_t s, size_t entries, const uint8_t* affine, uint32_t* table);
  hipError_t (*var_fused)(int grid, hipStream_t s, size_t n, const uint8_t* u2, const uint8_t* q, uint32_t* rows,
                          uint8_t* flags, uint32_t* scratch, uint32_t opts, const uint8_t* u1, const uint32_t* utable);
  // wire formats (kernels_codec.hpp): enc_bytes per compressed point; decompress writes x||y and
  // flags 0 point / 1 infinity / 2 rejected; compress takes x||y and optional infinity flags
  int enc_bytes;
  hipError_t (*decompress)(int grid, hipStream_t s, size_t n, const uint8_t* enc, uint8_t* out, uint8_t* flags);
  hipError_t (*compress)(int grid, hipStream_t s, size_t n, const uint8_t* xy, const uint8_t* inf, uint8_t* out);
  // the zcash uncompressed flavour, 2 FB bytes per point (bls12_381_g1 only, else null)
  hipError_t (*decompress_raw)(int grid, hipStream_t s, size_t n, const uint8_t* enc, uint8_t* out, uint8_t* flags);
  hipError_t (*compress_raw)(int grid, hipStream_t s, size_t n, const uint8_t* xy, const uint8_t* inf, uint8_t* out);
  // group law on the unsaturated field (default; point_add / to_affine_hom are the saturated,
  // reference-mirroring pair) and the normalisation of its (X, Y, Z) rows
  hipError_t (*point_add_u)(int grid, hipStream_t s, size_t n, const uint8_t* a, const uint8_t* a_inf, const uint8_t* b,
                            const uint8_t* b_inf, uint32_t* rows, uint8_t* flags, uint32_t opts);
  hipError_t (*to_affine_add_u)(int grid, hipStream_t s, size_t n, const uint32_t* rows, uint8_t* out, uint8_t* flags);
  // Weierstrass curves (null / 0 for edwards25519), kernels_coz.hpp: the variable-base ladder over an
  // affine window table (mixed additions); glv != 0 selects the endomorphism form (bls12_381_g1) for bases known to be
  // in the prime-order subgroup.  Rows for to_affine_var; scratch rows of coz_row_words.  Units whose
  // base point has order <= 16 come back marked (flag 0xFE) and are redone by var_fast launched with
  // the only-marked option.  subgroup_check: the membership test applied in place to decompressed
  // points (flags 0 -> 2 and zero bytes for points outside the subgroup)
  hipError_t (*var_coz)(int grid, hipStream_t s, size_t n, const uint8_t* scalars, const uint8_t* points, uint32_t* rows,
                        uint8_t* flags, uint32_t* scratch, uint32_t opts, int glv);
  int (*var_coz_grid)(int cus, size_t n, int glv);
  int coz_row_words;
  // the verify shape over the same ladder (arguments as var_fused)
  hipError_t (*var_coz_fused)(int grid, hipStream_t s, size_t n, const uint8_t* u2, const uint8_t* q, uint32_t* rows, uint8_t* flags,
                              uint32_t* scratch, uint32_t opts, const uint8_t* u1, const uint32_t* utable);
  int (*var_coz_fused_grid)(int cus, size_t n);
  hipError_t (*subgroup_check)(int grid, hipStream_t s, size_t n, uint8_t* xy, uint8_t* flags);
  // Secret scalars (ECCX_CT_SCAN, kernels_ct.hpp / kernels_coz.hpp): no address and no branch depends on a
  // scalar digit.
  //   base_ct   fixed base over a table of ct_windows slices x ct_entries entries x ct_entry_words words, entry
  //             (w, d) = d * 2^(ct_bits * w) * G for d = 1 .. ct_entries, made by ct_convert from affine bytes;
  //             rows for to_affine_var
  //   var_ct    variable base, Weierstrass (null for edwards25519): the affine-table ladder with every table
  //             row read at every lookup; slab rows of coz_row_words; rows for to_affine_var; units marked
  //             0xFE (from the base point alone) are redone by `var` with the scan and only-marked options
  //   base_ctg  the same with the lookup as a cross-lane gather (ECCX_CT_GATHER): its own table, ctg_bits-wide windows
  int ct_bits, ct_windows, ct_entries, ct_entry_words;
  int ctg_bits, ctg_windows, ctg_entries;
  hipError_t (*ct_convert)(hipStream_t s, size_t entries, const uint8_t* affine, uint32_t* table);
  hipError_t (*base_ct)(int grid, hipStream_t s, size_t n, const uint8_t* scalars, const uint32_t* table, uint32_t* rows,
                        uint8_t* flags);
  hipError_t (*base_ctg)(int grid, hipStream_t s, size_t n, const uint8_t* scalars, const uint32_t* table, uint32_t* rows,
                         uint8_t* flags);
  hipError_t (*var_ct)(int grid, hipStream_t s, size_t n, const uint8_t* scalars, const uint8_t* points, uint32_t* rows,
                       uint8_t* flags, uint32_t* scratch, uint32_t opts);
  int (*var_ct_grid)(int cus, size_t n);
  // var_ct for bases the caller vouches to be in the prime-order subgroup (ECCX_CT_SCAN | ECCX_ASSUME_SUBGROUP): the
  // accumulator == +-entry selects only in the windows a prime-order point can reach; null where the curve has no cofactor
  hipError_t (*var_ct_prime)(int grid, hipStream_t s, size_t n, const uint8_t* scalars, const uint8_t* points, uint32_t* rows,
                             uint8_t* flags, uint32_t* scratch, uint32_t opts);
  int (*var_ct_prime_grid)(int cus, size_t n);
  // normalisation of Jacobian rows to the x-coordinate alone, FB bytes per unit (ECCX_OUT_X_ONLY; Weierstrass)
  hipError_t (*to_affine_x)(int grid, hipStream_t s, size_t n, const uint32_t* rows, uint8_t* out, uint8_t* flags);
};
// units normalised per lane with one inversion: 16 where the prefix products fit the register
// file (8-limb fields), 8 above
constexpr int to_affine_u(int limbs) { return limbs <= 8 ? ECCX_NORM_U8 : ECCX_NORM_UBIG; }
// the same for the rows of the group law, whose producer is short: the normalisation is most of
// the operation there
#ifndef ECCX_NORM_ADD_U8
#define ECCX_NORM_ADD_U8 ECCX_NORM_U8
#endif
#ifndef ECCX_NORM_ADD_UBIG
#define ECCX_NORM_ADD_UBIG ECCX_NORM_UBIG
#endif
constexpr int to_affine_add_u(int limbs) { return limbs <= 8 ? ECCX_NORM_ADD_U8 : ECCX_NORM_ADD_UBIG; }

const CurveOps& ops_P256();
const CurveOps& ops_P384();
const CurveOps& ops_P521();
const CurveOps& ops_BLS12_381();
const CurveOps& ops_ED25519();

// curve25519 x-only ladder (k_ed25519.hip): rows of row_words<8>() = 24 words per unit
hipError_t launch_x25519_ladder(int grid, hipStream_t s, size_t n, const uint8_t* scalars, const uint8_t* u,
                                uint32_t* rows, uint8_t* flags, uint32_t opts);
hipError_t launch_x25519_to_u(int grid, hipStream_t s, size_t n, const uint32_t* rows, uint8_t* out, uint8_t* flags);

constexpr int LAUNCH_WG = 256;

template <class K>
inline int occupancy_per_cu(K kernel) {
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, LAUNCH_WG, 0) != hipSuccess || per_cu < 1) per_cu = 1;
  return per_cu;
}
inline int persistent_grid(int per_cu, int cus, size_t n) {
  size_t need = (n + LAUNCH_WG - 1) / LAUNCH_WG;
  size_t cap = (size_t)cus * (size_t)per_cu;
  size_t g = need < cap ? need : cap;
  return (int)(g ? g : 1);
}

}  // namespace eccx

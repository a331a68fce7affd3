/* eccx.h -- C ABI of the MI355X batched scalar-multiplication engine.
 *
 * Drop-in boundary for eccoxide's scalar-multiplication hot path.  The reference
 * (vincenthz/eccoxide, Rust) has no FFI or plugin layer: the seam is the trait +
 * inherent methods listed below, one point and one scalar per call.  Each entry
 * point here is the batched, C-callable form of one of them; INTEGRATION.md shows
 * the Rust `extern "C"` binding a maintainer would add.
 *
 *   eccx_scalarmul_var[_dev]   impl Mul<&Scalar> for &Point      src/curve/fiat/curve_macros.rs:321-327
 *                              -> Point::scale                   curve_macros.rs:47-49 (a=-3), :103-105 (a=0)
 *                              -> projective::Point::scale_{am3,a0}_ct   src/curve/projective.rs:905-918
 *                              edwards25519: Point::scale        src/curve/curve25519.rs:746-762
 *   eccx_scalarmul_base[_dev]  Point::mul_base / CurveGroup::mul_base
 *                              curve_macros.rs:55-63, :111-119; src/curve/group.rs:28-70
 *                              -> mul_base_table_{am3,a0}        projective.rs:945-981
 *                              edwards25519: Point::mul_base     curve25519.rs:840-851
 *   output normalisation       Point::to_affine / to_affine_ct   curve_macros.rs:247-268,
 *                              projective.rs:655-682, curve25519.rs:663-666
 *   eccx_comb_table            COMB_TABLE constants              src/params/comb/<curve>.rs
 *   eccx_point_add[_dev]       impl Add / Sub / Neg, CurveGroup::double   curve_macros.rs:297-411, group.rs:28-70
 *   eccx_double_scalarmul[_dev]  u1*G + u2*Q                     src/protocol/ecdsa.rs:215, ed25519.rs:145
 *   eccx_x25519[_dev]          MontgomeryPoint ladder / x25519   curve25519.rs:474-541, src/protocol/x25519.rs:14-51
 *   eccx_point_compress[_dev]  PointAffine::compress, to_compressed, to_uncompressed, encode_point
 *   eccx_point_decompress[_dev]  PointAffine::decompress, from_compressed[_oncurve_only],
 *                              from_uncompressed[_oncurve_only], decode_point
 *                              curve_macros.rs:211-223, src/curve/affine.rs:23-58,
 *                              src/curve/bls12_381/serialize.rs:253-383, src/protocol/ed25519.rs:27-59
 *
 * Byte conventions are the reference's:
 *   - Weierstrass curves (p256r1, p384r1, p521r1, BLS12-381 G1): field elements and
 *     scalars are big-endian, FB / SB bytes (field_macros.rs:6-29).
 *   - edwards25519: field elements little-endian (curve25519.rs:138); the SCALAR is the
 *     32-byte BIG-endian string the reference's loops index (Scalar::to_bytes_be,
 *     curve25519.rs:761, :842).
 *   - a point is its affine pair x||y, 2*FB bytes.  The Weierstrass point at infinity has
 *     no affine form (to_affine -> None, projective.rs:666-668): it is reported through
 *     the flag array with x = y = 0 bytes.
 *   - scalars are used as given: the ladder multiplies by the integer the SB bytes
 *     encode (projective.rs:871-896 accepts any byte string).  For P in the prime-order
 *     subgroup that equals (k mod n)*P.
 *
 * All functions return 0 on success or a negative ECCX_ERR_* code; none aborts.
 *
 * SIDE CHANNELS -- read before using this for secret scalars.  The reference's Point * Scalar,
 * mul_base and to_affine_ct are constant-time: select_from_table reads every table entry
 * (projective.rs:427-434, curve25519.rs:862-869) and nothing branches on the scalar.  The DEFAULT
 * kernels here (and ECCX_TABLE_IN_LDS / ECCX_TABLE_IN_L2) are NOT: they read their window and comb
 * tables at addresses chosen by scalar digits and take wave-uniform branches on data-dependent
 * ballots (accumulator at infinity, equal points).  They return the same bytes, and are meant for
 * PUBLIC scalars (signature verification, public-key checks) or callers who accept that.  For
 * secret scalars (key generation, signing, ECDH) pass ECCX_CT_SCAN.  What that option guarantees is a
 * STRUCTURE, checked on the compiled code (tools/isa_histogram.py --branches, profiles/r03_isa_ct_*.txt):
 *   - no memory address depends on a scalar digit: every lookup reads every entry of its table (fixed base:
 *     the window's slice, staged in LDS by the workgroup; variable base: all rows of the lane's own table)
 *     and keeps one with a per-lane select executed by every lane for every entry (v_cndmask_b32, or for the
 *     fixed-base slices v_pk_fma_f32 against 1.0 / 0.0 on words that are exact under it: kernels_ct.hpp
 *     ct_scan_lds_pk); the scalar bytes are read at addresses that depend on the window number;
 *   - no branch depends on scalar-derived data: signs, digit 0, accumulator at infinity and accumulator ==
 *     +-entry are resolved by selects (written as inline assembly: the compiler otherwise turns ?: into EXEC-
 *     masked regions it can skip); the conditional branches left are loop counters and batch bounds;
 *   - the batched normalisation substitutes Z = 0 and zeroes outputs by selects as well.
 * What still depends on data under ECCX_CT_SCAN: the BASE POINT -- rejected inputs (ECCX_VALIDATE_POINTS), a base
 * of order <= 8 (bls12_381_g1 cofactor points; never on a prime-order curve) or bytes that are no curve point
 * mark the unit, from the point alone, and it is redone by the reference-mirroring scan kernel; which units those
 * are is visible in timing.  ECCX_CT_GATHER (opt-in, fixed base) replaces the scan by a cross-lane register
 * gather; see the option.  eccx_x25519 is uniform by construction (conditional swaps are selects, no table).
 * What was MEASURED (tools/ct_trace_check.py, profiles/r03_ct_instruction_counts.json): six scalar sets -- random,
 * all zero, all ones, n - 1, one scalar repeated, random again -- execute exactly the same number of vector, scalar,
 * memory and LDS instructions in every secret-scalar kernel and the same busy cycles to 1-3 %.  Their DURATIONS are
 * not equal: a batch whose lanes all hold the same scalar runs 6-15 % shorter than a batch of random scalars,
 * because the chip's power management holds a higher clock when the lanes compute on equal data (2.38-2.40 GHz
 * against 2.25-2.29 for the p256r1 comb) -- the frequency side channel every DVFS processor has, the reference's
 * CPUs included; two batches of independent random scalars are indistinguishable at the precision of that
 * measurement.  GPU schedulers and caches are not modelled beyond these measurements.
 *
 * MEMORY AND BLOCKING.  A context is bound to one GPU and owns
 *   - the window-table slab of the variable-base ladders: resident lanes x 17 rows (P-256:
 *     0.86 GB, P-384 / BLS12-381: 0.62 GB, P-521: 0.79 GB; sized by the largest batch seen),
 *   - a buffer of un-normalised result rows, 112-224 bytes per unit of the largest batch seen,
 *   - the fixed-base tables of each curve used: the 16-bit-window table (134 MB for p256r1,
 *     ed25519 and bls12_381_g1, 201 MB p384r1, 415 MB p521r1), the reference-layout comb
 *     (64-265 KB), for ECCX_TABLE_IN_LDS a 155 KB image, for ECCX_CT_SCAN / ECCX_CT_GATHER a signed-window
 *     table each (0.1-0.5 MB),
 *   - the device-side copies the HOST-buffer entry points keep of their arguments (sized by the largest batch
 *     seen, or by eccx_reserve with ECCX_PREP_HOST): those calls allocate and free nothing once warm.
 * eccx_device_bytes() reports the total.  The buffers grow on demand: a call with a batch larger
 * than any before frees and reallocates them after a device-wide synchronisation, and the first
 * fixed-base / double-scalar call per curve builds the tables (64-271 ms) after waiting for the
 * caller's stream.  eccx_prepare() and eccx_reserve() pay both up front; after them the _dev entry
 * points never block, allocate or free.
 * Calls on ONE context must not overlap in time -- enqueue them on one stream, or serialise
 * them; use one context per stream / host thread for concurrency (table construction is guarded
 * by a mutex; the tables are read-only afterwards; eccx_last_error() hands each calling thread
 * its own copy of the message).
 */
#ifndef ECCX_H
#define ECCX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  ECCX_P256R1 = 0,       /* src/curve/sec2/p256r1.rs */
  ECCX_P384R1 = 1,       /* src/curve/sec2/p384r1.rs */
  ECCX_P521R1 = 2,       /* src/curve/sec2/p521r1.rs */
  ECCX_BLS12_381_G1 = 3, /* src/curve/bls12_381/g1.rs */
  ECCX_ED25519 = 4       /* src/curve/curve25519.rs (twisted Edwards form) */
} eccx_curve;

enum {
  ECCX_OK = 0,
  ECCX_ERR_CURVE = -1, /* unknown curve id */
  ECCX_ERR_ARG = -2,   /* null pointer / bad size */
  ECCX_ERR_HIP = -3,   /* a HIP call failed; see eccx_last_error() */
  ECCX_ERR_NOMEM = -4
};

/* option bits */
enum {
  ECCX_VALIDATE_POINTS = 1u << 0, /* reject input points that are non-canonical or off the curve
                                     (PointAffine::from_coordinate, src/curve/affine.rs:90-119):
                                     flag 2, zero output */
  ECCX_MIRROR_REFERENCE = 1u << 1 /* run the kernels that follow the reference operation for
                                     operation (RCB complete formulas in homogeneous coordinates,
                                     src/curve/projective.rs:340-423,586-646).  Implied when `proj`
                                     is requested.  The default kernels use Jacobian coordinates:
                                     same affine bytes and flags for every on-curve input, about
                                     1.5x faster.  Off-curve inputs (only reachable without
                                     ECCX_VALIDATE_POINTS) give unspecified output by default and
                                     the reference's arithmetic under this option. */
  ,
  ECCX_TABLE_IN_LDS = 1u << 2,   /* fixed base, edwards25519 only: keep the comb table in LDS --
                                     signed 6-bit windows, the widest table 160 KiB can hold (43
                                     additions), one 1024-thread workgroup per CU.  The default
                                     fixed-base path uses 16-bit windows (16 additions) over a
                                     134 MB table in HBM: same results, about 2x faster; see
                                     DESIGN.md for the numbers. */
  ECCX_TABLE_IN_L2 = 1u << 3,    /* fixed base: the reference's 4-bit comb, table read through L1/L2 */
  ECCX_X25519_RAW_LADDER = 1u << 4, /* eccx_x25519: raw MontgomeryPoint::scale_bytes semantics */
  ECCX_SUBTRACT = 1u << 5,         /* eccx_point_add: compute a - b */
  ECCX_CHECK_SUBGROUP = 1u << 6,   /* eccx_point_decompress, bls12_381_g1: reject points outside G1 */
  ECCX_UNCOMPRESSED = 1u << 7,     /* eccx_point_[de]compress, bls12_381_g1: the 96-byte zcash flavour */
  ECCX_CT_SCAN = 1u << 8,          /* eccx_scalarmul_var / _base: secret scalars (see SIDE CHANNELS above).  Fixed base: signed
                                      6-bit windows (edwards25519: 5), every entry of the window read by every lane, XYZZ mixed
                                      additions with select-only special cases / complete Edwards additions.  Variable base,
                                      Weierstrass: the affine-table ladder with signed 4-bit windows, all 8 rows of the lane's
                                      table read at every lookup; edwards25519: signed 3-bit windows over 4 rows normalised to
                                      Z = 1, complete additions.  With
                                      ECCX_MIRROR_REFERENCE (or proj): the reference-mirroring kernels with select_from_table's
                                      scan (src/curve/projective.rs:427-434, curve25519.rs:862-869).  Same bytes out.
                                      Not accepted by eccx_double_scalarmul (public data), nor with ECCX_TABLE_IN_LDS.
                                      With ECCX_ASSUME_SUBGROUP see there. */
  ECCX_CT_GATHER = 1u << 10,       /* with ECCX_CT_SCAN, eccx_scalarmul_base: look the window's entry up by a cross-lane
                                      gather (ds_bpermute_b32 from the lane that holds the entry) instead of the scan of the
                                      whole window.  Still no memory address and no branch that depends on a digit; the
                                      digit steers the wavefront's register crossbar, whose time was MEASURED the same for
                                      every index pattern tried (profiles/r03_select_rates.jsonl) -- not an architectural
                                      guarantee, hence opt-in.  About 1.4x faster than the scan. */
  ECCX_OUT_X_ONLY = 1u << 11,      /* eccx_double_scalarmul: write the x-coordinate alone, FB bytes per unit (`out` is then
                                      n x FB): Point::to_affine_x_ct (src/curve/projective.rs:690), which is all ECDSA
                                      verification reads (src/protocol/ecdsa.rs:383).  Weierstrass curves. */
  ECCX_ASSUME_SUBGROUP = 1u << 9   /* eccx_scalarmul_var, bls12_381_g1: the caller guarantees every base point
                                      is in the prime-order subgroup G1 (e.g. it was decoded under
                                      ECCX_CHECK_SUBGROUP, or is a multiple of the generator).  The
                                      ladder then splits k = k1 + k2*x^2 and uses the endomorphism
                                      sigma(P) = [-x^2]P (src/curve/bls12_381/g1.rs:90-109): half the
                                      doublings.  For a point outside G1 the result is NOT k*P.
                                      With ECCX_CT_SCAN (secret scalar, base in G1: sk * H(m)) the same split runs in
                                      secret-scalar form: branch-free split, every table row read at both lookups of a
                                      window, "accumulator == +-table entry" resolved by selects in the one window
                                      where a base of prime order can reach it (1.6x faster than ECCX_CT_SCAN alone).
                                      Other curves: no effect. */
};

/* eccx_prepare / eccx_reserve: which one-time costs to pay now */
enum {
  ECCX_PREP_VAR = 1u << 0,      /* variable base and double-scalar, default kernels: window-table slab */
  ECCX_PREP_BASE = 1u << 1,     /* fixed base and double-scalar: the comb tables of the curve */
  ECCX_PREP_BASE_LDS = 1u << 2, /* ECCX_TABLE_IN_LDS image (edwards25519) */
  ECCX_PREP_MIRROR = 1u << 3,   /* ECCX_MIRROR_REFERENCE / proj: slab of the mirror ladder */
  ECCX_PREP_HOST = 1u << 6,     /* eccx_reserve: the device-side copies the HOST-buffer entry points keep of their arguments */
  ECCX_PREP_CT_GATHER = 1u << 5, /* ECCX_CT_SCAN | ECCX_CT_GATHER: eccx_prepare builds that form's table */
  ECCX_PREP_CT = 1u << 4        /* ECCX_CT_SCAN: eccx_prepare builds the signed-window table of the secret-scalar
                                   fixed-base kernel (99-460 KB); eccx_reserve sizes the slabs of the scanning
                                   variable-base ladder and of its fix-up pass */
};

/* flag values written per unit */
enum { ECCX_FLAG_FINITE = 0, ECCX_FLAG_INFINITY = 1, ECCX_FLAG_REJECTED = 2 };

typedef struct eccx_ctx eccx_ctx;

/* sizes: field bytes FB, scalar bytes SB, comb windows NW = 2*SB; <0 on bad curve */
int eccx_field_bytes(int curve);
int eccx_scalar_bytes(int curve);

/* Create / destroy a context on HIP device `device`. */
int eccx_init(int device, eccx_ctx** out_ctx);
void eccx_shutdown(eccx_ctx* ctx);
/* the message of the last failing call on ctx; the pointer is a per-thread copy, valid until the
 * calling thread's next eccx_last_error */
const char* eccx_last_error(const eccx_ctx* ctx);
const char* eccx_strerror(int code);

/* Pay the one-time costs of later calls now (both block until done):
 *   eccx_prepare  builds the fixed-base tables `what` names for `curve`;
 *   eccx_reserve  sizes the scratch slab and the row buffer for batches of up to max_n units of
 *                 `curve` through the entry points `what` names (buffers only ever grow; reserve
 *                 every curve you will use, the largest footprint wins).
 * After both, a _dev call with n <= max_n returns without synchronising, allocating or freeing.
 * eccx_device_bytes: device memory the context currently owns. */
int eccx_prepare(eccx_ctx* ctx, int curve, uint32_t what);
int eccx_reserve(eccx_ctx* ctx, int curve, size_t max_n, uint32_t what);
size_t eccx_device_bytes(const eccx_ctx* ctx);

/* Variable base: out[i] = scalars[i] * points[i].
 *   scalars : n x SB          points : n x 2FB (affine x||y)
 *   out     : n x 2FB         flags  : n bytes (ECCX_FLAG_*)
 *   proj    : NULL, or n x 3FB (edwards25519: n x 4FB) receiving the reference's
 *             un-normalised result coordinates X||Y||Z(||T), canonical bytes
 * Host-pointer form: copies in, runs, copies out, synchronises. */
int eccx_scalarmul_var(eccx_ctx* ctx, int curve, size_t n, const uint8_t* scalars, const uint8_t* points,
                       uint8_t* out, uint8_t* flags, uint8_t* proj, uint32_t opts);

/* Fixed base: out[i] = scalars[i] * G (comb table, built once per context and curve). */
int eccx_scalarmul_base(eccx_ctx* ctx, int curve, size_t n, const uint8_t* scalars, uint8_t* out,
                        uint8_t* flags, uint8_t* proj, uint32_t opts);

/* Device-pointer forms: every buffer is device memory of ctx's GPU, the work is
 * enqueued on `stream` (a hipStream_t; NULL = HIP's default stream) and the call
 * returns without synchronising -- except for the one-time costs listed under MEMORY AND
 * BLOCKING above (first table build per curve: waits for `stream`, builds, blocks; a batch
 * larger than any before: device-wide synchronisation + reallocation), which eccx_prepare /
 * eccx_reserve move out of the way. */
int eccx_scalarmul_var_dev(eccx_ctx* ctx, int curve, size_t n, const void* d_scalars, const void* d_points,
                           void* d_out, void* d_flags, void* d_proj, uint32_t opts, void* stream);
int eccx_scalarmul_base_dev(eccx_ctx* ctx, int curve, size_t n, const void* d_scalars, void* d_out,
                            void* d_flags, void* d_proj, uint32_t opts, void* stream);

/* Group law on batches: out[i] = a[i] + b[i], or a[i] - b[i] with ECCX_SUBTRACT.
 * Mirrors impl Add / Sub / Neg for Point and CurveGroup::double
 * (src/curve/fiat/curve_macros.rs:297-411, src/curve/group.rs:28-70): the complete addition
 * (projective.rs:340-423 / :268-338; curve25519.rs:695-710) covers a == b, a == -b and the
 * point at infinity, so double(a) is eccx_point_add(a, a) and neg(a) is infinity - a.
 *   a, b        : n x 2FB affine x||y (host memory)
 *   a_inf, b_inf: NULL, or n flag bytes (1 = that operand is the point at infinity; Weierstrass only)
 *   out, flags  : as for eccx_scalarmul_var
 * Default kernels: the complete addition on the unsaturated field; ECCX_MIRROR_REFERENCE selects
 * the saturated-limb pair (same formulas, same bytes out). */
int eccx_point_add(eccx_ctx* ctx, int curve, size_t n, const uint8_t* a, const uint8_t* a_inf, const uint8_t* b,
                   const uint8_t* b_inf, uint8_t* out, uint8_t* flags, uint32_t opts);
/* Device-buffer form (d_a_inf / d_b_inf may be NULL), enqueued on `stream` without synchronising. */
int eccx_point_add_dev(eccx_ctx* ctx, int curve, size_t n, const void* d_a, const void* d_a_inf, const void* d_b,
                       const void* d_b_inf, void* d_out, void* d_flags, uint32_t opts, void* stream);

/* Double-scalar "verify shape": out[i] = u1[i]*G + u2[i]*Q[i]  (u1*G - u2*Q with ECCX_SUBTRACT).
 * The batched form of ECDSA verification's u1*G + u2*Q (src/protocol/ecdsa.rs:215) and of
 * Ed25519's [s]B - [k]A (src/protocol/ed25519.rs:145; the reference uses the variable-time
 * double_scalar_mul_base_vartime, src/curve/curve25519.rs:1157-1183 -- same point).  ONE kernel
 * per curve: Weierstrass, the variable-base ladder for u2*Q followed by the 16-bit-window comb of u1*G
 * accumulated onto the same Jacobian point, one normalisation; edwards25519 likewise with the
 * complete extended-coordinate additions.
 *   u1, u2 : n x SB scalars      q : n x 2FB affine points      out, flags: as above
 * ECCX_VALIDATE_POINTS applies to q.  ECCX_OUT_X_ONLY: out is n x FB, the x-coordinates alone. */
int eccx_double_scalarmul(eccx_ctx* ctx, int curve, size_t n, const uint8_t* u1, const uint8_t* u2,
                          const uint8_t* q, uint8_t* out, uint8_t* flags, uint32_t opts);
/* Device-buffer form: inputs and outputs already in this device's memory, work enqueued on
 * `stream` (NULL = HIP's default stream), no synchronisation -- as eccx_scalarmul_var_dev. */
int eccx_double_scalarmul_dev(eccx_ctx* ctx, int curve, size_t n, const void* d_u1, const void* d_u2, const void* d_q,
                              void* d_out, void* d_flags, uint32_t opts, void* stream);

/* X25519: the curve25519 x-only Montgomery ladder.
 *   default            protocol::x25519::x25519 (src/protocol/x25519.rs:36-45): `scalars` are
 *                      n x 32 little-endian RFC 7748 scalars, clamped on use (x25519.rs:15-20);
 *                      the top bit of each u-coordinate is masked (decode_u, :24-29).
 *   ECCX_X25519_RAW_LADDER  MontgomeryPoint::scale_bytes (src/curve/curve25519.rs:535-541 ->
 *                      ladder :474-513): `scalars` are the n x 32 BIG-endian strings the ladder
 *                      consumes, no clamping; u is reduced mod p as given.
 *   u    : n x 32 little-endian u-coordinates, or NULL for the base point u = 9
 *          (x25519_base, x25519.rs:49-51)
 *   out  : n x 32 little-endian u-coordinates of the results
 *   flags: 1 where the result is zero (point at infinity / low-order input: callers doing
 *          Diffie-Hellman must reject it, x25519.rs:33-35), else 0 */
int eccx_x25519(eccx_ctx* ctx, size_t n, const uint8_t* scalars, const uint8_t* u, uint8_t* out, uint8_t* flags,
                uint32_t opts);
int eccx_x25519_dev(eccx_ctx* ctx, size_t n, const void* d_scalars, const void* d_u, void* d_out, void* d_flags,
                    uint32_t opts, void* stream);

/* Point wire formats, batched: compressed encodings <-> the affine x||y records above.
 *   p256r1 / p384r1 / p521r1  SEC1 compressed, FB + 1 bytes: 0x02 | (y odd), then x big-endian;
 *                             a record of FB + 1 zero bytes stands for the point at infinity.
 *                             Byte form of PointAffine::compress / decompress, which trade in
 *                             (x, Sign) with Sign::Negative = y odd (src/curve/affine.rs:23-58,
 *                             src/curve/fiat/curve_macros.rs:211-223, field_macros.rs:557-565).
 *   bls12_381_g1              zcash compressed, 48 bytes (src/curve/bls12_381/serialize.rs:
 *                             253-262 to_compressed, :321-335 from_compressed_oncurve_only; with
 *                             ECCX_CHECK_SUBGROUP :299-313 from_compressed, whose membership
 *                             test [r]P = infinity runs through the variable-base kernel).
 *   edwards25519              RFC 8032, 32 bytes: y little-endian, low bit of x in bit 255
 *                             (src/protocol/ed25519.rs:27-59 encode_point / decode_point).
 * eccx_point_decompress: enc n x eccx_compressed_bytes(curve) -> out n x 2FB, flags n:
 *   0 point, 1 the encoding of the point at infinity (Weierstrass), 2 rejected -- bad prefix or
 *   flag bits, coordinate not below p, no point with this coordinate, (ed25519) x = 0 with the
 *   sign bit set, (ECCX_CHECK_SUBGROUP) not in G1.  Records flagged 1 or 2 are zero-filled.
 * eccx_point_compress: xy n x 2FB canonical coordinates, inf NULL or n bytes (non-zero = point
 *   at infinity, as the flags of the scalar multiplications report it) -> out n x
 *   eccx_compressed_bytes(curve).
 * ECCX_UNCOMPRESSED (bls12_381_g1 only, ECCX_ERR_ARG elsewhere) selects the zcash uncompressed
 *   flavour, 2FB = 96 bytes per point: x||y with the flag bits in the leading byte -- compression
 *   and sort bits clear, bit 6 alone for the point at infinity (to_uncompressed /
 *   from_uncompressed[_oncurve_only], serialize.rs:269-279,353-383); decoding checks both
 *   coordinates are below p and satisfy the curve equation.
 * The _dev forms take device memory and enqueue on `stream` without synchronising (except under
 * ECCX_CHECK_SUBGROUP, whose temporaries are released after a stream synchronisation), so
 * decompress -> scalarmul -> compress chains stay on the GPU. */
int eccx_compressed_bytes(int curve);
int eccx_point_decompress(eccx_ctx* ctx, int curve, size_t n, const uint8_t* enc, uint8_t* out, uint8_t* flags,
                          uint32_t opts);
int eccx_point_compress(eccx_ctx* ctx, int curve, size_t n, const uint8_t* xy, const uint8_t* inf, uint8_t* out,
                        uint32_t opts);
int eccx_point_decompress_dev(eccx_ctx* ctx, int curve, size_t n, const void* d_enc, void* d_out, void* d_flags,
                              uint32_t opts, void* stream);
int eccx_point_compress_dev(eccx_ctx* ctx, int curve, size_t n, const void* d_xy, const void* d_inf, void* d_out,
                            uint32_t opts, void* stream);

/* The fixed-base comb table in the reference's on-disk layout (src/params/comb/<curve>.rs):
 * NW x 15 entries (j+1)*16^i*G as x||y, FB bytes each, big-endian (little-endian for
 * edwards25519).  `out` is host memory of NW*15*2*FB bytes. */
int eccx_comb_table(eccx_ctx* ctx, int curve, uint8_t* out);

/* Split a host batch into contiguous shards over several contexts (one per GPU),
 * run them concurrently and gather into the caller's host buffers. */
int eccx_scalarmul_var_sharded(eccx_ctx** ctxs, int nctx, int curve, size_t n, const uint8_t* scalars,
                               const uint8_t* points, uint8_t* out, uint8_t* flags, uint32_t opts);
int eccx_scalarmul_base_sharded(eccx_ctx** ctxs, int nctx, int curve, size_t n, const uint8_t* scalars,
                                uint8_t* out, uint8_t* flags, uint32_t opts);

#ifdef __cplusplus
}
#endif
#endif /* ECCX_H */

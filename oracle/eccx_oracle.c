/* CPU ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C restatement of eccoxide's scalar-multiplication hot path
 * (reference: vincenthz/eccoxide, Rust).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; eccoxide_amd/ never
 * links or calls it.
 *
 * Follows, step for step:
 *   field      src/curve/fiat/{p256,p384,bls12_381}_64.rs (word-by-word Montgomery,
 *              canonical outputs) -- fe_mont.inc; p521/25519 by value only
 *   points     src/curve/projective.rs:268-423,544-646 (RCB complete formulas)
 *   ladders    src/curve/projective.rs:842-896 (fixed 4-bit window), :945-981 (comb)
 *   glue       src/curve/fiat/curve_macros.rs:47-63,103-119 (scale / mul_base)
 *   Edwards    src/curve/curve25519.rs:592-645,695-710 (extended add),
 *              :604-619,669-677 (double), :746-757 (scale_bytes), :840-902 (comb mul_base)
 *   affine     src/curve/projective.rs:655-682, curve25519.rs:663-666
 *
 * Parity pinning: tests/test_oracle_golden.py runs this library against every
 * golden vector the reference's tests hold for the path (the JSON files under
 * tests/golden: NIST kG, RFC 6979, RFC 8032, RFC 7748, BLS G1 KATs, comb tables) and against the
 * independent Python big-int oracle (oracle/ecc_ref.py).
 *
 * Build: make -C oracle   (gcc -O3 -march=native -shared -fPIC -pthread)
 */
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)

/* ---- field / curve instantiations ---------------------------------------- */
#define FE_L 4
#define FE_(n) CAT(fe4_, n)
#define CW_(n) CAT(cw4_, n)
#include "fe_mont.inc"
#include "curve_w.inc"
#undef FE_L
#undef FE_
#undef CW_

#define FE_L 6
#define FE_(n) CAT(fe6_, n)
#define CW_(n) CAT(cw6_, n)
#include "fe_mont.inc"
#include "curve_w.inc"
#undef FE_L
#undef FE_
#undef CW_

#define FE_L 9
#define FE_(n) CAT(fe9_, n)
#define CW_(n) CAT(cw9_, n)
#include "fe_mont.inc"
#include "curve_w.inc"
#undef FE_L
#undef FE_
#undef CW_

enum { ECCX_P256R1 = 0, ECCX_P384R1 = 1, ECCX_P521R1 = 2, ECCX_BLS12_381_G1 = 3, ECCX_ED25519 = 4, ECCX_NCURVES = 5 };

/* ---- standard curve constants (big-endian hex) -------------------------- */
typedef struct { const char *p, *b, *b3, *gx, *gy; int fb, sb, a0; } wparams;
static const wparams WP[4] = {
  { "ffffffff00000001000000000000000000000000ffffffffffffffffffffffff",
    "5ac635d8aa3a93e7b3ebbd55769886bc651d06b0cc53b0f63bce3c3e27d2604b",
    "1052a18afeafbbb61bc3380063c994352f57141164fb12e2b36ab4ba777720e2",
    "6b17d1f2e12c4247f8bce6e563a440f277037d812deb33a0f4a13945d898c296",
    "4fe342e2fe1a7f9b8ee7eb4a7c0f9e162bce33576b315ececbb6406837bf51f5", 32, 32, 0 },
  { "fffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffeffffffff0000000000000000ffffffff",
    "b3312fa7e23ee7e4988e056be3f82d19181d9c6efe8141120314088f5013875ac656398d8a2ed19d2a85c8edd3ec2aef",
    "", /* b3 unused for a = -3 */
    "aa87ca22be8b05378eb1c71ef320ad746e1d3b628ba79b9859f741e082542a385502f25dbf55296c3a545e3872760ab7",
    "3617de4a96262c6f5d9e98bf9292dc29f8f41dbd289a147ce9da3113b5f0b8c00a60b1ce1d7e819d7a431d7c90ea0e5f", 48, 48, 0 },
  { "01ffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffff",
    "0051953eb9618e1c9a1f929a21a0b68540eea2da725b99b315f3b8b489918ef109e156193951ec7e937b1652c0bd3bb1bf073573df883d2c34f1ef451fd46b503f00",
    "",
    "00c6858e06b70404e9cd9e3ecb662395b4429c648139053fb521f828af606b4d3dbaa14b5e77efe75928fe1dc127a2ffa8de3348b3c1856a429bf97e7e31c2e5bd66",
    "011839296a789a3bc0045c8a5fb42c7d1bd998f54449579b446817afbd17273e662c97ee72995ef42640c550b9013fad0761353c7086a272c24088be94769fd16650", 66, 66, 0 },
  { "1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab",
    "000000000000000000000000000000000000000000000000000000000000000000000000000000000000000000000004",
    "00000000000000000000000000000000000000000000000000000000000000000000000000000000000000000000000c",
    "17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb",
    "08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1", 48, 32, 1 },
};
/* edwards25519 (big-endian hex): p, d, 2d, Bx, By  (curve25519.rs:391-417) */
static const char* ED_P = "7fffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffed";
static const char* ED_D = "52036cee2b6ffe738cc740797779e89800700a4d4141d8ab75eb4dca135978a3";
static const char* ED_D2 = "2406d9dc56dffce7198e80f2eef3d13000e0149a8283b156ebd69b9426b2f159";
static const char* ED_GX = "216936d3cd6e53fec0a4e231fdd6dc5c692cc7609525a7b2c9562d608f25d51a";
static const char* ED_GY = "6666666666666666666666666666666666666666666666666666666666666658";

static void hex2bytes(uint8_t* out, const char* hex, int n) {
  for (int i = 0; i < n; i++) {
    unsigned v = 0;
    for (int k = 0; k < 2; k++) {
      char ch = hex[2 * i + k];
      v = v * 16 + (unsigned)(ch <= '9' ? ch - '0' : ch - 'a' + 10);
    }
    out[i] = (uint8_t)v;
  }
}

/* ---- Edwards (4-limb field) ------------------------------------------------ */
typedef struct { fe4_t x, y, z, t; } ed_pt;
typedef struct { fe4_field F; fe4_t d2, gx, gy; } ed_curve;

static void ed_identity(const ed_curve* C, ed_pt* p) {
  fe4_set_zero(&p->x); fe4_set_one(&C->F, &p->y); fe4_set_one(&C->F, &p->z); fe4_set_zero(&p->t);
}
static void ed_from_affine(const ed_curve* C, ed_pt* p, const fe4_t* x, const fe4_t* y) {
  p->x = *x; p->y = *y; fe4_set_one(&C->F, &p->z); fe4_mul(&C->F, &p->t, x, y);
}
/* curve25519.rs:695-710 */
static void ed_add(const ed_curve* C, ed_pt* r, const ed_pt* p, const ed_pt* q) {
  const fe4_field* F = &C->F;
  fe4_t aa, bb, cc, dd, e, f, g, h, u, v;
  fe4_sub(F, &u, &p->y, &p->x); fe4_sub(F, &v, &q->y, &q->x); fe4_mul(F, &aa, &u, &v);
  fe4_add(F, &u, &p->y, &p->x); fe4_add(F, &v, &q->y, &q->x); fe4_mul(F, &bb, &u, &v);
  fe4_mul(F, &u, &C->d2, &p->t); fe4_mul(F, &cc, &u, &q->t);
  fe4_mul(F, &u, &p->z, &q->z); fe4_add(F, &dd, &u, &u);
  fe4_sub(F, &e, &bb, &aa); fe4_sub(F, &f, &dd, &cc); fe4_add(F, &g, &dd, &cc); fe4_add(F, &h, &bb, &aa);
  fe4_mul(F, &r->x, &e, &f); fe4_mul(F, &r->y, &g, &h); fe4_mul(F, &r->z, &f, &g); fe4_mul(F, &r->t, &e, &h);
}
/* curve25519.rs:604-619 + :669-677 */
static void ed_dbl(const ed_curve* C, ed_pt* r, const ed_pt* p) {
  const fe4_field* F = &C->F;
  fe4_t a, b, c, d, e, f, g, h, xy, ab;
  fe4_sqr(F, &a, &p->x); fe4_sqr(F, &b, &p->y); fe4_sqr(F, &c, &p->z); fe4_add(F, &c, &c, &c);
  fe4_neg(F, &d, &a);
  fe4_add(F, &xy, &p->x, &p->y); fe4_sqr(F, &xy, &xy); fe4_add(F, &ab, &a, &b); fe4_sub(F, &e, &xy, &ab);
  fe4_add(F, &g, &d, &b); fe4_sub(F, &f, &g, &c); fe4_sub(F, &h, &d, &b);
  fe4_mul(F, &r->x, &e, &f); fe4_mul(F, &r->y, &g, &h); fe4_mul(F, &r->z, &f, &g); fe4_mul(F, &r->t, &e, &h);
}
static void ed_select(const ed_curve* C, ed_pt* r, const ed_pt* table, int n, unsigned index) {
  ed_identity(C, r);
  uint64_t* dst = (uint64_t*)r;
  for (int j = 0; j < n; j++) {
    uint64_t mask = (uint64_t)0 - (uint64_t)((unsigned)j == index);
    const uint64_t* src = (const uint64_t*)&table[j];
    for (unsigned w = 0; w < sizeof(ed_pt) / 8; w++) dst[w] = (src[w] & mask) | (dst[w] & ~mask);
  }
}
static void ed_to_affine(const ed_curve* C, fe4_t* ax, fe4_t* ay, const ed_pt* p) {
  fe4_t zi;
  fe4_inv(&C->F, &zi, &p->z);
  fe4_mul(&C->F, ax, &p->x, &zi);
  fe4_mul(&C->F, ay, &p->y, &zi);
}
/* curve25519.rs:746-757 */
static void ed_scale_bytes(const ed_curve* C, ed_pt* r, const ed_pt* p, const uint8_t* k_be, int len) {
  ed_pt q, added;
  ed_identity(C, &q);
  for (int i = 0; i < len; i++)
    for (int b = 7; b >= 0; b--) {
      ed_dbl(C, &q, &q);
      ed_add(C, &added, &q, p);
      uint64_t mask = (uint64_t)0 - (uint64_t)((k_be[i] >> b) & 1);
      uint64_t* dq = (uint64_t*)&q;
      const uint64_t* da = (const uint64_t*)&added;
      for (unsigned w = 0; w < sizeof(ed_pt) / 8; w++) dq[w] = (da[w] & mask) | (dq[w] & ~mask);
    }
  *r = q;
}
/* curve25519.rs:881-902 */
static ed_pt* ed_build_comb(const ed_curve* C) {
  ed_pt* tab = (ed_pt*)malloc(sizeof(ed_pt) * 16 * 64);
  ed_pt base;
  ed_from_affine(C, &base, &C->gx, &C->gy);
  for (int i = 0; i < 64; i++) {
    ed_pt* w = tab + 16 * i;
    ed_identity(C, &w[0]);
    w[1] = base;
    for (int j = 2; j < 16; j++) ed_add(C, &w[j], &w[j - 1], &base);
    ed_pt nb;
    ed_add(C, &nb, &w[15], &base);
    for (int j = 1; j < 16; j++) {
      fe4_t ax, ay;
      ed_to_affine(C, &ax, &ay, &w[j]);
      ed_from_affine(C, &w[j], &ax, &ay);
    }
    fe4_t ax, ay;
    ed_to_affine(C, &ax, &ay, &nb);
    ed_from_affine(C, &base, &ax, &ay);
  }
  return tab;
}
/* curve25519.rs:840-851 */
static void ed_mul_base(const ed_curve* C, ed_pt* r, const ed_pt* tab, const uint8_t* n_be) {
  ed_pt q, sel;
  ed_identity(C, &q);
  for (int i = 0; i < 64; i++) {
    uint8_t byte = n_be[32 - 1 - i / 2];
    unsigned digit = (i % 2 == 0) ? (byte & 0x0f) : (byte >> 4);
    ed_select(C, &sel, tab + 16 * i, 16, digit);
    ed_add(C, &q, &q, &sel);
  }
  *r = q;
}

/* Montgomery x-only ladder, src/curve/curve25519.rs:474-513 (a24 = 121666, :374-377) */
static void mont_ladder(const ed_curve* C, fe4_t* out_u, const fe4_t* base_u, const uint8_t* k_be, int len) {
  const fe4_field* F = &C->F;
  fe4_t a24, x1 = *base_u, x2, z2, x3 = *base_u, z3;
  {
    uint8_t buf[32] = {0};
    buf[29] = 0x01; buf[30] = 0xdb; buf[31] = 0x42;
    fe4_from_bytes_be(F, &a24, buf);
  }
  fe4_set_one(F, &x2); fe4_set_zero(&z2); fe4_set_one(F, &z3);
  unsigned swap = 0;
  for (int i = 0; i < len; i++)
    for (int b = 7; b >= 0; b--) {
      unsigned bit = (k_be[i] >> b) & 1;
      swap ^= bit;
      uint64_t mask = (uint64_t)0 - (uint64_t)swap;
      for (int w = 0; w < 4; w++) {
        uint64_t t = mask & (x2.v[w] ^ x3.v[w]); x2.v[w] ^= t; x3.v[w] ^= t;
        t = mask & (z2.v[w] ^ z3.v[w]); z2.v[w] ^= t; z3.v[w] ^= t;
      }
      swap = bit;
      fe4_t a, aa, bq, bb, e, c, d, da, cb, t;
      fe4_add(F, &a, &x2, &z2); fe4_sqr(F, &aa, &a);
      fe4_sub(F, &bq, &x2, &z2); fe4_sqr(F, &bb, &bq);
      fe4_sub(F, &e, &aa, &bb);
      fe4_add(F, &c, &x3, &z3); fe4_sub(F, &d, &x3, &z3);
      fe4_mul(F, &da, &d, &a); fe4_mul(F, &cb, &c, &bq);
      fe4_add(F, &t, &da, &cb); fe4_sqr(F, &x3, &t);
      fe4_sub(F, &t, &da, &cb); fe4_sqr(F, &t, &t); fe4_mul(F, &z3, &x1, &t);
      fe4_mul(F, &x2, &aa, &bb);
      fe4_mul(F, &t, &a24, &e); fe4_add(F, &t, &bb, &t); fe4_mul(F, &z2, &e, &t);
    }
  if (swap) { fe4_t t = x2; x2 = x3; x3 = t; t = z2; z2 = z3; z3 = t; }
  if (fe4_is_zero(&z2)) { fe4_set_zero(out_u); return; }  /* invert_or_zero */
  fe4_t zi;
  fe4_inv(F, &zi, &z2);
  fe4_mul(F, out_u, &x2, &zi);
}

/* ---- global state ------------------------------------------------------- */
static cw4_curve C256;
static cw6_curve C384, CBLS;
static cw9_curve C521;
static ed_curve CED;
static cw4_pt* T256;
static cw6_pt *T384, *TBLS;
static cw9_pt* T521;
static ed_pt* TED;
static pthread_once_t once_curves = PTHREAD_ONCE_INIT;
static pthread_once_t once_tab[ECCX_NCURVES] = { PTHREAD_ONCE_INIT, PTHREAD_ONCE_INIT, PTHREAD_ONCE_INIT, PTHREAD_ONCE_INIT, PTHREAD_ONCE_INIT };

#define INIT_W(C, L, w)                                                         \
  do {                                                                          \
    uint8_t buf[72];                                                            \
    hex2bytes(buf, (w)->p, (w)->fb); fe##L##_field_init(&(C).F, buf, (w)->fb);  \
    hex2bytes(buf, (w)->b, (w)->fb); fe##L##_from_bytes_be(&(C).F, &(C).b, buf); \
    if ((w)->b3[0]) { hex2bytes(buf, (w)->b3, (w)->fb); fe##L##_from_bytes_be(&(C).F, &(C).b3, buf); } \
    hex2bytes(buf, (w)->gx, (w)->fb); fe##L##_from_bytes_be(&(C).F, &(C).gx, buf); \
    hex2bytes(buf, (w)->gy, (w)->fb); fe##L##_from_bytes_be(&(C).F, &(C).gy, buf); \
    (C).a0 = (w)->a0; (C).sb = (w)->sb;                                          \
  } while (0)

static void init_curves(void) {
  INIT_W(C256, 4, &WP[0]);
  INIT_W(C384, 6, &WP[1]);
  INIT_W(C521, 9, &WP[2]);
  INIT_W(CBLS, 6, &WP[3]);
  uint8_t buf[32];
  hex2bytes(buf, ED_P, 32); fe4_field_init(&CED.F, buf, 32);
  hex2bytes(buf, ED_D2, 32); fe4_from_bytes_be(&CED.F, &CED.d2, buf);
  hex2bytes(buf, ED_GX, 32); fe4_from_bytes_be(&CED.F, &CED.gx, buf);
  hex2bytes(buf, ED_GY, 32); fe4_from_bytes_be(&CED.F, &CED.gy, buf);
  (void)ED_D;
}
static void init_t256(void) { T256 = cw4_build_comb(&C256); }
static void init_t384(void) { T384 = cw6_build_comb(&C384); }
static void init_t521(void) { T521 = cw9_build_comb(&C521); }
static void init_tbls(void) { TBLS = cw6_build_comb(&CBLS); }
static void init_ted(void) { TED = ed_build_comb(&CED); }
static void (*const init_tab_fn[ECCX_NCURVES])(void) = { init_t256, init_t384, init_t521, init_tbls, init_ted };

static const int FB[ECCX_NCURVES] = { 32, 48, 66, 48, 32 };
static const int SB[ECCX_NCURVES] = { 32, 48, 66, 32, 32 };

int eccx_oracle_field_bytes(int curve) { return (curve < 0 || curve >= ECCX_NCURVES) ? -1 : FB[curve]; }
int eccx_oracle_scalar_bytes(int curve) { return (curve < 0 || curve >= ECCX_NCURVES) ? -1 : SB[curve]; }

/* ---- per-item work ------------------------------------------------------- */
#define DEF_W_ITEM(L, CURVE, TAB)                                                                           \
  static void w_item_##CURVE(int base_mode, const uint8_t* k, const uint8_t* pt, uint8_t* out, uint8_t* inf, \
                             uint8_t* proj) {                                                               \
    const cw##L##_curve* C = &CURVE;                                                                        \
    int fb = C->F.bytes;                                                                                    \
    cw##L##_pt q;                                                                                           \
    if (base_mode) {                                                                                        \
      cw##L##_mul_base(C, &q, TAB, k);                                                                      \
    } else {                                                                                                \
      cw##L##_pt p;                                                                                         \
      fe##L##_from_bytes_be(&C->F, &p.x, pt);                                                               \
      fe##L##_from_bytes_be(&C->F, &p.y, pt + fb);                                                          \
      fe##L##_set_one(&C->F, &p.z);                                                                         \
      cw##L##_scalar_mul(C, &q, &p, k, (size_t)C->sb);                                                      \
    }                                                                                                       \
    if (proj) {                                                                                             \
      fe##L##_to_bytes_be(&C->F, proj, &q.x);                                                               \
      fe##L##_to_bytes_be(&C->F, proj + fb, &q.y);                                                          \
      fe##L##_to_bytes_be(&C->F, proj + 2 * fb, &q.z);                                                      \
    }                                                                                                       \
    fe##L##_t ax, ay;                                                                                       \
    int present = cw##L##_to_affine(C, &ax, &ay, &q);                                                       \
    fe##L##_to_bytes_be(&C->F, out, &ax);                                                                   \
    fe##L##_to_bytes_be(&C->F, out + fb, &ay);                                                              \
    *inf = (uint8_t)!present;                                                                               \
  }
DEF_W_ITEM(4, C256, T256)
DEF_W_ITEM(6, C384, T384)
DEF_W_ITEM(9, C521, T521)
DEF_W_ITEM(6, CBLS, TBLS)

/* Edwards: affine x||y little-endian (curve25519.rs:138); flag = neutral element */
static void ed_item(int base_mode, const uint8_t* k, const uint8_t* pt, uint8_t* out, uint8_t* inf, uint8_t* proj) {
  ed_pt q;
  if (base_mode) {
    ed_mul_base(&CED, &q, TED, k);
  } else {
    fe4_t x, y;
    ed_pt p;
    fe4_from_bytes_le(&CED.F, &x, pt);
    fe4_from_bytes_le(&CED.F, &y, pt + 32);
    ed_from_affine(&CED, &p, &x, &y);
    ed_scale_bytes(&CED, &q, &p, k, 32);
  }
  if (proj) {
    fe4_to_bytes_le(&CED.F, proj, &q.x); fe4_to_bytes_le(&CED.F, proj + 32, &q.y);
    fe4_to_bytes_le(&CED.F, proj + 64, &q.z); fe4_to_bytes_le(&CED.F, proj + 96, &q.t);
  }
  fe4_t ax, ay, one;
  ed_to_affine(&CED, &ax, &ay, &q);
  fe4_to_bytes_le(&CED.F, out, &ax);
  fe4_to_bytes_le(&CED.F, out + 32, &ay);
  fe4_set_one(&CED.F, &one);
  *inf = (uint8_t)(fe4_is_zero(&ax) && fe4_eq(&ay, &one));
}

typedef struct {
  int curve, base_mode;
  size_t lo, hi;
  const uint8_t *scalars, *points;
  uint8_t *out, *inf, *proj;
} job;

static void* run_job(void* arg) {
  job* j = (job*)arg;
  int fb = FB[j->curve], sb = SB[j->curve];
  int pw = (j->curve == ECCX_ED25519) ? 4 * fb : 3 * fb;
  for (size_t i = j->lo; i < j->hi; i++) {
    const uint8_t* k = j->scalars + i * (size_t)sb;
    const uint8_t* pt = j->points ? j->points + i * 2 * (size_t)fb : NULL;
    uint8_t* o = j->out + i * 2 * (size_t)fb;
    uint8_t* f = j->inf + i;
    uint8_t* pr = j->proj ? j->proj + i * (size_t)pw : NULL;
    switch (j->curve) {
      case ECCX_P256R1: w_item_C256(j->base_mode, k, pt, o, f, pr); break;
      case ECCX_P384R1: w_item_C384(j->base_mode, k, pt, o, f, pr); break;
      case ECCX_P521R1: w_item_C521(j->base_mode, k, pt, o, f, pr); break;
      case ECCX_BLS12_381_G1: w_item_CBLS(j->base_mode, k, pt, o, f, pr); break;
      case ECCX_ED25519: ed_item(j->base_mode, k, pt, o, f, pr); break;
    }
  }
  return NULL;
}

static int run(int curve, int base_mode, size_t n, const uint8_t* scalars, const uint8_t* points, uint8_t* out,
               uint8_t* inf, uint8_t* proj, int threads) {
  if (curve < 0 || curve >= ECCX_NCURVES) return -1;
  if (!scalars || !out || !inf || (!base_mode && !points)) return -2;
  pthread_once(&once_curves, init_curves);
  if (base_mode) pthread_once(&once_tab[curve], init_tab_fn[curve]);
  if (threads < 1) threads = 1;
  if ((size_t)threads > n) threads = n ? (int)n : 1;
  job* jobs = (job*)calloc((size_t)threads, sizeof(job));
  pthread_t* th = (pthread_t*)calloc((size_t)threads, sizeof(pthread_t));
  for (int t = 0; t < threads; t++) {
    jobs[t] = (job){ curve, base_mode, n * (size_t)t / (size_t)threads, n * (size_t)(t + 1) / (size_t)threads,
                     scalars, points, out, inf, proj };
    if (t > 0) pthread_create(&th[t], NULL, run_job, &jobs[t]);
  }
  run_job(&jobs[0]);
  for (int t = 1; t < threads; t++) pthread_join(th[t], NULL);
  free(jobs);
  free(th);
  return 0;
}

/* n x SB big-endian scalars, n x 2FB affine x||y -> n x 2FB affine + n flags.
 * proj (nullable): n x 3FB canonical X||Y||Z of the reference's un-normalised result
 * (Edwards: n x 4FB X||Y||Z||T, little-endian). */
int eccx_oracle_scalarmul_var(int curve, size_t n, const uint8_t* scalars, const uint8_t* points, uint8_t* out,
                              uint8_t* is_inf, uint8_t* proj, int threads) {
  return run(curve, 0, n, scalars, points, out, is_inf, proj, threads);
}

int eccx_oracle_scalarmul_base(int curve, size_t n, const uint8_t* scalars, uint8_t* out, uint8_t* is_inf,
                               uint8_t* proj, int threads) {
  return run(curve, 1, n, scalars, NULL, out, is_inf, proj, threads);
}

/* X25519 / Montgomery ladder.
 *   rfc = 1: protocol::x25519::x25519 (src/protocol/x25519.rs:36-45): scalars little-endian,
 *            clamped; top bit of u masked.
 *   rfc = 0: MontgomeryPoint::scale_bytes (src/curve/curve25519.rs:535-541): scalars are the
 *            big-endian strings the ladder consumes, u reduced mod p as given.
 * u == NULL uses the base point u = 9.  out: n x 32 little-endian u-coordinates;
 * flags[i] = 1 when the result is 0 (point at infinity / low-order input). */
typedef struct { size_t lo, hi; const uint8_t *k, *u; uint8_t *out, *flags; int rfc; } xjob;
static void* run_xjob(void* arg) {
  xjob* j = (xjob*)arg;
  for (size_t i = j->lo; i < j->hi; i++) {
    uint8_t k_be[32], ub[32];
    if (j->rfc) {
      uint8_t k[32];
      memcpy(k, j->k + 32 * i, 32);
      k[0] &= 248; k[31] &= 127; k[31] |= 64;
      for (int b = 0; b < 32; b++) k_be[b] = k[31 - b];
    } else {
      memcpy(k_be, j->k + 32 * i, 32);
    }
    if (j->u) memcpy(ub, j->u + 32 * i, 32); else { memset(ub, 0, 32); ub[0] = 9; }
    if (j->rfc) ub[31] &= 0x7f;
    fe4_t u, r;
    /* reduce the 256-bit little-endian value mod p: split off bit 255 (2^255 = 19 mod p) */
    unsigned top = ub[31] >> 7;
    ub[31] &= 0x7f;
    fe4_from_bytes_le(&CED.F, &u, ub);
    if (top) {
      fe4_t c19; uint8_t b19[32] = {0}; b19[0] = 19;
      fe4_from_bytes_le(&CED.F, &c19, b19);
      fe4_add(&CED.F, &u, &u, &c19);
    }
    mont_ladder(&CED, &r, &u, k_be, 32);
    fe4_to_bytes_le(&CED.F, j->out + 32 * i, &r);
    j->flags[i] = (uint8_t)fe4_is_zero(&r);
  }
  return NULL;
}
int eccx_oracle_x25519(size_t n, const uint8_t* scalars, const uint8_t* u, uint8_t* out, uint8_t* flags, int rfc,
                       int threads) {
  if (!scalars || !out || !flags) return -2;
  pthread_once(&once_curves, init_curves);
  if (threads < 1) threads = 1;
  if ((size_t)threads > n) threads = n ? (int)n : 1;
  xjob* jobs = (xjob*)calloc((size_t)threads, sizeof(xjob));
  pthread_t* th = (pthread_t*)calloc((size_t)threads, sizeof(pthread_t));
  for (int t = 0; t < threads; t++) {
    jobs[t] = (xjob){ n * (size_t)t / (size_t)threads, n * (size_t)(t + 1) / (size_t)threads, scalars, u, out, flags, rfc };
    if (t > 0) pthread_create(&th[t], NULL, run_xjob, &jobs[t]);
  }
  run_xjob(&jobs[0]);
  for (int t = 1; t < threads; t++) pthread_join(th[t], NULL);
  free(jobs);
  free(th);
  return 0;
}

/* The comb table in the reference's on-disk order and encoding
 * (src/params/comb/<curve>.rs): NW x 15 x (x||y), FB bytes each; big-endian for
 * the Weierstrass curves, little-endian for edwards25519. */
int eccx_oracle_comb_table(int curve, uint8_t* out) {
  if (curve < 0 || curve >= ECCX_NCURVES) return -1;
  pthread_once(&once_curves, init_curves);
  pthread_once(&once_tab[curve], init_tab_fn[curve]);
  int fb = FB[curve], nw = 2 * SB[curve];
  for (int i = 0; i < nw; i++)
    for (int j = 1; j < 16; j++) {
      uint8_t* o = out + ((size_t)i * 15 + (size_t)(j - 1)) * 2 * (size_t)fb;
      switch (curve) {
        case ECCX_P256R1: fe4_to_bytes_be(&C256.F, o, &T256[16 * i + j].x); fe4_to_bytes_be(&C256.F, o + fb, &T256[16 * i + j].y); break;
        case ECCX_P384R1: fe6_to_bytes_be(&C384.F, o, &T384[16 * i + j].x); fe6_to_bytes_be(&C384.F, o + fb, &T384[16 * i + j].y); break;
        case ECCX_P521R1: fe9_to_bytes_be(&C521.F, o, &T521[16 * i + j].x); fe9_to_bytes_be(&C521.F, o + fb, &T521[16 * i + j].y); break;
        case ECCX_BLS12_381_G1: fe6_to_bytes_be(&CBLS.F, o, &TBLS[16 * i + j].x); fe6_to_bytes_be(&CBLS.F, o + fb, &TBLS[16 * i + j].y); break;
        case ECCX_ED25519: fe4_to_bytes_le(&CED.F, o, &TED[16 * i + j].x); fe4_to_bytes_le(&CED.F, o + fb, &TED[16 * i + j].y); break;
      }
    }
  return 0;
}

#!/usr/bin/env python3
"""Generate eccoxide_amd/csrc/curve_consts.inc: per-curve Montgomery constants as
32-bit little-endian limb arrays for the HIP kernels.

Self-contained (standard SEC 2 / BLS12-381 / RFC 8032 constants, plain big-int
math); does not import oracle/ or read /root/reference.  tests/ cross-check the
generated values against tests/golden/params.json.

    python tools/gen_curve_consts.py > eccoxide_amd/csrc/curve_consts.inc
"""
import sys

CURVES = [
    # name, p, b, gx, gy, field bytes, scalar bytes, flavour(0 = a=-3, 1 = a=0); the order n of the
    # generator is in ORDERS below (only its bit length reaches the kernels)
    ("P256", 2**256 - 2**224 + 2**192 + 2**96 - 1,
     0x5AC635D8AA3A93E7B3EBBD55769886BC651D06B0CC53B0F63BCE3C3E27D2604B,
     0x6B17D1F2E12C4247F8BCE6E563A440F277037D812DEB33A0F4A13945D898C296,
     0x4FE342E2FE1A7F9B8EE7EB4A7C0F9E162BCE33576B315ECECBB6406837BF51F5, 32, 32, 0),
    ("P384", 2**384 - 2**128 - 2**96 + 2**32 - 1,
     0xB3312FA7E23EE7E4988E056BE3F82D19181D9C6EFE8141120314088F5013875AC656398D8A2ED19D2A85C8EDD3EC2AEF,
     0xAA87CA22BE8B05378EB1C71EF320AD746E1D3B628BA79B9859F741E082542A385502F25DBF55296C3A545E3872760AB7,
     0x3617DE4A96262C6F5D9E98BF9292DC29F8F41DBD289A147CE9DA3113B5F0B8C00A60B1CE1D7E819D7A431D7C90EA0E5F, 48, 48, 0),
    ("P521", 2**521 - 1,
     0x0051953EB9618E1C9A1F929A21A0B68540EEA2DA725B99B315F3B8B489918EF109E156193951EC7E937B1652C0BD3BB1BF073573DF883D2C34F1EF451FD46B503F00,
     0x00C6858E06B70404E9CD9E3ECB662395B4429C648139053FB521F828AF606B4D3DBAA14B5E77EFE75928FE1DC127A2FFA8DE3348B3C1856A429BF97E7E31C2E5BD66,
     0x011839296A789A3BC0045C8A5FB42C7D1BD998F54449579B446817AFBD17273E662C97EE72995EF42640C550B9013FAD0761353C7086A272C24088BE94769FD16650, 66, 66, 0),
    ("BLS12_381", 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB,
     4,
     0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB,
     0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1, 48, 32, 1),
]

# order of the generator (prime) and cofactor of the curve
ORDERS = {
    "P256": (0xFFFFFFFF00000000FFFFFFFFFFFFFFFFBCE6FAADA7179E84F3B9CAC2FC632551, 1),
    "P384": (0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFC7634D81F4372DDF581A0DB248B0A77AECEC196ACCC52973, 1),
    "P521": (0x01FFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFA51868783BF2F966B7FCC0148F709A5D03BB5C9B8899C47AEBB6FB71E91386409, 1),
    "BLS12_381": (0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001, 0x396C8C005555E1568C00AAAB0000AAAB),
}

P25519 = 2**255 - 19
ED_D = (-121665 * pow(121666, -1, P25519)) % P25519
ED_GX = 0x216936D3CD6E53FEC0A4E231FDD6DC5C692CC7609525A7B2C9562D608F25D51A
ED_GY = 0x6666666666666666666666666666666666666666666666666666666666666658


def limbs(x, n):
    return [(x >> (32 * i)) & 0xFFFFFFFF for i in range(n)]


# ---- BLS12-381 G1 endomorphism (subgroup test, GLV split) ------------------------------------
# seed x = -0xd201000000010000; group order r = x^4 - x^2 + 1, so lambda = -x^2 satisfies
# lambda^2 + lambda + 1 = 0 mod r and sigma(X, Y) = (beta X, Y) acts on G1 as [lambda] for one of
# the two primitive cube roots of unity beta in Fp.  Which one is settled by computing [x^2]G with
# textbook affine arithmetic: [x^2]G = -sigma(G) = (beta Gx, -Gy).
BLS_X_ABS = 0xD201000000010000


def _bls_beta():
    _, p, _, gx, gy, _, _, _ = CURVES[3]

    def add(P, Q):
        if P is None:
            return Q
        if Q is None:
            return P
        (x1, y1), (x2, y2) = P, Q
        if x1 == x2:
            if (y1 + y2) % p == 0:
                return None
            lam = 3 * x1 * x1 * pow(2 * y1, -1, p) % p
        else:
            lam = (y2 - y1) * pow(x2 - x1, -1, p) % p
        x3 = (lam * lam - x1 - x2) % p
        return x3, (lam * (x1 - x3) - y1) % p

    acc = None
    for bit in bin(BLS_X_ABS * BLS_X_ABS)[2:]:
        acc = add(acc, acc)
        if bit == "1":
            acc = add(acc, (gx, gy))
    beta = acc[0] * pow(gx, -1, p) % p
    assert pow(beta, 3, p) == 1 and beta != 1 and (p - acc[1]) % p == gy
    return beta


def arr(name, vals):
    body = ", ".join("0x%08xu" % v for v in vals)
    return "  static constexpr uint32_t %s[%d] = {%s};" % (name, len(vals), body)


def emit_field(out, p, L, mersenne=0, pm19=0):
    # Mersenne (2^k - 1) and pseudo-Mersenne (2^255 - 19) fields are kept in plain
    # (non-Montgomery) form: R = 1
    R = 1 if (mersenne or pm19) else 1 << (32 * L)
    out.append("  static constexpr int MERSENNE = %d;  // k if p = 2^k - 1 (plain representation, fold reduction), else 0" % mersenne)
    out.append("  static constexpr int PM19 = %d;  // 1 if p = 2^255 - 19 (plain representation, fold by 38), else 0" % pm19)
    out.append(arr("P", limbs(p, L)))
    out.append(arr("ONE", limbs(R % p, L)))
    out.append(arr("R2", limbs(R * R % p, L)))
    out.append(arr("PM2", limbs(p - 2, L)))
    out.append("  static constexpr uint32_t N0 = 0x%08xu;  // -p^-1 mod 2^32" % ((-pow(p, -1, 1 << 32)) % (1 << 32)))
    # p + 1 (used when N0 == 1: "+ m*p" is then "drop the low limb, + m*(p+1)")
    out.append(arr("PP1", limbs((p + 1) % (1 << (32 * L)), L)))
    out.append("  static constexpr int PBITS = %d;" % p.bit_length())
    # constants of the division-step inversion (inv_gcd.hpp): the modulus in 30-bit limbs,
    # p^-1 mod 2^30, and the number of 30-step batches that the Bernstein-Yang bound
    # floor((49 bits + 57) / 17) divsteps (delta = 1 variant, any odd modulus) rounds up to (but see INV30_HD)
    nl = (p.bit_length() + 1 + 29) // 30 + (1 if (p.bit_length() + 1) % 30 == 0 else 0)
    nl = max(nl, (p.bit_length() + 2 + 29) // 30)  # room for values in (-2p, p) with a sign bit
    out.append("  static constexpr int INV30_N = %d;" % nl)
    out.append("  static constexpr int32_t P30[%d] = {%s};" % (nl, ", ".join("0x%08x" % ((p >> (30 * i)) & 0x3FFFFFFF) for i in range(nl))))
    out.append("  static constexpr uint32_t P30_INV = 0x%08xu;  // p^-1 mod 2^30" % pow(p, -1, 1 << 30))
    # moduli of at most 256 bits: the division step started at delta = 1/2, for which 590 steps are PROVEN
    # to suffice for every odd modulus below 2^256 (convex-hull bound of github.com/sipa/safegcd-bounds, the
    # figure libsecp256k1's modinv32 relies on: 20 batches of 30) -- against 724 / 741 with delta = 1
    hd = p.bit_length() <= 256
    out.append("  static constexpr bool INV30_HD = %s;  // %s" % (
        ("true", "division steps start at delta = 1/2 (590-step bound below 2^256)") if hd else
        ("false", "division steps start at delta = 1 (Bernstein-Yang bound)")))
    out.append("  static constexpr int INV30_BATCHES = %d;" % (20 if hd else ((49 * p.bit_length() + 57) // 17 + 29) // 30))
    return R


def digits(x, bits, n):
    return [(x >> (bits * i)) & ((1 << bits) - 1) for i in range(n)]


def runs_of(e):
    """Binary expansion of e from the top bit as (ones, zeros) runs."""
    segs, b = [], bin(e)[2:]
    i = 0
    while i < len(b):
        j = i
        while j < len(b) and b[j] == "1":
            j += 1
        k = j
        while k < len(b) and b[k] == "0":
            k += 1
        segs.append((j - i, k - j))
        i = k
    assert sum(o + z for o, z in segs) == e.bit_length()
    return segs


def emit_unsat(out, name, sat_name, p, gx, gy, bits, n, kind, extra=(), solinas=(), sparse=False, root_exp=0):
    """Constants of the unsaturated representation: n limbs of `bits` bits in 32-bit registers.
    kind 0: Montgomery, R = 2^(bits*n), p = -1 mod 2^bits (reduce with the digits of p + 1)
    kind 1: Montgomery, general p (m = acc * N0B mod 2^bits)
    kind 2: p = 2^k - 1, plain representation, 2^(bits*n) = 2^(bits*n - k) mod p folded into the product
    kind 3: p = 2^255 - 19, plain representation, 2^(bits*n) = 19 * 2^(bits*n - 255) mod p
    BIAS is 4p written with every limb >= 2^bits - 1 (the largest tight limb) so that
    a + BIAS - b never borrows for tight b < 3p."""
    mont = kind in (0, 1)
    R = (1 << (bits * n)) if mont else 1
    pbits = p.bit_length()
    if kind == 0:
        assert p % (1 << bits) == (1 << bits) - 1, "needs p = -1 mod 2^bits (m = low limb)"
    d = digits(4 * p, bits, n)
    assert sum(x << (bits * i) for i, x in enumerate(d)) == 4 * p or 4 * p >= 1 << (bits * n)
    d[n - 1] = (4 * p) >> (bits * (n - 1))  # the top digit takes whatever is left
    bias = [d[0] + (1 << bits)] + [d[i] + (1 << bits) - 1 for i in range(1, n - 1)] + [d[n - 1] - 1]
    assert sum(b << (bits * i) for i, b in enumerate(bias)) == 4 * p
    assert all(b >= (1 << bits) - 1 for b in bias[:-1]) and all(b < (1 << (bits + 1)) for b in bias)
    # a tight value below 3p (Montgomery kinds) or below 2^(bits*n) (plain kinds) has a top digit <= bias top
    top_tight = ((3 * p) >> (bits * (n - 1))) if (mont or kind == 3) else (1 << bits) - 1
    assert bias[-1] >= top_tight, (name, hex(bias[-1]), hex(top_tight))
    topshift = pbits - bits * (n - 1)
    assert 0 < topshift < bits
    ptop = p >> (bits * (n - 1))
    # quotient estimate for the weak reduction: q = top >> TOPSHIFT when the top digit of p is
    # all ones below bit TOPSHIFT (Solinas / Mersenne), else q = mulhi(top, QMUL) (never too large)
    qmul = 0 if ptop == (1 << topshift) - 1 else (1 << 32) // (ptop + 1)
    out.append("struct %s {" % name)
    out.append("  using Sat = %s;          // saturated twin (byte I/O, validation, normalisation)" % sat_name)
    out.append("  static constexpr int N = %d;     // limbs" % n)
    out.append("  static constexpr int B = %d;    // bits per limb" % bits)
    out.append("  static constexpr int KIND = %d;  // 0 Montgomery p = -1 mod 2^B, 1 Montgomery general, 2 Mersenne (plain), 3 2^255-19 (plain)" % kind)
    out.append("  static constexpr uint32_t FOLD = %du;  // 2^(B*N) mod p where that is small (kinds 2, 3), else 0" % ((1 << (bits * n)) % p if kind in (2, 3) else 0))
    out.append("  static constexpr int PBITS = %d;" % pbits)
    out.append("  static constexpr int TOPSHIFT = %d;  // bit PBITS inside the top limb" % topshift)
    out.append("  static constexpr uint32_t QMUL = 0x%08xu;  // 0: quotient estimate is a shift" % qmul)
    out.append("  static constexpr uint32_t RP = %du;  // floor(R / p), capped" % min(R // p, 1 << 20))
    out.append("  static constexpr uint32_t MASK = 0x%08xu;" % ((1 << bits) - 1))
    out.append("  static constexpr uint32_t N0B = 0x%08xu;  // -p^-1 mod 2^B" % ((-pow(p, -1, 1 << bits)) % (1 << bits)))
    out.append(arr("P", digits(p, bits, n)))
    # p + 1: adding m*p with m = the low limb of the accumulator is "drop that limb, add
    # m*(p+1)", and p + 1 has far fewer non-zero digits (p = -1 mod 2^96 for P-256)
    out.append(arr("PP1", digits(p + 1, bits, n)))
    out.append(arr("P2", digits(2 * p, bits, n)))
    out.append(arr("ONE", digits(R % p, bits, n)))
    out.append(arr("R2", digits(R * R % p, bits, n)))
    out.append(arr("BIAS", bias))
    # R of the saturated twin (2^(32 L) mod p, plain digits): multiplying by it on the way out
    # yields the twin's Montgomery form directly
    Ls = (pbits + 31) // 32
    out.append(arr("RS", digits(((1 << (32 * Ls)) % p) if mont else 1, bits, n)))
    # Solinas form p + 1 = 2^PBITS + sum(sign * 2^e): lets the weak reduction take off q*p with
    # shifts of q instead of multiplications (kind 0 only); terms as (limb, shift, sign)
    if solinas:
        assert (1 << pbits) + sum(sg << e for e, sg in solinas) == p + 1
        terms = [(e // bits, e % bits, sg) for e, sg in solinas]
        assert all(l < n - 1 for l, _, _ in terms)
        pos = [l for l, _, sg in terms if sg > 0]   # a positive term is subtracted: may borrow
        out.append("  static constexpr int SOL_N = %d;" % len(terms))
        out.append("  static constexpr int SOL_LIMB[%d] = {%s};" % (len(terms), ", ".join(str(t[0]) for t in terms)))
        out.append("  static constexpr int SOL_SHIFT[%d] = {%s};" % (len(terms), ", ".join(str(t[1]) for t in terms)))
        out.append("  static constexpr int SOL_SIGN[%d] = {%s};" % (len(terms), ", ".join(str(t[2]) for t in terms)))
        out.append("  static constexpr int SOL_BIAS_FROM = %d;  // lowest limb a subtraction touches" % (min(pos) if pos else n))
    else:
        out.append("  static constexpr int SOL_N = 0;")
    # sparse Montgomery reduction: m*(p+1) added as signed shifted copies of m, one per term of
    # p + 1 (column offset, shift inside the column, sign) instead of one product per non-zero
    # digit -- pays when p + 1 has long runs of ones (P-384: 4 terms against 12 digits)
    if sparse:
        allterms = [(pbits, 1)] + list(solinas)
        out.append("  static constexpr int SPARSE_N = %d;" % len(allterms))
        out.append("  static constexpr int SPARSE_OFF[%d] = {%s};" % (len(allterms), ", ".join(str(e // bits) for e, _ in allterms)))
        out.append("  static constexpr int SPARSE_SHIFT[%d] = {%s};" % (len(allterms), ", ".join(str(e % bits) for e, _ in allterms)))
        out.append("  static constexpr int SPARSE_SIGN[%d] = {%s};" % (len(allterms), ", ".join(str(sg) for _, sg in allterms)))
        assert all(0 < e // bits < n for e, _ in allterms)
    else:
        out.append("  static constexpr int SPARSE_N = 0;")
    out.append(arr("GX", digits(gx * R % p, bits, n)))
    out.append(arr("GY", digits(gy * R % p, bits, n)))
    for cname, val in extra:
        out.append(arr(cname, digits(val * R % p, bits, n)))
    # exponent of the square-root candidate (point decompression): (p + 1) / 4 for p = 3 mod 4,
    # (p - 5) / 8 for 2^255 - 19.  As runs of ones and zeros from the top bit (a few long runs
    # for the Solinas / Mersenne primes: addition chain on x^(2^k - 1)), and as 32-bit words
    # for the 2-bit-window loop used when the pattern is irregular (BLS12-381).
    segs = runs_of(root_exp)
    out.append("  static constexpr int ROOT_BITS = %d;" % root_exp.bit_length())
    out.append("  static constexpr int ROOT_CHAIN = %d;  // 1: few long runs, use the run chain" % (1 if len(segs) <= 8 else 0))
    if len(segs) <= 8:
        out.append("  static constexpr int ROOT_SEGS = %d;" % len(segs))
        out.append("  static constexpr int ROOT_ONES[%d] = {%s};" % (len(segs), ", ".join(str(o) for o, _ in segs)))
        out.append("  static constexpr int ROOT_ZEROS[%d] = {%s};" % (len(segs), ", ".join(str(z) for _, z in segs)))
    else:
        out.append("  static constexpr int ROOT_SEGS = 0;")
    nw = (root_exp.bit_length() + 31) // 32
    out.append("  static constexpr uint32_t ROOT_EXP[%d] = {%s};" % (nw, ", ".join("0x%08xu" % ((root_exp >> (32 * i)) & 0xFFFFFFFF) for i in range(nw))))
    out.append("};")
    out.append("")


def main():
    out = ["// @generated by tools/gen_curve_consts.py -- do not edit.",
           "// Montgomery constants, 32-bit little-endian limbs, R = 2^(32*L).", ""]
    for name, p, b, gx, gy, fb, sb, a0 in CURVES:
        L = (p.bit_length() + 31) // 32
        out.append("struct %s {" % name)
        out.append("  static constexpr int L = %d;   // 32-bit limbs" % L)
        out.append("  static constexpr int FB = %d;  // field bytes" % fb)
        out.append("  static constexpr int SB = %d;  // scalar bytes" % sb)
        out.append("  static constexpr int A0 = %d;  // 1: a = 0 (uses B3), 0: a = -3 (uses B)" % a0)
        order, cof = ORDERS[name]
        out.append("  static constexpr int NBITS = %d;  // bit length of the generator's (prime) order" % order.bit_length())
        out.append("  static constexpr int PRIME_ORDER = %d;  // 1: cofactor 1, every curve point has that order" % (1 if cof == 1 else 0))
        R = emit_field(out, p, L, 521 if name == "P521" else 0)
        out.append(arr("B", limbs(b * R % p, L)))
        out.append(arr("B3", limbs(3 * b * R % p, L)))
        out.append(arr("GX", limbs(gx * R % p, L)))
        out.append(arr("GY", limbs(gy * R % p, L)))
        out.append("};")
        out.append("")
    bb = lambda i: (("CB", CURVES[i][2]), ("CB3", 3 * CURVES[i][2] % CURVES[i][1]))   # curve constant b and 3b
    emit_unsat(out, "P256U", "P256", CURVES[0][1], CURVES[0][3], CURVES[0][4], 29, 9, 0,
               solinas=((224, -1), (192, 1), (96, 1)), extra=bb(0), root_exp=(CURVES[0][1] + 1) // 4)
    emit_unsat(out, "P384U", "P384", CURVES[1][1], CURVES[1][3], CURVES[1][4], 28, 14, 0,
               solinas=((128, -1), (96, -1), (32, 1)), sparse=True, extra=bb(1), root_exp=(CURVES[1][1] + 1) // 4)
    emit_unsat(out, "P521U", "P521", CURVES[2][1], CURVES[2][3], CURVES[2][4], 29, 18, 2, extra=bb(2), root_exp=(CURVES[2][1] + 1) // 4)
    emit_unsat(out, "BLS12_381U", "BLS12_381", CURVES[3][1], CURVES[3][3], CURVES[3][4], 28, 14, 1,
               extra=bb(3) + (("BETA", _bls_beta()),), root_exp=(CURVES[3][1] + 1) // 4)
    # scalar split k = k1 + k2 * x^2 (ECCX_ASSUME_SUBGROUP): x^2 and floor(2^256 / x^2) in 32-bit limbs
    x2 = BLS_X_ABS * BLS_X_ABS
    out.append("struct BLS12_381_GLV {")
    out.append("  static constexpr uint64_t SEED_ABS = 0x%016xull;  // |x|, x the curve's seed (negative)" % BLS_X_ABS)
    out.append(arr("X2", limbs(x2, 4)))
    out.append(arr("MU", limbs((1 << 256) // x2, 5)))
    out.append("  static constexpr int K_BITS = %d;  // bits of the halves: k1 < x^2, k2 <= (2^256 - 1) / x^2" % max(x2.bit_length(), (((1 << 256) - 1) // x2).bit_length()))
    out.append("};")
    out.append("")
    L = 8
    out.append("struct ED25519;")
    emit_unsat(out, "ED25519U", "ED25519", P25519, ED_GX, ED_GY, 29, 9, 3, extra=(("D2", 2 * ED_D % P25519), ("D", ED_D), ("SQRT_M1", pow(2, (P25519 - 1) // 4, P25519))),
               root_exp=(P25519 - 5) // 8)
    out.append("struct ED25519 {")
    out.append("  static constexpr int L = 8;")
    out.append("  static constexpr int FB = 32;")
    out.append("  static constexpr int SB = 32;")
    R = emit_field(out, P25519, L, 0, 1)
    out.append(arr("D2", limbs(2 * ED_D * R % P25519, L)))
    out.append(arr("GX", limbs(ED_GX * R % P25519, L)))
    out.append(arr("GY", limbs(ED_GY * R % P25519, L)))
    out.append("};")
    sys.stdout.write("\n".join(out) + "\n")


if __name__ == "__main__":
    main()

// Exercises include/eccx.hpp (the C++ mirror of eccoxide's Point/Scalar surface) end to end:
// batched Point::mul_base and &Point * &Scalar on p256r1 must agree with each other and with
// the generator's published coordinates.  Exit code 0 = ok.  Needs a GPU to run.
#include <cstdio>
#include <cstring>
#include <vector>

#include "eccx.hpp"

static const uint8_t GX[32] = {0x6b, 0x17, 0xd1, 0xf2, 0xe1, 0x2c, 0x42, 0x47, 0xf8, 0xbc, 0xe6, 0xe5, 0x63, 0xa4, 0x40, 0xf2,
                               0x77, 0x03, 0x7d, 0x81, 0x2d, 0xeb, 0x33, 0xa0, 0xf4, 0xa1, 0x39, 0x45, 0xd8, 0x98, 0xc2, 0x96};
static const uint8_t GY[32] = {0x4f, 0xe3, 0x42, 0xe2, 0xfe, 0x1a, 0x7f, 0x9b, 0x8e, 0xe7, 0xeb, 0x4a, 0x7c, 0x0f, 0x9e, 0x16,
                               0x2b, 0xce, 0x33, 0x57, 0x6b, 0x31, 0x5e, 0xce, 0xcb, 0xb6, 0x40, 0x68, 0x37, 0xbf, 0x51, 0xf5};

int main() {
  try {
    using C = eccx::P256r1;
    eccx::Engine eng(0, eccx::Secrecy::Public);
    const size_t n = 200;
    std::vector<uint8_t> k(n * C::SB, 0), g(n * 2 * C::FB);
    for (size_t i = 0; i < n; ++i) {
      k[i * C::SB + 31] = (uint8_t)(i & 0xff);          // scalars 0..199
      k[i * C::SB + 5] = (uint8_t)(i * 37);             // plus some high bits
      std::memcpy(&g[i * 64], GX, 32);
      std::memcpy(&g[i * 64 + 32], GY, 32);
    }
    std::memset(&k[0], 0, C::SB);                       // unit 0: scalar 0 -> infinity
    std::memset(&k[C::SB], 0, C::SB); k[C::SB + 31] = 1; // unit 1: scalar 1 -> G
    std::memset(&k[2 * C::SB], 0, C::SB); k[2 * C::SB + 31] = 2; // unit 2: scalar 2 -> 2G
    auto scalars = eccx::Scalars<C>::from_bytes(k.data(), n);
    auto base = eccx::Points<C>::mul_base(eng, scalars);
    auto gen = eccx::Points<C>::from_affine(eccx::PointsAffine<C>::from_coordinates(g.data(), n));
    auto var = eccx::on(eng, gen) * scalars;
    const auto& a = base.to_affine();
    const auto& b = var.to_affine();
    if (!a.is_infinity(0) || !b.is_infinity(0)) { std::puts("FAIL: 0*G is not infinity"); return 1; }
    if (std::memcmp(a.x(1), GX, 32) || std::memcmp(a.y(1), GY, 32)) { std::puts("FAIL: 1*G != G"); return 1; }
    for (size_t i = 0; i < n; ++i) {
      if (a.is_infinity(i) != b.is_infinity(i) || std::memcmp(a.x(i), b.x(i), 64)) {
        std::printf("FAIL: mul_base != generic at %zu\n", i);
        return 1;
      }
    }
    // secret-scalar mode (ECCX_CT_SCAN through the engine switch) gives the same points, and the
    // one-time costs can be paid up front
    {
      eccx::Engine ct(0, eccx::Secrecy::Secret);
      ct.prepare<C>(n);
      if (ct.device_bytes() == 0) { std::puts("FAIL: prepare left the context empty"); return 1; }
      auto sbase = eccx::Points<C>::mul_base(ct, scalars);
      auto svar = eccx::on(ct, gen) * scalars;
      // ECCX_ASSUME_SUBGROUP is accepted beside ECCX_CT_SCAN (a no-op on this cofactor-1 curve)
      auto ssub = gen.mul(ct, scalars, /*validate=*/false, eccx::Bases::InSubgroup).to_affine();
      const auto& sb = sbase.to_affine();
      const auto& sv = svar.to_affine();
      for (size_t i = 0; i < n; ++i) {
        if (sb.is_infinity(i) != a.is_infinity(i) || std::memcmp(sb.x(i), a.x(i), 64) ||
            sv.is_infinity(i) != a.is_infinity(i) || std::memcmp(sv.x(i), a.x(i), 64) ||
            ssub.is_infinity(i) != a.is_infinity(i) || std::memcmp(ssub.x(i), a.x(i), 64)) {
          std::printf("FAIL: scanning kernels differ at %zu\n", i);
          return 1;
        }
      }
    }
    // CurveGroup::double and Add: 2*G == G + G == G.dbl(), and (k*G) - (k*G) is infinity
    auto two_g = gen.dbl(eng).to_affine();
    auto g_plus_g = gen.add(eng, gen).to_affine();
    if (std::memcmp(two_g.x(5), a.x(2), 64) || std::memcmp(g_plus_g.x(7), a.x(2), 64)) { std::puts("FAIL: G + G != 2*G"); return 1; }
    auto zero = base.sub(eng, var).to_affine();
    for (size_t i = 0; i < n; ++i)
      if (!zero.is_infinity(i)) { std::printf("FAIL: P - P is not infinity at %zu\n", i); return 1; }
    // verify shape: k*G + k*G == (2k)*G == dbl(k*G), and k*G - k*G is infinity
    auto both = gen.mul_add_base(eng, scalars, scalars).to_affine();
    auto twice = base.dbl(eng).to_affine();
    auto none = gen.mul_add_base(eng, scalars, scalars, /*subtract=*/true).to_affine();
    for (size_t i = 0; i < n; ++i) {
      if (both.is_infinity(i) != twice.is_infinity(i) || std::memcmp(both.x(i), twice.x(i), 64)) {
        std::printf("FAIL: k*G + k*G != 2(k*G) at %zu\n", i);
        return 1;
      }
      if (!none.is_infinity(i)) { std::printf("FAIL: k*G - k*G is not infinity at %zu\n", i); return 1; }
    }
    // compress / decompress: SEC1 records round-trip, the infinity record included, and the
    // generator's encoding is 0x03 || Gx (Gy is odd)
    auto enc = a.compress(eng);
    if (enc.size() != n * 33) { std::puts("FAIL: compressed size"); return 1; }
    if (enc[0] != 0 || enc[33] != 0x03 || std::memcmp(&enc[34], GX, 32)) { std::puts("FAIL: SEC1 encoding of infinity / G"); return 1; }
    auto back = eccx::PointsAffine<C>::decompress(eng, enc.data(), n);
    for (size_t i = 0; i < n; ++i) {
      if (back.is_infinity(i) != a.is_infinity(i) || back.is_rejected(i) || std::memcmp(back.x(i), a.x(i), 64)) {
        std::printf("FAIL: decompress(compress(P)) != P at %zu\n", i);
        return 1;
      }
    }
    enc[33] = 0x05;  // not a SEC1 compressed prefix
    if (!eccx::PointsAffine<C>::decompress(eng, enc.data(), n).is_rejected(1)) { std::puts("FAIL: bad prefix accepted"); return 1; }
    std::puts("mirror_check ok");
    return 0;
  } catch (const std::exception& e) {
    std::printf("FAIL: %s\n", e.what());
    return 2;
  }
}

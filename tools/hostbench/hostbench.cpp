// Times the HOST-buffer entry points of the C ABI (PCIe copies included) for 2^20 p256r1 units:
// the numbers quoted in DESIGN.md section 6.  Needs a GPU.
//   make -C tools/hostbench && tools/hostbench/hostbench
#include <chrono>
#include <cstdio>
#include <vector>

#include "eccx.h"

int main() {
  eccx_ctx* ctx = nullptr;
  if (eccx_init(0, &ctx)) { std::puts("eccx_init failed"); return 1; }
  const size_t n = (size_t)1 << 20;
  std::vector<uint8_t> k(n * 32), pts(n * 64), out(n * 64), fl(n);
  for (size_t i = 0; i < k.size(); ++i) k[i] = (uint8_t)((i * 2654435761u) >> 13);
  for (size_t i = 0; i < n; ++i) k[i * 32] &= 0x7f;   // below the group order
  if (eccx_scalarmul_base(ctx, ECCX_P256R1, n, k.data(), pts.data(), fl.data(), nullptr, 0)) {
    std::printf("mul_base failed: %s\n", eccx_last_error(ctx));
    return 1;
  }
  for (int rep = 0; rep < 3; ++rep) {
    auto t0 = std::chrono::steady_clock::now();
    int rc1 = eccx_scalarmul_base(ctx, ECCX_P256R1, n, k.data(), out.data(), fl.data(), nullptr, 0);
    auto t1 = std::chrono::steady_clock::now();
    int rc2 = eccx_scalarmul_var(ctx, ECCX_P256R1, n, k.data(), pts.data(), out.data(), fl.data(), nullptr, 0);
    auto t2 = std::chrono::steady_clock::now();
    if (rc1 || rc2) { std::printf("failed: %s\n", eccx_last_error(ctx)); return 1; }
    std::printf("host buffers, 2^20 p256r1 units: mul_base %.2f ms   variable base %.2f ms\n",
                std::chrono::duration<double, std::milli>(t1 - t0).count(),
                std::chrono::duration<double, std::milli>(t2 - t1).count());
  }
  eccx_shutdown(ctx);
  return 0;
}

// Times the HOST-buffer entry points of the C ABI (PCIe copies included) against the device-resident forms
// for 2^20 p256r1 units: the numbers behind bench.py's `host_path` object and DESIGN.md section 6.  Needs a GPU.
//   make -C tools/hostbench && tools/hostbench/hostbench > profiles/r03_hostbench.jsonl
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

#include "eccx.h"

static double ms(std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
  return std::chrono::duration<double, std::milli>(b - a).count();
}

int main() {
  eccx_ctx* ctx = nullptr;
  if (eccx_init(0, &ctx)) { std::puts("eccx_init failed"); return 1; }
  const size_t n = (size_t)1 << 20;
  std::vector<uint8_t> k(n * 32), pts(n * 64), out(n * 64), fl(n);
  for (size_t i = 0; i < k.size(); ++i) k[i] = (uint8_t)((i * 2654435761u) >> 13);
  for (size_t i = 0; i < n; ++i) k[i * 32] &= 0x7f;  // below the group order
  eccx_prepare(ctx, ECCX_P256R1, ECCX_PREP_BASE | ECCX_PREP_CT);
  eccx_reserve(ctx, ECCX_P256R1, n, ECCX_PREP_VAR | ECCX_PREP_CT | ECCX_PREP_HOST);
  if (eccx_scalarmul_base(ctx, ECCX_P256R1, n, k.data(), pts.data(), fl.data(), nullptr, 0)) {
    std::printf("mul_base failed: %s\n", eccx_last_error(ctx));
    return 1;
  }
  const size_t bytes_before = eccx_device_bytes(ctx);
  // device-resident reference: the same batch with the buffers already on the GPU
  uint8_t *d_k, *d_p, *d_o, *d_f;
  hipMalloc(&d_k, n * 32); hipMalloc(&d_p, n * 64); hipMalloc(&d_o, n * 64); hipMalloc(&d_f, n);
  hipMemcpy(d_k, k.data(), n * 32, hipMemcpyHostToDevice);
  hipMemcpy(d_p, pts.data(), n * 64, hipMemcpyHostToDevice);
  std::vector<double> hb, hv, db, dv, cin, cout_, hbs, hvs, dbs, dvs;  // ..s: secret scalars (ECCX_CT_SCAN)
  for (int rep = 0; rep < 7; ++rep) {
    auto t0 = std::chrono::steady_clock::now();
    int rc1 = eccx_scalarmul_base(ctx, ECCX_P256R1, n, k.data(), out.data(), fl.data(), nullptr, 0);
    auto t1 = std::chrono::steady_clock::now();
    int rc2 = eccx_scalarmul_var(ctx, ECCX_P256R1, n, k.data(), pts.data(), out.data(), fl.data(), nullptr, 0);
    auto t2 = std::chrono::steady_clock::now();
    int rc3 = eccx_scalarmul_base_dev(ctx, ECCX_P256R1, n, d_k, d_o, d_f, nullptr, 0, nullptr);
    hipDeviceSynchronize();
    auto t3 = std::chrono::steady_clock::now();
    int rc4 = eccx_scalarmul_var_dev(ctx, ECCX_P256R1, n, d_k, d_p, d_o, d_f, nullptr, 0, nullptr);
    hipDeviceSynchronize();
    auto t4 = std::chrono::steady_clock::now();
    hipMemcpy(d_p, pts.data(), n * 64, hipMemcpyHostToDevice);  // the copies alone, pageable host memory
    auto t5 = std::chrono::steady_clock::now();
    hipMemcpy(out.data(), d_o, n * 64, hipMemcpyDeviceToHost);
    auto t6 = std::chrono::steady_clock::now();
    int rc5 = eccx_scalarmul_base(ctx, ECCX_P256R1, n, k.data(), out.data(), fl.data(), nullptr, ECCX_CT_SCAN);
    auto t7 = std::chrono::steady_clock::now();
    int rc6 = eccx_scalarmul_var(ctx, ECCX_P256R1, n, k.data(), pts.data(), out.data(), fl.data(), nullptr, ECCX_CT_SCAN);
    auto t8 = std::chrono::steady_clock::now();
    int rc7 = eccx_scalarmul_base_dev(ctx, ECCX_P256R1, n, d_k, d_o, d_f, nullptr, ECCX_CT_SCAN, nullptr);
    hipDeviceSynchronize();
    auto t9 = std::chrono::steady_clock::now();
    int rc8 = eccx_scalarmul_var_dev(ctx, ECCX_P256R1, n, d_k, d_p, d_o, d_f, nullptr, ECCX_CT_SCAN, nullptr);
    hipDeviceSynchronize();
    auto t10 = std::chrono::steady_clock::now();
    if (rc1 || rc2 || rc3 || rc4 || rc5 || rc6 || rc7 || rc8) { std::printf("failed: %s\n", eccx_last_error(ctx)); return 1; }
    hbs.push_back(ms(t6, t7)); hvs.push_back(ms(t7, t8)); dbs.push_back(ms(t8, t9)); dvs.push_back(ms(t9, t10));
    hb.push_back(ms(t0, t1)); hv.push_back(ms(t1, t2)); db.push_back(ms(t2, t3)); dv.push_back(ms(t3, t4));
    cin.push_back(ms(t4, t5)); cout_.push_back(ms(t5, t6));
  }
  auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
  const double h2d64 = med(cin), d2h64 = med(cout_);
  std::printf("{\"units\": %zu, \"curve\": \"p256r1\", \"host_mul_base_ms\": %.3f, \"device_mul_base_ms\": %.3f, "
              "\"host_var_ms\": %.3f, \"device_var_ms\": %.3f, \"h2d_64MiB_ms\": %.3f, \"d2h_64MiB_ms\": %.3f, "
              "\"mul_base_copies_ms\": %.3f, \"var_copies_ms\": %.3f, "
              "\"secret_host_mul_base_ms\": %.3f, \"secret_device_mul_base_ms\": %.3f, \"secret_host_var_ms\": %.3f, \"secret_device_var_ms\": %.3f, "
              "\"device_bytes_before\": %zu, \"device_bytes_after\": %zu}\n",
              n, med(hb), med(db), med(hv), med(dv), h2d64, d2h64, h2d64 * 0.5 + d2h64 * (65.0 / 64.0), h2d64 * 1.5 + d2h64 * (65.0 / 64.0),
              med(hbs), med(dbs), med(hvs), med(dvs), bytes_before, eccx_device_bytes(ctx));
  hipFree(d_k); hipFree(d_p); hipFree(d_o); hipFree(d_f);
  eccx_shutdown(ctx);
  return 0;
}

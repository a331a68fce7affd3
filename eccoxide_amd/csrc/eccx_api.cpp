// C ABI of the engine (include/eccx.h): contexts, buffers, comb-table construction and
// kernel launches.  No arithmetic happens on the host: the comb tables are produced by
// the engine's own variable-base kernel at first use (each entry (j+1)*16^i*G is the
// generator times a scalar with a single non-zero nibble), which stands in for the
// reference's generated constants (src/params/comb/<curve>.rs, sage/comb.sage) and its
// build_comb_table (src/curve/projective.rs:451-472, src/curve/curve25519.rs:881-902).
#include "../../include/eccx.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "launch.hpp"

namespace {

using eccx::CurveOps;

constexpr int NCURVES = 5;
// internal kernel option bits (kernels.hpp)
constexpr uint32_t K_BASE_IS_GENERATOR = 1u << 0;
constexpr uint32_t K_OUT_TABLE = 1u << 1;
constexpr uint32_t K_VALIDATE = 1u << 2;
constexpr uint32_t K_OUT_ROWS = 1u << 3;
constexpr uint32_t K_NEGATE_B = 1u << 5;  // second operand negated: a - b, u1*G - u2*Q
constexpr uint32_t K_CT_SCAN = 1u << 6;   // table lookups scan every entry (ECCX_CT_SCAN)
constexpr uint32_t K_ONLY_MARKED = 1u << 7;  // variable base: redo only the units marked 0xFE

const CurveOps* ops_of(int curve) {
  switch (curve) {
    case ECCX_P256R1: return &eccx::ops_P256();
    case ECCX_P384R1: return &eccx::ops_P384();
    case ECCX_P521R1: return &eccx::ops_P521();
    case ECCX_BLS12_381_G1: return &eccx::ops_BLS12_381();
    case ECCX_ED25519: return &eccx::ops_ED25519();
    default: return nullptr;
  }
}

}  // namespace

struct eccx_ctx {
  int device = 0;
  int cus = 0;
  hipStream_t stream = nullptr;
  hipStream_t in_stream = nullptr, out_stream = nullptr;  // host-buffer entry points: copies beside the compute
  uint32_t* comb[NCURVES] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  uint32_t* comb_u[NCURVES] = {nullptr, nullptr, nullptr, nullptr, nullptr};  // unsaturated-field copies
  uint32_t* comb_lds[NCURVES] = {nullptr, nullptr, nullptr, nullptr, nullptr};  // images for the LDS variant
  uint32_t* comb_ct[NCURVES] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // signed-window tables of the secret-scalar path
  uint32_t* comb_ctg[NCURVES] = {nullptr, nullptr, nullptr, nullptr, nullptr};  // ... of its lane-gather form (ECCX_CT_GATHER)
  std::mutex comb_mu;
  uint32_t* scratch = nullptr;
  size_t scratch_words = 0;
  uint32_t* jac = nullptr;  // un-normalised results of the fast kernels
  size_t jac_words = 0;
  std::mutex scratch_mu;
  size_t table_bytes = 0;  // fixed-base tables owned by the context (eccx_device_bytes)
  // device-side I/O buffers of the HOST-buffer entry points (grow-only, like the slabs; eccx_reserve with
  // ECCX_PREP_HOST sizes them) and the events their chunked copies use: after warm-up a host-buffer call
  // allocates and frees nothing
  static constexpr int NIO = 7;
  uint8_t* io[NIO] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  size_t io_cap[NIO] = {0, 0, 0, 0, 0, 0, 0};
  static constexpr int NEV = 10;
  hipEvent_t evs[NEV] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  std::mutex err_mu;       // err is written by whichever host thread's call failed last
  std::string err;
  void set_err(std::string m) {
    std::lock_guard<std::mutex> g(err_mu);
    err = std::move(m);
  }
};

namespace {

#define HIP_TRY(ctx, call)                                                                     \
  do {                                                                                         \
    hipError_t e_ = (call);                                                                    \
    if (e_ != hipSuccess) {                                                                    \
      (ctx)->set_err(std::string(#call) + ": " + hipGetErrorString(e_));                       \
      return e_ == hipErrorOutOfMemory ? ECCX_ERR_NOMEM : ECCX_ERR_HIP;                        \
    }                                                                                          \
  } while (0)

// argument errors leave a message too: eccx_last_error() must never hand out a stale HIP string for them
int arg_err(eccx_ctx* ctx, const char* what) {
  if (ctx) ctx->set_err(std::string("bad argument: ") + what);
  return ECCX_ERR_ARG;
}
int curve_err(eccx_ctx* ctx) {
  if (ctx) ctx->set_err("unknown curve id");
  return ECCX_ERR_CURVE;
}

// Persistent grid: the variable-base kernel keeps a 16-row window table per lane in a
// scratch slab indexed by workgroup, so the grid is capped at a few workgroups per CU
// and each workgroup strides over the batch.
int grid_for(const eccx_ctx* ctx, size_t n) {
  size_t need = (n + eccx::LAUNCH_WG - 1) / eccx::LAUNCH_WG;
  size_t cap = (size_t)ctx->cus * 4;
  return (int)std::max<size_t>(1, std::min(need, cap));
}

// grow-only device buffer owned by the context
int ensure_buffer(eccx_ctx* ctx, uint32_t** buf, size_t* have, size_t words) {
  if (words <= *have) return ECCX_OK;
  if (*buf) {
    HIP_TRY(ctx, hipDeviceSynchronize());
    HIP_TRY(ctx, hipFree(*buf));
    *buf = nullptr;
    *have = 0;
  }
  HIP_TRY(ctx, hipMalloc(buf, words * sizeof(uint32_t)));
  *have = words;
  return ECCX_OK;
}

int ensure_scratch(eccx_ctx* ctx, int row_words, int grid) {
  // 17 rows per lane: the signed-window table of the fast kernel holds entries 1..16
  size_t words = (size_t)grid * 17 * eccx::LAUNCH_WG * (size_t)row_words;
  std::lock_guard<std::mutex> g(ctx->scratch_mu);
  return ensure_buffer(ctx, &ctx->scratch, &ctx->scratch_words, words);
}

// buffer of un-normalised result rows (X, Y, Z limbs) for the batched normalisation
int ensure_rows(eccx_ctx* ctx, const CurveOps* ops, size_t n) {
  std::lock_guard<std::mutex> g(ctx->scratch_mu);
  return ensure_buffer(ctx, &ctx->jac, &ctx->jac_words, n * (size_t)ops->info.jac_words);
}

// I/O slots of the host-buffer entry points
enum { IO_K = 0, IO_P = 1, IO_O = 2, IO_F = 3, IO_J = 4, IO_A = 5, IO_B = 6 };
int ensure_io(eccx_ctx* ctx, int slot, size_t bytes, uint8_t** out) {
  std::lock_guard<std::mutex> g(ctx->scratch_mu);
  if (bytes > ctx->io_cap[slot]) {
    if (ctx->io[slot]) {
      HIP_TRY(ctx, hipDeviceSynchronize());
      HIP_TRY(ctx, hipFree(ctx->io[slot]));
      ctx->io[slot] = nullptr;
      ctx->io_cap[slot] = 0;
    }
    HIP_TRY(ctx, hipMalloc(&ctx->io[slot], bytes));
    ctx->io_cap[slot] = bytes;
  }
  *out = ctx->io[slot];
  return ECCX_OK;
}

int norm_grid(const eccx_ctx* ctx, size_t n) {
  size_t tile = (size_t)eccx::LAUNCH_WG * 8;  // sizing only: the kernels stride over tiles
  size_t tiles = (n + tile - 1) / tile;
  return (int)std::max<size_t>(1, std::min(tiles, (size_t)ctx->cus * 4));
}

int flat_grid(const eccx_ctx* ctx, size_t n) {
  size_t need = (n + eccx::LAUNCH_WG - 1) / eccx::LAUNCH_WG;
  return (int)std::max<size_t>(1, std::min(need, (size_t)ctx->cus * 8));
}

// Variable base.  mirror = run the reference-mirroring kernel (homogeneous RCB formulas,
// un-normalised X:Y:Z available); otherwise the fast Jacobian kernel + batched
// normalisation where the curve has one.
int launch_var(eccx_ctx* ctx, const CurveOps* ops, size_t n, const uint8_t* d_scalars, const uint8_t* d_points,
               uint8_t* d_out, uint8_t* d_flags, uint8_t* d_proj, uint32_t kopts, bool mirror, hipStream_t s,
               bool glv = false, bool ct = false, bool ct_prime = false) {
  if (n == 0) return ECCX_OK;
  if (ct && ops->var_ct && !d_proj && d_points && !(kopts & (K_OUT_TABLE | K_BASE_IS_GENERATOR))) {
    // secret scalars: the windowed ladder that reads every table row at every lookup and resolves its special
    // cases by selects (Weierstrass: kernels_coz.hpp, CT = true; edwards25519: k_ed_scalarmul_var_unsat, CT = true).
    // Weierstrass units the kernel marks -- from the BASE POINT alone: order <= 2^(WB-1), or not a curve point --
    // are skipped by the normalisation and redone by the reference-mirroring ladder with the scan (complete
    // formulas), which writes their bytes itself; the complete Edwards formulas have no such units.
    const bool ed = ops->info.edwards != 0;
    const bool prime = ct_prime && ops->var_ct_prime;
    const int grid = prime ? ops->var_ct_prime_grid(ctx->cus, n) : ops->var_ct_grid(ctx->cus, n);
    const int grid2 = std::min(ops->var_grid ? ops->var_grid(ctx->cus, n) : grid_for(ctx, n), ctx->cus);
    int rc = ensure_scratch(ctx, ed ? ops->info.row5_words : ops->coz_row_words, grid);
    if (rc) return rc;
    if (!ed) {
      rc = ensure_scratch(ctx, ops->info.row_words, grid2);
      if (rc) return rc;
    }
    rc = ensure_rows(ctx, ops, n);
    if (rc) return rc;
    // ct_prime: the bases are vouched to have prime order (ECCX_ASSUME_SUBGROUP on a curve with a cofactor)
    HIP_TRY(ctx, (prime ? ops->var_ct_prime : ops->var_ct)(grid, s, n, d_scalars, d_points, ctx->jac, d_flags, ctx->scratch,
                                                           kopts & ~K_CT_SCAN));
    HIP_TRY(ctx, ops->to_affine_var(norm_grid(ctx, n), s, n, ctx->jac, d_out, d_flags));
    if (!ed)
      HIP_TRY(ctx, ops->var(grid2, s, n, d_scalars, d_points, d_out, d_flags, nullptr, ctx->scratch,
                            (kopts & K_VALIDATE) | K_CT_SCAN | K_ONLY_MARKED));
    return ECCX_OK;
  }
  const bool fast = !mirror && ops->var_fast && !d_proj && !(kopts & K_OUT_TABLE);
  if (fast && ops->var_coz && d_points && !(kopts & K_BASE_IS_GENERATOR)) {
    // Weierstrass curves: the ladder over an affine window table (kernels_coz.hpp); glv: bases known to be in the
    // prime-order subgroup.  Units with a base point of order <= 16 come back marked and are redone
    // by the generic ladder, which otherwise only reads the flags.
    const int g = glv ? 1 : 0;
    const int grid = ops->var_coz_grid(ctx->cus, n, g);
    const int grid2 = ops->var_fast_grid ? ops->var_fast_grid(ctx->cus, n) : grid_for(ctx, n);
    int rc = ensure_scratch(ctx, ops->coz_row_words, grid);
    if (rc) return rc;
    rc = ensure_scratch(ctx, ops->info.row5_words, grid2);
    if (rc) return rc;
    rc = ensure_rows(ctx, ops, n);
    if (rc) return rc;
    HIP_TRY(ctx, ops->var_coz(grid, s, n, d_scalars, d_points, ctx->jac, d_flags, ctx->scratch, kopts, g));
    HIP_TRY(ctx, ops->var_fast(grid2, s, n, d_scalars, d_points, ctx->jac, d_flags, ctx->scratch, kopts | K_ONLY_MARKED));
    HIP_TRY(ctx, ops->to_affine_var(norm_grid(ctx, n), s, n, ctx->jac, d_out, d_flags));
    return ECCX_OK;
  }
  // persistent grid sized to the kernel's real residency (registers decide it)
  int grid = fast ? (ops->var_fast_grid ? ops->var_fast_grid(ctx->cus, n) : grid_for(ctx, n))
                  : (ops->var_grid ? ops->var_grid(ctx->cus, n) : grid_for(ctx, n));
  if (fast) {
    int rc = ensure_scratch(ctx, ops->info.row5_words, grid);
    if (rc) return rc;
    rc = ensure_rows(ctx, ops, n);
    if (rc) return rc;
    HIP_TRY(ctx, ops->var_fast(grid, s, n, d_scalars, d_points, ctx->jac, d_flags, ctx->scratch, kopts));
    HIP_TRY(ctx, ops->to_affine_var(norm_grid(ctx, n), s, n, ctx->jac, d_out, d_flags));
    return ECCX_OK;
  }
  if (ops->info.row_words) {
    int rc = ensure_scratch(ctx, ops->info.row_words, grid);
    if (rc) return rc;
  }
  if (!d_proj && !(kopts & K_OUT_TABLE) && ops->to_affine_hom) {
    // un-normalised rows, then one inversion per to_affine_u() units
    int rc = ensure_rows(ctx, ops, n);
    if (rc) return rc;
    HIP_TRY(ctx, ops->var(grid, s, n, d_scalars, d_points, reinterpret_cast<uint8_t*>(ctx->jac), d_flags, nullptr,
                          ctx->scratch, kopts | K_OUT_ROWS));
    HIP_TRY(ctx, ops->to_affine_hom(norm_grid(ctx, n), s, n, ctx->jac, d_out, d_flags));
    return ECCX_OK;
  }
  HIP_TRY(ctx, ops->var(grid, s, n, d_scalars, d_points, d_out, d_flags, d_proj, ctx->scratch, kopts));
  return ECCX_OK;
}

// scalars with one non-zero nibble: row w*16+d encodes d * 16^w (big-endian, SB bytes)
std::vector<uint8_t> comb_scalars(const CurveOps* ops) {
  int sb = ops->info.sb, nw = 2 * sb;
  std::vector<uint8_t> k((size_t)nw * 16 * sb, 0);
  for (int w = 0; w < nw; ++w)
    for (int d = 0; d < 16; ++d) {
      uint8_t* row = k.data() + ((size_t)w * 16 + d) * sb;
      row[sb - 1 - w / 2] = (uint8_t)((w & 1) ? (d << 4) : d);
    }
  return k;
}

// device allocations scoped to one call: freed on every return path
struct DevMem {
  std::vector<void*> ptrs;
  ~DevMem() {
    for (void* p : ptrs)
      if (p) (void)hipFree(p);
  }
  template <class T>
  hipError_t alloc(T** out, size_t bytes) {
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e == hipSuccess) ptrs.push_back(p);
    *out = static_cast<T*>(p);
    return e;
  }
  void release(void* p) {  // ownership moves to the caller
    for (auto& q : ptrs)
      if (q == p) q = nullptr;
  }
};

// The build runs on the context's own stream and uses the context's scratch slab and row buffer, which
// work enqueued earlier -- on the caller's stream, or on any other stream when the call is eccx_prepare --
// may still be using: the build waits for the whole device first (it blocks the host anyway; eccx_prepare
// pays it up front).
int ensure_comb(eccx_ctx* ctx, int curve, const CurveOps* ops, hipStream_t caller) {
  std::lock_guard<std::mutex> g(ctx->comb_mu);
  if (ctx->comb[curve]) return ECCX_OK;
  (void)caller;
  HIP_TRY(ctx, hipDeviceSynchronize());
  int nw = 2 * ops->info.sb;
  size_t rows = (size_t)nw * 16;
  std::vector<uint8_t> k = comb_scalars(ops);
  DevMem mem;
  uint8_t* d_k = nullptr;
  uint32_t* d_tab = nullptr;
  HIP_TRY(ctx, mem.alloc(&d_k, k.size()));
  HIP_TRY(ctx, mem.alloc(&d_tab, rows * ops->info.table_words * sizeof(uint32_t)));
  HIP_TRY(ctx, hipMemcpyAsync(d_k, k.data(), k.size(), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemsetAsync(d_tab, 0, rows * ops->info.table_words * sizeof(uint32_t), ctx->stream));
  int rc = launch_var(ctx, ops, rows, d_k, nullptr, reinterpret_cast<uint8_t*>(d_tab), nullptr, nullptr,
                      K_BASE_IS_GENERATOR | K_OUT_TABLE, true, ctx->stream);
  if (rc) return rc;
  // wide-window table of the unsaturated fixed-base kernel: entry (w, d) = d * 2^(W*w) * G,
  // computed by the engine's own variable-base path (entries whose scalar would not fit the
  // scalar width belong to digits the top window cannot produce and stay zero)
  uint32_t* d_utab = nullptr;
  if (ops->base_unsat) {
    const int W = ops->comb_bits, sbytes = ops->info.sb;
    const int nwin = (8 * sbytes + W - 1) / W;
    const size_t urows = (size_t)nwin << W, pb = 2 * (size_t)ops->info.fb;
    std::vector<uint8_t> uk(urows * sbytes, 0);
    for (int w = 0; w < nwin; ++w)
      for (uint32_t d = 0; d < (1u << W); ++d) {
        uint8_t* row = uk.data() + (((size_t)w << W) + d) * sbytes;
        bool fits = true;
        for (int bit = 0; bit < W; ++bit)
          if ((d >> bit) & 1u) {
            const int pos = w * W + bit;
            if (pos >= 8 * sbytes) { fits = false; break; }
            row[sbytes - 1 - (pos >> 3)] |= (uint8_t)(1u << (pos & 7));
          }
        if (!fits) std::fill(row, row + sbytes, (uint8_t)0);
      }
    uint8_t *d_uk = nullptr, *d_aff = nullptr, *d_fl = nullptr;
    HIP_TRY(ctx, mem.alloc(&d_uk, uk.size()));
    HIP_TRY(ctx, mem.alloc(&d_aff, urows * pb));
    HIP_TRY(ctx, mem.alloc(&d_fl, urows));
    HIP_TRY(ctx, mem.alloc(&d_utab, urows * (size_t)ops->utable_words * sizeof(uint32_t)));
    HIP_TRY(ctx, hipMemcpyAsync(d_uk, uk.data(), uk.size(), hipMemcpyHostToDevice, ctx->stream));
    rc = launch_var(ctx, ops, urows, d_uk, nullptr, d_aff, d_fl, nullptr, K_BASE_IS_GENERATOR, false, ctx->stream);
    if (rc) return rc;
    HIP_TRY(ctx, ops->comb_convert(ctx->stream, urows, d_aff, d_utab));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // uk lives on the host until the copy is done
  }
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  mem.release(d_tab);  // the tables now belong to the context
  if (d_utab) mem.release(d_utab);
  ctx->comb[curve] = d_tab;
  ctx->comb_u[curve] = d_utab;
  ctx->table_bytes += rows * ops->info.table_words * sizeof(uint32_t);
  if (d_utab) {
    const int W = ops->comb_bits;
    ctx->table_bytes += ((size_t)((8 * ops->info.sb + W - 1) / W) << W) * (size_t)ops->utable_words * sizeof(uint32_t);
  }
  return ECCX_OK;
}

// table image of the LDS-resident fixed-base variant: entry (w, d) = d * 2^(bits*w) * G for the
// digits 0 .. 2^(bits-1), built like the wide tables by the engine's own variable-base path
int ensure_comb_lds(eccx_ctx* ctx, int curve, const CurveOps* ops, hipStream_t caller) {
  std::lock_guard<std::mutex> g(ctx->comb_mu);
  if (ctx->comb_lds[curve]) return ECCX_OK;
  (void)caller;
  HIP_TRY(ctx, hipDeviceSynchronize());  // as ensure_comb
  const int sbytes = ops->info.sb;
  const size_t entries = (size_t)ops->lds_windows * ops->lds_digits, pb = 2 * (size_t)ops->info.fb;
  std::vector<uint8_t> k(entries * sbytes, 0);
  for (int w = 0; w < ops->lds_windows; ++w)
    for (int d = 0; d < ops->lds_digits; ++d) {
      uint8_t* row = k.data() + ((size_t)w * ops->lds_digits + d) * sbytes;
      bool fits = true;
      for (int bit = 0; bit < 16; ++bit)
        if ((d >> bit) & 1) {
          const int pos = w * ops->lds_bits + bit;
          if (pos >= 8 * sbytes) { fits = false; break; }
          row[sbytes - 1 - (pos >> 3)] |= (uint8_t)(1u << (pos & 7));
        }
      if (!fits) std::fill(row, row + sbytes, (uint8_t)0);
    }
  DevMem mem;
  uint8_t *d_k = nullptr, *d_aff = nullptr, *d_fl = nullptr;
  uint32_t* d_tab = nullptr;
  HIP_TRY(ctx, mem.alloc(&d_k, k.size()));
  HIP_TRY(ctx, mem.alloc(&d_aff, entries * pb));
  HIP_TRY(ctx, mem.alloc(&d_fl, entries));
  HIP_TRY(ctx, mem.alloc(&d_tab, entries * (size_t)ops->lds_entry_words * sizeof(uint32_t)));
  HIP_TRY(ctx, hipMemcpyAsync(d_k, k.data(), k.size(), hipMemcpyHostToDevice, ctx->stream));
  int rc = launch_var(ctx, ops, entries, d_k, nullptr, d_aff, d_fl, nullptr, K_BASE_IS_GENERATOR, false, ctx->stream);
  if (rc) return rc;
  HIP_TRY(ctx, ops->lds_convert(ctx->stream, entries, d_aff, d_tab));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  mem.release(d_tab);
  ctx->comb_lds[curve] = d_tab;
  ctx->table_bytes += entries * (size_t)ops->lds_entry_words * sizeof(uint32_t);
  return ECCX_OK;
}

// table of the secret-scalar fixed-base kernels (kernels_ct.hpp): entry (w, d) = d * 2^(ct_bits * w) * G for
// d = 1 .. ct_entries, built by the engine's own variable-base path (the generator is public; digits the top
// window cannot produce get the zero scalar and stay unused)
int ensure_comb_ct(eccx_ctx* ctx, int curve, const CurveOps* ops, hipStream_t caller, bool gather = false) {
  std::lock_guard<std::mutex> g(ctx->comb_mu);
  uint32_t** slot_ptr = gather ? &ctx->comb_ctg[curve] : &ctx->comb_ct[curve];
  if (*slot_ptr) return ECCX_OK;
  const int ct_bits = gather ? ops->ctg_bits : ops->ct_bits, ct_windows = gather ? ops->ctg_windows : ops->ct_windows,
            ct_entries = gather ? ops->ctg_entries : ops->ct_entries;
  if (!(gather ? ops->base_ctg : ops->base_ct) || !ops->ct_convert) {
    ctx->set_err("no secret-scalar fixed-base kernel for this curve");
    return ECCX_ERR_ARG;
  }
  (void)caller;
  HIP_TRY(ctx, hipDeviceSynchronize());  // as ensure_comb
  const int sbytes = ops->info.sb, W = ct_bits;
  const size_t entries = (size_t)ct_windows * ct_entries, pb = 2 * (size_t)ops->info.fb;
  // one more row: 2^(8 SB - 1) * G, from which the one reachable entry whose scalar does not fit SB bytes is made
  // below -- the top window's digit 2^(8 SB - W w_top) stands for 2^(8 SB) * G (a scalar of all ones recodes to it)
  std::vector<uint8_t> k((entries + 1) * sbytes, 0);
  for (int w = 0; w < ct_windows; ++w)
    for (int d = 1; d <= ct_entries; ++d) {
      uint8_t* row = k.data() + ((size_t)w * ct_entries + (size_t)(d - 1)) * sbytes;
      bool fits = true;
      for (int bit = 0; bit < 16; ++bit)
        if ((d >> bit) & 1) {
          const int pos = w * W + bit;
          if (pos >= 8 * sbytes) { fits = false; break; }
          row[sbytes - 1 - (pos >> 3)] |= (uint8_t)(1u << (pos & 7));
        }
      if (!fits) std::fill(row, row + sbytes, (uint8_t)0);
    }
  k[entries * sbytes] = 0x80;
  DevMem mem;
  uint8_t *d_k = nullptr, *d_aff = nullptr, *d_fl = nullptr;
  uint32_t* d_tab = nullptr;
  const size_t tab_bytes = entries * (size_t)ops->ct_entry_words * sizeof(uint32_t);
  HIP_TRY(ctx, mem.alloc(&d_k, k.size()));
  HIP_TRY(ctx, mem.alloc(&d_aff, (entries + 1) * pb));
  HIP_TRY(ctx, mem.alloc(&d_fl, entries + 1));
  HIP_TRY(ctx, mem.alloc(&d_tab, tab_bytes));
  HIP_TRY(ctx, hipMemcpyAsync(d_k, k.data(), k.size(), hipMemcpyHostToDevice, ctx->stream));
  int rc = launch_var(ctx, ops, entries + 1, d_k, nullptr, d_aff, d_fl, nullptr, K_BASE_IS_GENERATOR, false, ctx->stream);
  if (rc) return rc;
  {
    const int w_top = ct_windows - 1, shift = 8 * sbytes - W * w_top;  // 2^(8 SB) = 2^shift * 2^(W w_top)
    if (shift >= 0 && shift < W && (1 << shift) <= ct_entries) {
      const size_t slot = (size_t)w_top * ct_entries + (size_t)((1 << shift) - 1);
      rc = eccx_point_add_dev(ctx, curve, 1, d_aff + entries * pb, nullptr, d_aff + entries * pb, nullptr, d_aff + slot * pb,
                              d_fl + slot, 0, ctx->stream);  // the complete addition doubles
      if (rc) return rc;
    }
  }
  HIP_TRY(ctx, ops->ct_convert(ctx->stream, entries, d_aff, d_tab));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  mem.release(d_tab);
  *slot_ptr = d_tab;
  ctx->table_bytes += tab_bytes;
  return ECCX_OK;
}

uint32_t kopts_of(uint32_t opts) { return (opts & ECCX_VALIDATE_POINTS) ? K_VALIDATE : 0u; }

size_t proj_bytes(const CurveOps* ops) { return (size_t)(ops->info.edwards ? 4 : 3) * ops->info.fb; }

// ---- host-buffer entry points: copies in, kernels, copies out ------------------------------------------------
// Device-side copies of the caller's buffers live in the context's I/O slots (grow-only: nothing is allocated or
// freed once a batch of this size has been seen, or after eccx_reserve(..., ECCX_PREP_HOST)), the events come from
// the context's pool.
struct HostIn { uint8_t* dev; const uint8_t* host; size_t width; };   // width: bytes per unit
struct HostOut { uint8_t* host; const uint8_t* dev; size_t width; };

// Large batches go through in chunks so that the PCIe copies of chunk i+1 (in) and i-1 (out) run beside the
// kernels of chunk i: three streams, events between them.  The host buffers are pageable, so each copy call
// returns when its data has been staged; the kernels it overlaps with are already enqueued.  `chunked` is the
// caller's judgement that the kernels outlast the copies (variable base: 20.8 -> 19.3 ms for 2^20 p256 units; the
// public-scalar fixed-base kernels are shorter than their copies and lose to the per-chunk launch costs, 3.4 ->
// 4.5 ms; the secret-scalar combs take the chunks, 5.3 -> 4.3 ms).  launch(lo, cnt) enqueues the kernels of units
// lo .. lo + cnt on ctx->stream and returns an ECCX code.
template <class Launch>
int host_pipeline(eccx_ctx* ctx, size_t n, const HostIn* ins, int nins, const HostOut* outs, int nouts, bool chunked,
                  Launch launch) {
  const size_t nchunks = (chunked && n >= ((size_t)1 << 17)) ? 4 : 1;
  static_assert(2 * 4 <= eccx_ctx::NEV, "two events per chunk");
  const size_t step = ((n + nchunks - 1) / nchunks + 4095) / 4096 * 4096;
  // a single chunk keeps everything on the compute stream (crossing streams costs ~1 ms of idle gaps)
  hipStream_t s_in = nchunks > 1 ? ctx->in_stream : ctx->stream;
  hipStream_t s_out = nchunks > 1 ? ctx->out_stream : ctx->stream;
  int next_ev = 0;
  auto settle = [&]() {  // on an error path: nothing of this call may still be running when the caller's buffers go away
    (void)hipStreamSynchronize(s_in);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipStreamSynchronize(s_out);
  };
#define TRY2_(call)                                 \
  do {                                              \
    hipError_t e_ = (call);                         \
    if (e_ != hipSuccess) {                         \
      ctx->set_err(std::string(#call) + ": " + hipGetErrorString(e_)); \
      settle();                                     \
      return e_ == hipErrorOutOfMemory ? ECCX_ERR_NOMEM : ECCX_ERR_HIP; \
    }                                               \
  } while (0)
  hipEvent_t prev_done = nullptr;
  size_t prev_lo = 0, prev_cnt = 0;
  auto copy_out = [&](size_t lo, size_t cnt, hipEvent_t done) -> hipError_t {
    hipError_t r = s_out == ctx->stream ? hipSuccess : hipStreamWaitEvent(s_out, done, 0);
    for (int o = 0; o < nouts && r == hipSuccess; ++o)
      if (outs[o].host)
        r = hipMemcpyAsync(outs[o].host + lo * outs[o].width, outs[o].dev + lo * outs[o].width, cnt * outs[o].width,
                           hipMemcpyDeviceToHost, s_out);
    return r;
  };
  for (size_t lo = 0; lo < n; lo += step) {
    const size_t cnt = std::min(step, n - lo);
    for (int i = 0; i < nins; ++i)
      if (ins[i].host)
        TRY2_(hipMemcpyAsync(ins[i].dev + lo * ins[i].width, ins[i].host + lo * ins[i].width, cnt * ins[i].width,
                             hipMemcpyHostToDevice, s_in));
    hipEvent_t done = nullptr;
    if (nchunks > 1) {
      hipEvent_t in_ready = ctx->evs[next_ev++];
      done = ctx->evs[next_ev++];
      TRY2_(hipEventRecord(in_ready, s_in));
      TRY2_(hipStreamWaitEvent(ctx->stream, in_ready, 0));
    }
    const int rc = launch(lo, cnt);
    if (rc) { settle(); return rc; }
    if (nchunks > 1) {
      TRY2_(hipEventRecord(done, ctx->stream));
      if (prev_done) TRY2_(copy_out(prev_lo, prev_cnt, prev_done));
    }
    prev_done = done; prev_lo = lo; prev_cnt = cnt;
  }
  TRY2_(copy_out(prev_lo, prev_cnt, prev_done));
  TRY2_(hipStreamSynchronize(s_out));
  if (s_out != ctx->stream) TRY2_(hipStreamSynchronize(ctx->stream));
#undef TRY2_
  return ECCX_OK;
}

// host-buffer wrapper shared by var / base
int run_host(eccx_ctx* ctx, int curve, bool base, size_t n, const uint8_t* scalars, const uint8_t* points,
             uint8_t* out, uint8_t* flags, uint8_t* proj, uint32_t opts) {
  const CurveOps* ops = ops_of(curve);
  if (!ctx) return ECCX_ERR_ARG;
  if (!ops) return curve_err(ctx);
  if (n == 0) return ECCX_OK;
  if (!scalars || !out || !flags || (!base && !points)) return arg_err(ctx, "null buffer");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  size_t sb = ops->info.sb, pb = 2 * (size_t)ops->info.fb;
  uint8_t *d_k = nullptr, *d_p = nullptr, *d_o = nullptr, *d_f = nullptr, *d_j = nullptr;
  int rc = ensure_io(ctx, IO_K, n * sb, &d_k);
  if (!rc) rc = ensure_io(ctx, IO_O, n * pb, &d_o);
  if (!rc) rc = ensure_io(ctx, IO_F, n, &d_f);
  if (!rc && !base) rc = ensure_io(ctx, IO_P, n * pb, &d_p);
  if (!rc && proj) rc = ensure_io(ctx, IO_J, n * proj_bytes(ops), &d_j);
  if (rc) return rc;
  // in chunks where the kernels outlast the copies (host_pipeline); proj (the mirror kernels' X:Y:Z) in one piece
  const bool chunked = !proj && (!base || ((opts & ECCX_CT_SCAN) && !(opts & (ECCX_MIRROR_REFERENCE | ECCX_TABLE_IN_L2))));
  const HostIn ins[2] = {{d_k, scalars, sb}, {d_p, base ? nullptr : points, pb}};
  const HostOut outs[3] = {{out, d_o, pb}, {flags, d_f, 1}, {proj, d_j, proj_bytes(ops)}};
  return host_pipeline(ctx, n, ins, 2, outs, 3, chunked, [&](size_t lo, size_t cnt) {
    if (base) return eccx_scalarmul_base_dev(ctx, curve, cnt, d_k + lo * sb, d_o + lo * pb, d_f + lo, d_j, opts, ctx->stream);
    return eccx_scalarmul_var_dev(ctx, curve, cnt, d_k + lo * sb, d_p + lo * pb, d_o + lo * pb, d_f + lo, d_j, opts, ctx->stream);
  });
}

int run_sharded(eccx_ctx** ctxs, int nctx, int curve, bool base, size_t n, const uint8_t* scalars,
                const uint8_t* points, uint8_t* out, uint8_t* flags, uint32_t opts) {
  const CurveOps* ops = ops_of(curve);
  if (!ops) return ECCX_ERR_CURVE;
  if (!ctxs || nctx < 1) return ECCX_ERR_ARG;
  for (int i = 0; i < nctx; ++i)
    if (!ctxs[i]) return ECCX_ERR_ARG;
  size_t sb = ops->info.sb, pb = 2 * (size_t)ops->info.fb;
  std::vector<int> rcs((size_t)nctx, ECCX_OK);
  std::vector<std::thread> th;
  for (int g = 0; g < nctx; ++g) {
    size_t lo = n * (size_t)g / (size_t)nctx, hi = n * (size_t)(g + 1) / (size_t)nctx;  // contiguous shards
    th.emplace_back([=, &rcs]() {
      rcs[(size_t)g] = run_host(ctxs[g], curve, base, hi - lo, scalars + lo * sb, base ? nullptr : points + lo * pb,
                                out + lo * pb, flags + lo, nullptr, opts);
    });
  }
  for (auto& t : th) t.join();
  for (int rc : rcs)
    if (rc) return rc;
  return ECCX_OK;
}

}  // namespace

extern "C" {

int eccx_field_bytes(int curve) {
  const CurveOps* o = ops_of(curve);
  return o ? o->info.fb : ECCX_ERR_CURVE;
}
int eccx_scalar_bytes(int curve) {
  const CurveOps* o = ops_of(curve);
  return o ? o->info.sb : ECCX_ERR_CURVE;
}

int eccx_init(int device, eccx_ctx** out_ctx) {
  if (!out_ctx) return ECCX_ERR_ARG;
  *out_ctx = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return ECCX_ERR_HIP;
  if (hipSetDevice(device) != hipSuccess) return ECCX_ERR_HIP;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return ECCX_ERR_HIP;
  eccx_ctx* ctx = new (std::nothrow) eccx_ctx();
  if (!ctx) return ECCX_ERR_NOMEM;
  ctx->device = device;
  ctx->cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
      hipStreamCreateWithFlags(&ctx->in_stream, hipStreamNonBlocking) != hipSuccess ||
      hipStreamCreateWithFlags(&ctx->out_stream, hipStreamNonBlocking) != hipSuccess) {
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    if (ctx->in_stream) (void)hipStreamDestroy(ctx->in_stream);
    delete ctx;
    return ECCX_ERR_HIP;
  }
  for (auto& e : ctx->evs)
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
      eccx_shutdown(ctx);
      return ECCX_ERR_HIP;
    }
  *out_ctx = ctx;
  return ECCX_OK;
}

void eccx_shutdown(eccx_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  for (auto& t : ctx->comb)
    if (t) (void)hipFree(t);
  for (auto& t : ctx->comb_u)
    if (t) (void)hipFree(t);
  for (auto& t : ctx->comb_lds)
    if (t) (void)hipFree(t);
  for (auto& t : ctx->comb_ct)
    if (t) (void)hipFree(t);
  for (auto& t : ctx->comb_ctg)
    if (t) (void)hipFree(t);
  if (ctx->scratch) (void)hipFree(ctx->scratch);
  if (ctx->jac) (void)hipFree(ctx->jac);
  for (auto& b : ctx->io)
    if (b) (void)hipFree(b);
  for (auto& e : ctx->evs)
    if (e) (void)hipEventDestroy(e);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  if (ctx->in_stream) (void)hipStreamDestroy(ctx->in_stream);
  if (ctx->out_stream) (void)hipStreamDestroy(ctx->out_stream);
  delete ctx;
}

const char* eccx_last_error(const eccx_ctx* ctx) {
  if (!ctx) return "null context";
  // a copy per calling thread: the string another thread's failing call replaces is never handed out
  static thread_local std::string copy;
  eccx_ctx* c = const_cast<eccx_ctx*>(ctx);
  std::lock_guard<std::mutex> g(c->err_mu);
  copy = c->err;
  return copy.c_str();
}

const char* eccx_strerror(int code) {
  switch (code) {
    case ECCX_OK: return "ok";
    case ECCX_ERR_CURVE: return "unknown curve id";
    case ECCX_ERR_ARG: return "bad argument";
    case ECCX_ERR_HIP: return "HIP runtime error";
    case ECCX_ERR_NOMEM: return "out of device memory";
    default: return "unknown error";
  }
}

int eccx_prepare(eccx_ctx* ctx, int curve, uint32_t what) {
  const CurveOps* ops = ops_of(curve);
  if (!ctx) return ECCX_ERR_ARG;
  if (!ops) return curve_err(ctx);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int rc = ECCX_OK;
  if (what & ECCX_PREP_BASE) rc = ensure_comb(ctx, curve, ops, ctx->stream);
  if (!rc && (what & ECCX_PREP_BASE_LDS)) {
    if (!ops->base_lds || !ops->lds_convert) {
      ctx->set_err("ECCX_PREP_BASE_LDS: this curve has no LDS-resident fixed-base kernel (edwards25519 only)");
      return ECCX_ERR_ARG;
    }
    rc = ensure_comb_lds(ctx, curve, ops, ctx->stream);
  }
  if (!rc && (what & ECCX_PREP_CT)) rc = ensure_comb_ct(ctx, curve, ops, ctx->stream, false);
  if (!rc && (what & ECCX_PREP_CT_GATHER)) rc = ensure_comb_ct(ctx, curve, ops, ctx->stream, true);
  return rc;
}

int eccx_reserve(eccx_ctx* ctx, int curve, size_t max_n, uint32_t what) {
  const CurveOps* ops = ops_of(curve);
  if (!ctx) return ECCX_ERR_ARG;
  if (!ops) return curve_err(ctx);
  if (max_n == 0) return ECCX_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int rc = ensure_rows(ctx, ops, max_n);  // every entry point writes un-normalised rows first
  if (rc) return rc;
  if (what & ECCX_PREP_VAR) {  // window-table slab of the default ladder (also the fused double-scalar kernel)
    const int grid = ops->var_fast_grid ? ops->var_fast_grid(ctx->cus, max_n) : grid_for(ctx, max_n);
    rc = ensure_scratch(ctx, ops->info.row5_words, grid);
    if (rc) return rc;
    if (ops->var_coz) {  // the affine-table ladder's slab (both forms)
      int g = std::max(ops->var_coz_grid(ctx->cus, max_n, 0), ops->var_coz_grid(ctx->cus, max_n, 1));
      if (ops->var_coz_fused_grid) g = std::max(g, ops->var_coz_fused_grid(ctx->cus, max_n));  // the verify shape
      rc = ensure_scratch(ctx, ops->coz_row_words, g);
      if (rc) return rc;
    }
  }
  if (what & ECCX_PREP_HOST) {  // device-side copies of the host-buffer entry points' arguments
    const size_t pb = 2 * (size_t)ops->info.fb, sbytes = (size_t)ops->info.sb;
    uint8_t* dummy = nullptr;
    rc = ensure_io(ctx, IO_K, max_n * std::max(pb, sbytes), &dummy);   // scalars; first operand of the group law
    if (!rc) rc = ensure_io(ctx, IO_P, max_n * pb, &dummy);
    if (!rc) rc = ensure_io(ctx, IO_O, max_n * pb, &dummy);
    if (!rc) rc = ensure_io(ctx, IO_F, max_n, &dummy);
    if (!rc) rc = ensure_io(ctx, IO_J, max_n * sbytes, &dummy);            // second scalar of the verify shape
    if (!rc) rc = ensure_io(ctx, IO_A, max_n, &dummy);
    if (!rc) rc = ensure_io(ctx, IO_B, max_n, &dummy);
    if (rc) return rc;
  }
  if ((what & ECCX_PREP_CT) && ops->var_ct) {  // secret scalars: the scanning ladder + (Weierstrass) its fix-up
    const bool ed = ops->info.edwards != 0;
    rc = ensure_scratch(ctx, ed ? ops->info.row5_words : ops->coz_row_words, ops->var_ct_grid(ctx->cus, max_n));
    if (rc) return rc;
    if (ops->var_ct_prime) {  // ECCX_CT_SCAN | ECCX_ASSUME_SUBGROUP: a kernel of its own, possibly at another occupancy
      rc = ensure_scratch(ctx, ops->coz_row_words, ops->var_ct_prime_grid(ctx->cus, max_n));
      if (rc) return rc;
    }
    if (!ed) {
      const int grid2 = std::min(ops->var_grid ? ops->var_grid(ctx->cus, max_n) : grid_for(ctx, max_n), ctx->cus);
      rc = ensure_scratch(ctx, ops->info.row_words, grid2);
      if (rc) return rc;
    }
  }
  // slab of the reference-mirroring ladder (also what ECCX_CT_SCAN runs on a curve without a scanning fast ladder)
  const bool mirror_slab = (what & ECCX_PREP_MIRROR) || ((what & ECCX_PREP_CT) && !ops->var_ct);
  if (mirror_slab && ops->info.row_words) {
    const int grid = ops->var_grid ? ops->var_grid(ctx->cus, max_n) : grid_for(ctx, max_n);
    rc = ensure_scratch(ctx, ops->info.row_words, grid);
    if (rc) return rc;
  }
  return ECCX_OK;
}

size_t eccx_device_bytes(const eccx_ctx* ctx) {
  if (!ctx) return 0;
  eccx_ctx* c = const_cast<eccx_ctx*>(ctx);
  std::lock_guard<std::mutex> g1(c->comb_mu);
  std::lock_guard<std::mutex> g2(c->scratch_mu);
  size_t io = 0;
  for (size_t b : c->io_cap) io += b;
  return c->table_bytes + (c->scratch_words + c->jac_words) * sizeof(uint32_t) + io;
}

int eccx_scalarmul_var_dev(eccx_ctx* ctx, int curve, size_t n, const void* d_scalars, const void* d_points,
                           void* d_out, void* d_flags, void* d_proj, uint32_t opts, void* stream) {
  const CurveOps* ops = ops_of(curve);
  if (!ctx) return ECCX_ERR_ARG;
  if (!ops) return curve_err(ctx);
  if (n == 0) return ECCX_OK;
  if (!d_scalars || !d_points || !d_out || !d_flags) return arg_err(ctx, "null buffer");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t s = static_cast<hipStream_t>(stream);  // NULL = HIP's default stream
  // ECCX_CT_SCAN: the reference-mirroring ladder (complete formulas, no data-dependent branch) with
  // select_from_table's full scan; edwards25519's mirror ladder is bit-serial and has no table
  const bool ct = (opts & ECCX_CT_SCAN) != 0;
  // ECCX_CT_SCAN | ECCX_ASSUME_SUBGROUP: not the endomorphism ladder (it has no secret-scalar form) but the secret-scalar
  // ladder with the accumulator == +-entry selects confined to the windows a PRIME-ORDER base can reach, as on the
  // cofactor-1 curves (bls12_381_g1: -11 % multiplies; what sk * H(m) needs).  No effect on the other curves.
  const bool subgroup = (opts & ECCX_ASSUME_SUBGROUP) != 0;
  // secret scalars: the scanning affine-table ladder where the curve has one (Weierstrass), unless the
  // reference-mirroring kernels are asked for (ECCX_MIRROR_REFERENCE, proj): those scan as the reference does
  const bool ct_fast = ct && !(opts & ECCX_MIRROR_REFERENCE) && !d_proj && ops->var_ct;
  return launch_var(ctx, ops, n, static_cast<const uint8_t*>(d_scalars), static_cast<const uint8_t*>(d_points),
                    static_cast<uint8_t*>(d_out), static_cast<uint8_t*>(d_flags), static_cast<uint8_t*>(d_proj),
                    kopts_of(opts) | (ct ? K_CT_SCAN : 0u), ct || (opts & ECCX_MIRROR_REFERENCE) != 0, s,
                    subgroup && !ct, ct_fast, ct_fast && subgroup);
}

int eccx_scalarmul_base_dev(eccx_ctx* ctx, int curve, size_t n, const void* d_scalars, void* d_out, void* d_flags,
                            void* d_proj, uint32_t opts, void* stream) {
  const CurveOps* ops = ops_of(curve);
  if (!ctx) return ECCX_ERR_ARG;
  if (!ops) return curve_err(ctx);
  if (n == 0) return ECCX_OK;
  if (!d_scalars || !d_out || !d_flags) return arg_err(ctx, "null buffer");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t s = static_cast<hipStream_t>(stream);  // NULL = HIP's default stream
  int rc = ECCX_OK;
  size_t need = (n + eccx::LAUNCH_WG - 1) / eccx::LAUNCH_WG;
  if ((opts & ECCX_CT_SCAN) && !(opts & (ECCX_MIRROR_REFERENCE | ECCX_TABLE_IN_L2)) && !d_proj && ops->base_ct) {
    // secret scalars: signed windows, every entry of a window read by every lane (kernels_ct.hpp)
    if (opts & ECCX_TABLE_IN_LDS) {
      ctx->set_err("ECCX_CT_SCAN | ECCX_TABLE_IN_LDS: the LDS-resident comb indexes its table by the digit");
      return ECCX_ERR_ARG;
    }
    const bool gather = (opts & ECCX_CT_GATHER) != 0 && ops->base_ctg;
    rc = ensure_comb_ct(ctx, curve, ops, s, gather);
    if (rc) return rc;
    rc = ensure_rows(ctx, ops, n);
    if (rc) return rc;
    const int cgrid = (int)std::max<size_t>(1, std::min(need, (size_t)ctx->cus * 16));
    HIP_TRY(ctx, (gather ? ops->base_ctg : ops->base_ct)(cgrid, s, n, static_cast<const uint8_t*>(d_scalars),
                                                         gather ? ctx->comb_ctg[curve] : ctx->comb_ct[curve], ctx->jac,
                                                         static_cast<uint8_t*>(d_flags)));
    HIP_TRY(ctx, ops->to_affine_var(norm_grid(ctx, n), s, n, ctx->jac, static_cast<uint8_t*>(d_out),
                                    static_cast<uint8_t*>(d_flags)));
    return ECCX_OK;
  }
  rc = ensure_comb(ctx, curve, ops, s);
  if (rc) return rc;
  // up to 16 workgroups per CU: at 2^20 units every lane then takes ONE unit and the hardware's dispatcher
  // balances the tail (measured against 8 and 4 per CU: Ed25519 0.733 / 0.747 / 0.768 ms, P-256 1.110 / 1.126 / 1.142)
  int grid = (int)std::max<size_t>(1, std::min(need, (size_t)ctx->cus * 16));
  // default: 16-bit windows over the engine's own wide table (the 4-bit comb of the reference's
  // layout stays reachable through ECCX_MIRROR_REFERENCE / ECCX_TABLE_IN_LDS / ECCX_TABLE_IN_L2)
  const uint32_t ct = (opts & ECCX_CT_SCAN) ? K_CT_SCAN : 0u;  // reference-layout 4-bit comb, every entry read
  if (!d_proj && !(opts & (ECCX_MIRROR_REFERENCE | ECCX_TABLE_IN_LDS | ECCX_TABLE_IN_L2 | ECCX_CT_SCAN)) && ops->base_unsat &&
      ctx->comb_u[curve]) {
    rc = ensure_rows(ctx, ops, n);
    if (rc) return rc;
    const int ugrid = ops->var_fast_grid ? std::max(grid, ops->var_fast_grid(ctx->cus, n)) : grid;
    HIP_TRY(ctx, ops->base_unsat(ugrid, s, n, static_cast<const uint8_t*>(d_scalars), ctx->comb_u[curve], ctx->jac,
                                 static_cast<uint8_t*>(d_flags)));
    HIP_TRY(ctx, ops->to_affine_var(norm_grid(ctx, n), s, n, ctx->jac, static_cast<uint8_t*>(d_out),
                                    static_cast<uint8_t*>(d_flags)));
    return ECCX_OK;
  }
  // LDS-resident table (ECCX_TABLE_IN_LDS, edwards25519): signed 6-bit windows, the widest table
  // that fits 160 KiB
  if (!d_proj && !ct && (opts & ECCX_TABLE_IN_LDS) && ops->base_lds && ops->lds_convert && ops->to_affine_var) {
    rc = ensure_comb_lds(ctx, curve, ops, s);
    if (rc) return rc;
    rc = ensure_rows(ctx, ops, n);
    if (rc) return rc;
    HIP_TRY(ctx, ops->base_lds(ctx->cus, s, n, static_cast<const uint8_t*>(d_scalars), ctx->comb_lds[curve], ctx->jac,
                               static_cast<uint8_t*>(d_flags)));
    HIP_TRY(ctx, ops->to_affine_var(norm_grid(ctx, n), s, n, ctx->jac, static_cast<uint8_t*>(d_out),
                                    static_cast<uint8_t*>(d_flags)));
    return ECCX_OK;
  }
  if (!d_proj && ops->to_affine_hom) {
    rc = ensure_rows(ctx, ops, n);
    if (rc) return rc;
    HIP_TRY(ctx, ops->base(grid, s, n, static_cast<const uint8_t*>(d_scalars), ctx->comb[curve],
                           reinterpret_cast<uint8_t*>(ctx->jac), static_cast<uint8_t*>(d_flags), nullptr,
                           K_OUT_ROWS | ct));
    HIP_TRY(ctx, ops->to_affine_hom(norm_grid(ctx, n), s, n, ctx->jac, static_cast<uint8_t*>(d_out),
                                    static_cast<uint8_t*>(d_flags)));
    return ECCX_OK;
  }
  HIP_TRY(ctx, ops->base(grid, s, n, static_cast<const uint8_t*>(d_scalars), ctx->comb[curve],
                         static_cast<uint8_t*>(d_out), static_cast<uint8_t*>(d_flags),
                         static_cast<uint8_t*>(d_proj), ct));
  return ECCX_OK;
}

int eccx_scalarmul_var(eccx_ctx* ctx, int curve, size_t n, const uint8_t* scalars, const uint8_t* points,
                       uint8_t* out, uint8_t* flags, uint8_t* proj, uint32_t opts) {
  return run_host(ctx, curve, false, n, scalars, points, out, flags, proj, opts);
}

int eccx_scalarmul_base(eccx_ctx* ctx, int curve, size_t n, const uint8_t* scalars, uint8_t* out, uint8_t* flags,
                        uint8_t* proj, uint32_t opts) {
  return run_host(ctx, curve, true, n, scalars, nullptr, out, flags, proj, opts);
}

int eccx_point_add_dev(eccx_ctx* ctx, int curve, size_t n, const void* d_a, const void* d_a_inf, const void* d_b,
                       const void* d_b_inf, void* d_out, void* d_flags, uint32_t opts, void* stream) {
  const CurveOps* ops = ops_of(curve);
  if (!ctx) return ECCX_ERR_ARG;
  if (!ops) return curve_err(ctx);
  if (n == 0) return ECCX_OK;
  if (!d_a || !d_b || !d_out || !d_flags) return arg_err(ctx, "null buffer");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t s = static_cast<hipStream_t>(stream);
  int rc = ensure_rows(ctx, ops, n);
  if (rc) return rc;
  // default: complete addition on the unsaturated field; ECCX_MIRROR_REFERENCE: the saturated pair
  const bool mirror = (opts & ECCX_MIRROR_REFERENCE) != 0 || !ops->point_add_u;
  HIP_TRY(ctx, (mirror ? ops->point_add : ops->point_add_u)(
                   flat_grid(ctx, n), s, n, static_cast<const uint8_t*>(d_a), static_cast<const uint8_t*>(d_a_inf),
                   static_cast<const uint8_t*>(d_b), static_cast<const uint8_t*>(d_b_inf), ctx->jac,
                   static_cast<uint8_t*>(d_flags), (opts & ECCX_SUBTRACT) ? K_NEGATE_B : 0u));
  HIP_TRY(ctx, (mirror ? ops->to_affine_hom : ops->to_affine_add_u)(norm_grid(ctx, n), s, n, ctx->jac, static_cast<uint8_t*>(d_out),
                                                                    static_cast<uint8_t*>(d_flags)));
  return ECCX_OK;
}

int eccx_point_add(eccx_ctx* ctx, int curve, size_t n, const uint8_t* a, const uint8_t* a_inf, const uint8_t* b,
                   const uint8_t* b_inf, uint8_t* out, uint8_t* flags, uint32_t opts) {
  const CurveOps* ops = ops_of(curve);
  if (!ctx) return ECCX_ERR_ARG;
  if (!ops) return curve_err(ctx);
  if (n == 0) return ECCX_OK;
  if (!a || !b || !out || !flags) return arg_err(ctx, "null buffer");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const size_t pb = 2 * (size_t)ops->info.fb;
  uint8_t *d_a = nullptr, *d_b = nullptr, *d_ai = nullptr, *d_bi = nullptr, *d_out = nullptr, *d_flags = nullptr;
  int rc = ensure_io(ctx, IO_K, n * pb, &d_a);
  if (!rc) rc = ensure_io(ctx, IO_P, n * pb, &d_b);
  if (!rc && a_inf) rc = ensure_io(ctx, IO_A, n, &d_ai);
  if (!rc && b_inf) rc = ensure_io(ctx, IO_B, n, &d_bi);
  if (!rc) rc = ensure_io(ctx, IO_O, n * pb, &d_out);
  if (!rc) rc = ensure_io(ctx, IO_F, n, &d_flags);
  if (rc) return rc;
  HIP_TRY(ctx, hipMemcpyAsync(d_a, a, n * pb, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(d_b, b, n * pb, hipMemcpyHostToDevice, ctx->stream));
  if (a_inf) HIP_TRY(ctx, hipMemcpyAsync(d_ai, a_inf, n, hipMemcpyHostToDevice, ctx->stream));
  if (b_inf) HIP_TRY(ctx, hipMemcpyAsync(d_bi, b_inf, n, hipMemcpyHostToDevice, ctx->stream));
  rc = eccx_point_add_dev(ctx, curve, n, d_a, d_ai, d_b, d_bi, d_out, d_flags, opts, ctx->stream);
  if (rc) { (void)hipStreamSynchronize(ctx->stream); return rc; }
  HIP_TRY(ctx, hipMemcpyAsync(out, d_out, n * pb, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(flags, d_flags, n, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return ECCX_OK;
}

int eccx_compressed_bytes(int curve) {
  const CurveOps* ops = ops_of(curve);
  return ops ? ops->enc_bytes : ECCX_ERR_CURVE;
}

int eccx_point_decompress_dev(eccx_ctx* ctx, int curve, size_t n, const void* d_enc, void* d_out, void* d_flags,
                              uint32_t opts, void* stream) {
  const CurveOps* ops = ops_of(curve);
  if (!ctx) return ECCX_ERR_ARG;
  if (!ops) return curve_err(ctx);
  if (n == 0) return ECCX_OK;
  if (!d_enc || !d_out || !d_flags) return arg_err(ctx, "null buffer");
  // the sec2 curves have cofactor 1 (nothing to check); decode_point makes no such test
  if ((opts & ECCX_CHECK_SUBGROUP) && ops->info.edwards) return arg_err(ctx, "ECCX_CHECK_SUBGROUP: edwards25519 decoding makes no subgroup test");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t s = static_cast<hipStream_t>(stream);
  uint8_t* out = static_cast<uint8_t*>(d_out);
  uint8_t* flags = static_cast<uint8_t*>(d_flags);
  if (opts & ECCX_UNCOMPRESSED) {
    if (!ops->decompress_raw) return arg_err(ctx, "ECCX_UNCOMPRESSED: the flavour exists for bls12_381_g1 only");
    HIP_TRY(ctx, ops->decompress_raw(flat_grid(ctx, n), s, n, static_cast<const uint8_t*>(d_enc), out, flags));
  } else {
    HIP_TRY(ctx, ops->decompress(flat_grid(ctx, n), s, n, static_cast<const uint8_t*>(d_enc), out, flags));
  }
  if ((opts & ECCX_CHECK_SUBGROUP) && ops->subgroup_check) {
    // the reference's endomorphism test sigma(P) == [-x^2]P (g1.rs:90-109), in place on the decoded
    // points: no temporaries, no synchronisation
    HIP_TRY(ctx, ops->subgroup_check(flat_grid(ctx, n), s, n, out, flags));
  }
  return ECCX_OK;
}

int eccx_point_compress_dev(eccx_ctx* ctx, int curve, size_t n, const void* d_xy, const void* d_inf, void* d_out,
                            uint32_t opts, void* stream) {
  const CurveOps* ops = ops_of(curve);
  if (!ctx) return ECCX_ERR_ARG;
  if (!ops) return curve_err(ctx);
  if (n == 0) return ECCX_OK;
  if (!d_xy || !d_out) return arg_err(ctx, "null buffer");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if ((opts & ECCX_UNCOMPRESSED) && !ops->compress_raw) return arg_err(ctx, "ECCX_UNCOMPRESSED: the flavour exists for bls12_381_g1 only");
  HIP_TRY(ctx, ((opts & ECCX_UNCOMPRESSED) ? ops->compress_raw : ops->compress)(
                   flat_grid(ctx, n), static_cast<hipStream_t>(stream), n, static_cast<const uint8_t*>(d_xy),
                   static_cast<const uint8_t*>(d_inf), static_cast<uint8_t*>(d_out)));
  return ECCX_OK;
}

int eccx_point_decompress(eccx_ctx* ctx, int curve, size_t n, const uint8_t* enc, uint8_t* out, uint8_t* flags,
                          uint32_t opts) {
  const CurveOps* ops = ops_of(curve);
  if (!ctx) return ECCX_ERR_ARG;
  if (!ops) return curve_err(ctx);
  if (n == 0) return ECCX_OK;
  if (!enc || !out || !flags) return arg_err(ctx, "null buffer");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const size_t pb = 2 * (size_t)ops->info.fb, eb = (opts & ECCX_UNCOMPRESSED) ? pb : (size_t)ops->enc_bytes;
  uint8_t *d_enc = nullptr, *d_out = nullptr, *d_flags = nullptr;
  int rc = ensure_io(ctx, IO_K, n * eb, &d_enc);
  if (!rc) rc = ensure_io(ctx, IO_O, n * pb, &d_out);
  if (!rc) rc = ensure_io(ctx, IO_F, n, &d_flags);
  if (rc) return rc;
  HIP_TRY(ctx, hipMemcpyAsync(d_enc, enc, n * eb, hipMemcpyHostToDevice, ctx->stream));
  rc = eccx_point_decompress_dev(ctx, curve, n, d_enc, d_out, d_flags, opts, ctx->stream);
  if (rc) { (void)hipStreamSynchronize(ctx->stream); return rc; }
  HIP_TRY(ctx, hipMemcpyAsync(out, d_out, n * pb, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(flags, d_flags, n, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return ECCX_OK;
}

int eccx_point_compress(eccx_ctx* ctx, int curve, size_t n, const uint8_t* xy, const uint8_t* inf, uint8_t* out,
                        uint32_t opts) {
  const CurveOps* ops = ops_of(curve);
  if (!ctx) return ECCX_ERR_ARG;
  if (!ops) return curve_err(ctx);
  if (n == 0) return ECCX_OK;
  if (!xy || !out) return arg_err(ctx, "null buffer");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const size_t pb = 2 * (size_t)ops->info.fb, eb = (opts & ECCX_UNCOMPRESSED) ? pb : (size_t)ops->enc_bytes;
  uint8_t *d_xy = nullptr, *d_inf = nullptr, *d_out = nullptr;
  int rc = ensure_io(ctx, IO_P, n * pb, &d_xy);
  if (!rc && inf) rc = ensure_io(ctx, IO_A, n, &d_inf);
  if (!rc) rc = ensure_io(ctx, IO_O, n * eb, &d_out);
  if (rc) return rc;
  HIP_TRY(ctx, hipMemcpyAsync(d_xy, xy, n * pb, hipMemcpyHostToDevice, ctx->stream));
  if (inf) HIP_TRY(ctx, hipMemcpyAsync(d_inf, inf, n, hipMemcpyHostToDevice, ctx->stream));
  rc = eccx_point_compress_dev(ctx, curve, n, d_xy, d_inf, d_out, opts, ctx->stream);
  if (rc) { (void)hipStreamSynchronize(ctx->stream); return rc; }
  HIP_TRY(ctx, hipMemcpyAsync(out, d_out, n * eb, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return ECCX_OK;
}

int eccx_double_scalarmul_dev(eccx_ctx* ctx, int curve, size_t n, const void* d_u1, const void* d_u2, const void* d_q,
                              void* d_out, void* d_flags, uint32_t opts, void* stream) {
  const CurveOps* ops = ops_of(curve);
  if (!ctx) return ECCX_ERR_ARG;
  if (!ops) return curve_err(ctx);
  if (n == 0) return ECCX_OK;
  if (!d_u1 || !d_u2 || !d_q || !d_out || !d_flags) return arg_err(ctx, "null buffer");
  if (!ops->var_fused || !ops->to_affine_var) {
    ctx->set_err("eccx_double_scalarmul: no fused kernel for this curve");
    return ECCX_ERR_ARG;
  }
  if ((opts & ECCX_OUT_X_ONLY) && !ops->to_affine_x)
    return arg_err(ctx, "ECCX_OUT_X_ONLY: Weierstrass curves only (Ed25519 verification compares encoded points)");
  if (opts & ECCX_CT_SCAN) {  // the verify shape works on public data; no scanning form
    ctx->set_err("eccx_double_scalarmul: ECCX_CT_SCAN is not accepted (signature verification handles public data)");
    return ECCX_ERR_ARG;
  }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  // one kernel: the ladder for u2*Q, then the 16-bit comb of u1*G onto the same point
  hipStream_t s = static_cast<hipStream_t>(stream);  // NULL = HIP's default stream
  int rc = ensure_comb(ctx, curve, ops, s);
  if (rc) return rc;
  if (!ctx->comb_u[curve]) return ECCX_ERR_HIP;
  const int grid = ops->var_fast_grid ? ops->var_fast_grid(ctx->cus, n) : grid_for(ctx, n);
  rc = ensure_scratch(ctx, ops->info.row5_words, grid);
  if (rc) return rc;
  rc = ensure_rows(ctx, ops, n);
  if (rc) return rc;
  const uint32_t kopts = kopts_of(opts) | ((opts & ECCX_SUBTRACT) ? K_NEGATE_B : 0u);
  if (ops->var_coz_fused) {
    // Weierstrass curves: the ladder over an affine window table (kernels_coz.hpp), then the generic fused
    // kernel for the units it marked (none unless a base has order <= 16 or is not a curve point)
    const int gridc = ops->var_coz_fused_grid(ctx->cus, n);
    rc = ensure_scratch(ctx, ops->coz_row_words, gridc);
    if (rc) return rc;
    HIP_TRY(ctx, ops->var_coz_fused(gridc, s, n, static_cast<const uint8_t*>(d_u2), static_cast<const uint8_t*>(d_q), ctx->jac,
                                    static_cast<uint8_t*>(d_flags), ctx->scratch, kopts, static_cast<const uint8_t*>(d_u1),
                                    ctx->comb_u[curve]));
  }
  HIP_TRY(ctx, ops->var_fused(grid, s, n, static_cast<const uint8_t*>(d_u2), static_cast<const uint8_t*>(d_q), ctx->jac,
                              static_cast<uint8_t*>(d_flags), ctx->scratch, kopts | (ops->var_coz_fused ? K_ONLY_MARKED : 0u),
                              static_cast<const uint8_t*>(d_u1), ctx->comb_u[curve]));
  // ECCX_OUT_X_ONLY: the x-coordinate alone (what ECDSA verification compares with r): FB bytes per unit, one product less
  HIP_TRY(ctx, ((opts & ECCX_OUT_X_ONLY) ? ops->to_affine_x : ops->to_affine_var)(norm_grid(ctx, n), s, n, ctx->jac,
                                                                                  static_cast<uint8_t*>(d_out),
                                                                                  static_cast<uint8_t*>(d_flags)));
  return ECCX_OK;
}

int eccx_double_scalarmul(eccx_ctx* ctx, int curve, size_t n, const uint8_t* u1, const uint8_t* u2,
                          const uint8_t* q, uint8_t* out, uint8_t* flags, uint32_t opts) {
  const CurveOps* ops = ops_of(curve);
  if (!ctx) return ECCX_ERR_ARG;
  if (!ops) return curve_err(ctx);
  if (n == 0) return ECCX_OK;
  if (!u1 || !u2 || !q || !out || !flags) return arg_err(ctx, "null buffer");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const size_t pb = 2 * (size_t)ops->info.fb, sb = (size_t)ops->info.sb;
  uint8_t *d_u1 = nullptr, *d_u2 = nullptr, *d_q = nullptr, *d_o = nullptr, *d_f = nullptr;
  int rc = ensure_io(ctx, IO_K, n * sb, &d_u2);
  if (!rc) rc = ensure_io(ctx, IO_J, n * sb, &d_u1);
  if (!rc) rc = ensure_io(ctx, IO_P, n * pb, &d_q);
  if (!rc) rc = ensure_io(ctx, IO_O, n * pb, &d_o);
  if (!rc) rc = ensure_io(ctx, IO_F, n, &d_f);
  if (rc) return rc;
  const size_t ob = (opts & ECCX_OUT_X_ONLY) ? pb / 2 : pb;
  const HostIn ins[3] = {{d_u1, u1, sb}, {d_u2, u2, sb}, {d_q, q, pb}};
  const HostOut outs[2] = {{out, d_o, ob}, {flags, d_f, 1}};
  return host_pipeline(ctx, n, ins, 3, outs, 2, /*chunked=*/true, [&](size_t lo, size_t cnt) {
    return eccx_double_scalarmul_dev(ctx, curve, cnt, d_u1 + lo * sb, d_u2 + lo * sb, d_q + lo * pb, d_o + lo * ob, d_f + lo, opts,
                                     ctx->stream);
  });
}

int eccx_x25519_dev(eccx_ctx* ctx, size_t n, const void* d_scalars, const void* d_u, void* d_out, void* d_flags,
                    uint32_t opts, void* stream) {
  if (!ctx) return ECCX_ERR_ARG;
  if (n == 0) return ECCX_OK;
  if (!d_scalars || !d_out || !d_flags) return arg_err(ctx, "null buffer");
  const CurveOps* ops = ops_of(ECCX_ED25519);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t s = static_cast<hipStream_t>(stream);
  int rc = ensure_rows(ctx, ops, n);
  if (rc) return rc;
  size_t need = (n + eccx::LAUNCH_WG - 1) / eccx::LAUNCH_WG;
  int grid = (int)std::max<size_t>(1, std::min(need, (size_t)ctx->cus * 8));
  const uint32_t kopts = (opts & ECCX_X25519_RAW_LADDER) ? 0u : (1u << 4);  // OPT_X25519_RFC
  HIP_TRY(ctx, eccx::launch_x25519_ladder(grid, s, n, static_cast<const uint8_t*>(d_scalars),
                                          static_cast<const uint8_t*>(d_u), ctx->jac, static_cast<uint8_t*>(d_flags),
                                          kopts));
  HIP_TRY(ctx, eccx::launch_x25519_to_u(norm_grid(ctx, n), s, n, ctx->jac, static_cast<uint8_t*>(d_out),
                                        static_cast<uint8_t*>(d_flags)));
  return ECCX_OK;
}

int eccx_x25519(eccx_ctx* ctx, size_t n, const uint8_t* scalars, const uint8_t* u, uint8_t* out, uint8_t* flags,
                uint32_t opts) {
  if (!ctx) return ECCX_ERR_ARG;
  if (n == 0) return ECCX_OK;
  if (!scalars || !out || !flags) return arg_err(ctx, "null buffer");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  uint8_t *d_k = nullptr, *d_u = nullptr, *d_o = nullptr, *d_f = nullptr;
  int rc = ensure_io(ctx, IO_K, n * 32, &d_k);
  if (!rc) rc = ensure_io(ctx, IO_O, n * 32, &d_o);
  if (!rc) rc = ensure_io(ctx, IO_F, n, &d_f);
  if (!rc && u) rc = ensure_io(ctx, IO_P, n * 32, &d_u);
  if (rc) return rc;
  const HostIn ins[2] = {{d_k, scalars, 32}, {d_u, u, 32}};
  const HostOut outs[2] = {{out, d_o, 32}, {flags, d_f, 1}};
  return host_pipeline(ctx, n, ins, 2, outs, 2, /*chunked=*/true, [&](size_t lo, size_t cnt) {
    return eccx_x25519_dev(ctx, cnt, d_k + lo * 32, d_u ? d_u + lo * 32 : nullptr, d_o + lo * 32, d_f + lo, opts, ctx->stream);
  });
}

int eccx_comb_table(eccx_ctx* ctx, int curve, uint8_t* out) {
  const CurveOps* ops = ops_of(curve);
  if (!ctx || !out) return ECCX_ERR_ARG;
  if (!ops) return curve_err(ctx);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int nw = 2 * ops->info.sb;
  size_t rows = (size_t)nw * 16, pb = 2 * (size_t)ops->info.fb;
  std::vector<uint8_t> k = comb_scalars(ops);
  std::vector<uint8_t> aff(rows * pb), fl(rows);
  DevMem mem;
  uint8_t *d_k = nullptr, *d_o = nullptr, *d_f = nullptr;
  HIP_TRY(ctx, mem.alloc(&d_k, k.size()));
  HIP_TRY(ctx, mem.alloc(&d_o, aff.size()));
  HIP_TRY(ctx, mem.alloc(&d_f, rows));
  HIP_TRY(ctx, hipMemcpyAsync(d_k, k.data(), k.size(), hipMemcpyHostToDevice, ctx->stream));
  int rc = launch_var(ctx, ops, rows, d_k, nullptr, d_o, d_f, nullptr, K_BASE_IS_GENERATOR, false, ctx->stream);
  if (rc) return rc;
  HIP_TRY(ctx, hipMemcpyAsync(aff.data(), d_o, aff.size(), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  for (int w = 0; w < nw; ++w)
    for (int d = 1; d < 16; ++d)
      std::memcpy(out + ((size_t)w * 15 + (size_t)(d - 1)) * pb, aff.data() + ((size_t)w * 16 + (size_t)d) * pb, pb);
  return ECCX_OK;
}

int eccx_scalarmul_var_sharded(eccx_ctx** ctxs, int nctx, int curve, size_t n, const uint8_t* scalars,
                               const uint8_t* points, uint8_t* out, uint8_t* flags, uint32_t opts) {
  if (n && (!scalars || !points || !out || !flags)) return ECCX_ERR_ARG;
  return run_sharded(ctxs, nctx, curve, false, n, scalars, points, out, flags, opts);
}

int eccx_scalarmul_base_sharded(eccx_ctx** ctxs, int nctx, int curve, size_t n, const uint8_t* scalars, uint8_t* out,
                                uint8_t* flags, uint32_t opts) {
  if (n && (!scalars || !out || !flags)) return ECCX_ERR_ARG;
  return run_sharded(ctxs, nctx, curve, true, n, scalars, nullptr, out, flags, opts);
}

}  // extern "C"

// slq.hip — C-ABI implementation (include/slq.h) over the kernels in slq_kernels.hpp.
// Host logic only: handles, workspace, launch sequencing, HIP-event profiling.
// Built with: hipcc --offload-arch=gfx950 -O3 -shared -fPIC (see __graft_entry__.build()).
#include "../../include/slq.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <atomic>
#include <mutex>
#include <new>
#include <system_error>
#include <thread>
#include <type_traits>
#include <vector>

#include "slq_kernels.hpp"
#include "slq_ring_api.h"
#include "slq_ring.hpp"  // (RingGeo constants: no kernel of it is instantiated here)
#include "slq_build.hpp"  // an operator's derived data built on the device

using namespace slq;

// ---------------------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------------------
static thread_local char g_err[1024] = "";

static int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess)                                                                      \
      return fail(_e == hipErrorOutOfMemory ? SLQ_ENOMEM : SLQ_EHIP, "%s failed: %s (%s:%d)",  \
                  #expr, hipGetErrorString(_e), __FILE__, __LINE__);                           \
  } while (0)

#define SLQ_TRY(expr)           \
  do {                          \
    int _rc = (expr);           \
    if (_rc != SLQ_OK) return _rc; \
  } while (0)

extern "C" const char *slq_last_error(void) { return g_err; }
extern "C" int slq_version(void) { return SLQ_VERSION; }

// ---------------------------------------------------------------------------------------------------
// handles
// ---------------------------------------------------------------------------------------------------
struct slq_context {
  int device;
  hipStream_t stream;
  bool own_stream;
  int num_cus;
  // Operators, plans, diag accumulators and device matrices hold the context they were created on. A context
  // destroyed while such objects are alive (a garbage collector tears handles down in no particular order) only
  // becomes unusable for NEW objects; its stream lives until the last dependant has been destroyed.
  int refs = 0;
  bool dead = false;
  // pinned staging ring for host -> device uploads of probes (slq_plan_set_probes): two buffers, so that the host-side copy
  // of one chunk runs while the previous chunk is on its way over PCIe
  void *pin[2] = {nullptr, nullptr};
  hipEvent_t pin_ev[2] = {nullptr, nullptr};
  bool pin_busy[2] = {false, false};
  size_t pin_bytes = 0;
};
static void ctx_retain(slq_context *ctx) { ++ctx->refs; }
static void ctx_free(slq_context *ctx) {
  hipSetDevice(ctx->device);
  for (int b = 0; b < 2; ++b) {
    if (ctx->pin[b]) hipHostFree(ctx->pin[b]);
    if (ctx->pin_ev[b]) hipEventDestroy(ctx->pin_ev[b]);
  }
  if (ctx->own_stream) hipStreamDestroy(ctx->stream);
  delete ctx;
}
static void ctx_release(slq_context *ctx) {
  if (--ctx->refs == 0 && ctx->dead) ctx_free(ctx);
}

enum { OP_CSR = 0, OP_DENSE = 1, OP_CALLBACK = 2, OP_DEVICE_CALLBACK = 3, OP_GRAM = 4 };

struct slq_operator {
  slq_context *ctx;
  int kind, dtype;
  int64_t n, nnz;
  int32_t *rowptr, *colind;
  void *vals;   // CSR values or dense matrix (device)
  int64_t lda;
  bool owns;
  slq_matvec_fn fn;
  void *user;
  int32_t *perm_d;               // device: stored row i = caller row perm[i]; null if not reordered
  std::vector<int32_t> *perm_h;  // host copy (diag un-permutation)
  TileMeta tiles;                // workgroup LDS tiles of the fused passes (tile_ptr == null: none; SLQ_TILES)
  bool tiles_ringed = false;     // ... built to the caps of k_csr_ring_pass (SLQ_TILES=2), which reads these two:
  int32_t *inv_perm_d = nullptr; // device: caller row r is stored row inv_perm[r] (null if not reordered)
  int32_t *tile_desc = nullptr;  // 64 words per tile
  char *tile_rec = nullptr;      // the tiles' CSR records
  int32_t *tile_desc_u = nullptr;  // the same over the upper triangle (exactly symmetric operators): the alpha-only pass
  char *tile_rec_u = nullptr;
  int32_t xcd_tile_u[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // tile ranges of the upper-triangle stream, whose tiles are runs of the base tiles (regroup_upper_tiles, r04)
  size_t tile_desc_bytes = 0, tile_rec_bytes = 0, tile_desc_u_bytes = 0, tile_rec_u_bytes = 0;  // (what SLQ_DEVICE_BUILD=2 compares)
  int tile_max_lines_u = 0;        // longest line list of a tile in that stream (short lists: a ring geometry with one slot more)
  bool tile_u_padded = false;      // ... with every row's entries padded to a multiple of four (build_ring_stream: pad_rows)
  double upper_per_row = 0.0;      // distinct panel rows per row that stream lands (what decides whether the alpha-only pass takes it)
  // narrow panels (slq_ring.hpp): R = 2, 4 consecutive tiles merged into one, built the first time a plan asks for them
  // (ensure_ring_stream); [0] R = 2, [1] R = 4; *_u over the upper triangle where the operator has that stream
  struct MergedStream {
    bool u_padded = false;  // the upper stream's rows are padded to whole chunks (build_ring_stream: pad_rows)
    int32_t *desc = nullptr, *desc_u = nullptr;
    char *rec = nullptr, *rec_u = nullptr;
    int32_t xcd_tile[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    int max_lines = 0, max_lines_u = 0;  // longest line lists (full rows / upper triangle)
    bool tried = false;
  } merged[2];
  std::mutex *merged_lock = nullptr;
  // the update pass with the next step's alpha dot fused in (slq_ring_fa.hpp; built the first time a plan asks: ensure_fa_stream):
  // the upper-triangle stream WITHOUT the entries that cross from one XCD chunk into the next, and those entries as a list
  // (row, column, doubled value) for k_alpha_edges
  struct FaStream {
    int32_t *desc = nullptr;
    char *rec = nullptr;
    int32_t *er = nullptr, *ec = nullptr;
    void *ev = nullptr;
    int nedges = 0;
    bool tried = false;
  } fa;
  // exactly symmetric CSR only: upper triangle (diagonal + 2x strict upper) for the alpha pass, whose
  // q^T A q = sum_i q_i (a_ii q_i + 2 sum_{j>i} a_ij q_j) then gathers half the panel rows (null: none)
  int32_t *rowptr_u = nullptr, *colind_u = nullptr;
  void *vals_u = nullptr;
  slq_matmat_device_fn dev_fn = nullptr;  // OP_DEVICE_CALLBACK: Y = A X on device buffers, enqueued on our stream
  double rms_dist = -1.0;  // rms |i - j| over the stored nonzeros inside an XCD chunk (-1: unknown)
  int64_t nnz_u = 0;       // entries of the upper-triangle copy
  double far_per_row = 0.0;  // stored nonzeros per row with |i - j| > 4096 (0 when unknown: device-resident CSR)
  // OP_GRAM (x -> A^T (A x), A is mrows x n): rowptr/colind/vals hold A, the *_t arrays its transpose (n rows)
  int64_t mrows = 0;
  int32_t *rowptr_t = nullptr, *colind_t = nullptr;
  // affine CSR operator A + t B (slq_csr_affine_create): values of A and B on the union pattern (vals = va + t vb)
  void *vals_a = nullptr, *vals_b = nullptr;
  void *vals_t = nullptr;    // OP_DENSE, non-symmetric input only (also the values of A^T for OP_GRAM): the transpose, for the kernel that walks A by columns (null: A == A^T)
};

struct ProfEvent {
  hipEvent_t a, b;
  int kind;
};

// Every run-time switch of the launch sequence, read from the environment ONCE, when the plan is
// created: a plan never changes its behaviour afterwards, and key() is part of what identifies its captured hipGraph.
struct Switches {
  int fused;       // SLQ_FUSED     1: recompute-SpMM passes where they pay, 0: store-and-revisit sweeps, 2: passes always
  int nt;          // SLQ_NT        nontemporal hints on streamed-once rows
  int graph;       // SLQ_GRAPH     capture the k-step launch sequence into a hipGraph
  int mgs;         // SLQ_MGS       exact modified-Gram-Schmidt order
  int stored_u;    // SLQ_STORED_U  non-local operators: merged pass stores u, update pass reads it back
  int merged;      // SLQ_MERGED    alpha from the merged alpha+dots pass
  int cross;       // SLQ_CROSS     q_c.q_p from the update pass's cross term
  int tiles;       // SLQ_TILES     fused passes on the operator's LDS workgroup tiles (when it has them)
  int ring_alpha;  // SLQ_RING_ALPHA the alpha-only pass of a tiled symmetric operator: 2 ring over the upper-triangle stream, 1 ring over
                   //               the full rows, 0 the generic upper-triangle pass
  int ring_rev;    // SLQ_RING_REV  the ring-fed update pass sweeps panels and tiles in reverse (it starts where the dots pass ended)
  int dense_mfma;  // SLQ_DENSE_MFMA dense operator on the matrix cores (fp64 and fp32)
  int dense_tile16;  // SLQ_DENSE_TILE16 keep the 16-row dense kernel also for wide panels (A/B runs)
  int dense_lds;     // SLQ_DENSE_LDS  fp64 dense product with the operands staged in LDS (k_dense_mfma_lds; 0: k_dense_mfma_tile)
  int pipe;        // SLQ_PIPE      pipelined row loop in the dots/update passes (-1: by operator, slq_plan_create)
  int ring32;      // SLQ_RING32    opt-in: finished Lanczos vectors archived as fp32 for deep reorthogonalisation
  int fused_pad;   // SLQ_FUSED_LDS_PAD (-1: by row loop)
  int spmm_pad;    // SLQ_SPMM_LDS_PAD
  int fused_alpha; // SLQ_FUSED_ALPHA the ring-fed update pass of wide panels also takes the next step's alpha dot (slq_ring_fa.hpp; r04)
  int defer_axpy;  // SLQ_DEFER_AXPY the block-CGS sweeps apply `w -= cB W_c` in the update sweep: the dots sweeps are read-only (r04; 0: first chunk stores)
  unsigned key() const {
    unsigned k = 0;
    for (int v : {fused, nt, graph, mgs, stored_u, merged, cross, tiles, ring_alpha, ring_rev, dense_mfma, dense_tile16, dense_lds, pipe, ring32, fused_pad, spmm_pad, fused_alpha, defer_axpy})
      k = k * 1000003u + (unsigned)(v + 7);
    return k;
  }
};

struct slq_plan {
  slq_context *ctx;
  slq_operator *op;
  int dtype, n, nprobes, deg, orth, keep_basis;
  int LPR, PW, NP, bpad, S;
  size_t esz;
  int64_t slot_stride;  // elements between ring slots
  void *ring;
  void *T;              // product panel for dense / callback / Gram operators
  void *T2;             // Gram operator: the intermediate A W_c (mrows rows per panel)
  void *stage;          // column-major staging (probe upload, callback round trips)
  int stage_cols;
  StepState st;
  double *scal;         // one allocation behind all StepState arrays
  double *part;
  int nblkA, nblkS, nblkU, nblkT;  // grids: SpMM, streaming sweeps, fused dots/update passes, tiled passes
  int nblkF;                       // grid of the fused alpha pass
  size_t alpha_pad;                // LDS padding that caps its residency
  double *quad_d, *nodes_d, *weights_d;
  int *fail_d;
  int *ring_fail_d;
  int rmax;
  bool probes_ready, ran;
  int pdf_sphere;
  bool prof;
  std::vector<ProfEvent> events;
  std::vector<ProfEvent> pool;
  slq_profile acc;
  std::vector<char> hbuf;  // host staging for callback operators
  size_t bytes;
  int nstale;                 // > 0: the reorthogonalisation also sees nstale preloaded vectors t = -1 .. -nstale
  hipGraphExec_t graph_exec;  // the k-step launch sequence captured once per (plan, rtol, variant)
  double graph_rtol;
  unsigned graph_variant;
  Switches sw;
  bool pipelined;             // dots/update passes run the pipelined row loop (slq_plan_create)
  float *ring32;              // fp32 archive ring (ring32_on): slot t % S32 holds vector t as floats, same panel layout
  int S32;
  bool ring32_on;
  int dense_ks;               // dense MFMA operator with big tiles: K split over this many workgroups per row tile (0: 16-row kernel)
  // ring-fed tile passes (k_csr_ring_pass / k_ring_pass): ringR = panel rows per wave instruction of the tile stream the plan
  // uses (0: no tiles; 1: the tiles as clustered, 1-KiB panel rows; 2, 4: merged tiles, 512-B / 256-B panel rows)
  int ringR;
  bool ring_gen;              // every ring-fed pass of the plan runs k_ring_pass (always for ringR > 1; SLQ_RING_GEN for ringR = 1)
  bool ring_deep;             // steps with 4..8 ring columns run k_ring_pass with 8 waves (SLQ_RING_DEEP; else the generic passes)
  bool gram;                  // steps with 1..8 ring columns take their projections from Gram rows of the update passes (SLQ_GRAM)
  bool gram_csr;              // ... also where the plan's fused passes are the generic ones (k_csr_pass<PASS_UPDATEG>: narrow panels, small operators; SLQ_GRAM_CSR, r04)
  const int32_t *rs_desc, *rs_desc_u;  // the stream the plan's ring-fed passes read (full rows / upper triangle or null)
  const char *rs_rec, *rs_rec_u;
  bool rs_u_padded;            // ... whose rows are padded to whole chunks of four entries (the alpha-only pass's branch-free consumer)
  int32_t rs_xcd[9];
  int32_t rs_xcd_u[9];         // ... and of the upper-triangle stream (its tiles are runs of the base tiles)
  bool ring_staged;           // the alpha-only pass's loaders go through registers (SLQ_RING_STAGED; slq_ring.hpp: GEO 1)
  // the update pass takes the next step's alpha dot a fixed lag of tile rounds behind its write front (slq_ring_fa.hpp; SLQ_FUSED_ALPHA):
  bool fa_on = false;
  int *fa_cnt = nullptr;      // chunk counters [NP][8][fa_rounds] + the XCC table [8], zeroed at the start of every run
  int fa_rounds = 0;
  int part_maxblk = 0;        // blocks per slab of `part`
  bool launch_error = false;  // a launcher declined (mis-dispatch): the run is invalid (enqueue_run)
  bool sweep_skip = true;     // the update sweep does not read ring columns whose coefficient is zero for every probe of the panel (SLQ_SWEEP_SKIP=0: reads them all)
  unsigned long long *sweep_cols_d = nullptr;  // {ring columns the update sweeps read, columns they were offered}, summed over launches and panels (slq_plan_sweep_columns)
  bool last_nostore = true;   // the update pass of a run's last step does not store W_deg (plans without a kept basis; SLQ_LAST_STORE=1 stores)
};

// SLQ_TILES: 0 none, 1 workgroup tiles landed behind barriers (k_csr_tile_pass), 2 tiles fed through a ring of LDS slots by
// loader waves (k_csr_ring_pass). Read when an operator is created (the rows are regrouped into the tiles) and when a plan
// is created (whether its passes use them).
constexpr int kTilesDefault = 2;
// Opt-in (SLQ_FUSED_ALPHA=1). Built and parity-green in r04, and slower than the two passes it replaces: configs[1], 256 probes, orth 3: 2.5 ms per step even
// WITHOUT its counter polls (racy) against 1.36 + 0.40 ms; 5.9 ms with them (DESIGN.md §4.7 has the break-down and what it says about the ring).
constexpr int kFusedAlphaDefault = 0;
// set when a fused update + alpha pass found two XCDs behind one value of blockIdx.x % 8 (slq_ring_fa.hpp): later plans keep the
// separate alpha-only pass
static std::atomic<bool> g_fa_broken{false};
constexpr double kTileMaxColsPerRow = 4.5;      // tiles are kept when a tile row needs at most this many distinct panel rows
constexpr double kTileAlphaColsPerRow = 2.6;    // upper-triangle tiles: the alpha-only pass takes the ring up to this many landed rows per row (r03: 7-point grids too)
constexpr double kTileAlphaMergedColsPerRow = 2.6;  // ... and on the merged tiles of narrow panels up to this many (of the unmerged tiles)
constexpr double kTileLevelRows = 320.0;        // level sets the tile sweep's base order should not exceed (csr_create_impl)

static int env_int(const char *name, int dflt) {
  const char *s = getenv(name);
  return (s && *s) ? atoi(s) : dflt;
}
static int tiles_mode() { return env_int("SLQ_TILES", kTilesDefault); }

// Host-side work of an operator's creation (row orders, clusters, tile lists, streams) is cut into independent pieces -
// XCD chunks, tile ranges, row ranges - and run on a few threads: fn(piece, begin, end) over [0, count). Results never depend
// on the number of threads (every piece writes its own slots or its own buffer, joined in piece order). SLQ_HOST_THREADS
// overrides the default of min(16, hardware threads). Exceptions do not leave a worker: the first failure is reported.
static int host_threads() {
  const int hw = (int)std::thread::hardware_concurrency();
  return std::max(1, std::min(64, env_int("SLQ_HOST_THREADS", std::max(1, std::min(16, hw)))));
}
template <typename Fn> static bool parallel_pieces(int pieces, int64_t count, Fn fn) {
  pieces = (int)std::max<int64_t>(1, std::min<int64_t>(pieces, count));
  const int64_t per = (count + pieces - 1) / pieces;
  if (pieces == 1) {
    try { fn(0, (int64_t)0, count); } catch (...) { return false; }
    return true;
  }
  std::vector<char> ok((size_t)pieces, 1);
  std::vector<std::thread> th;
  th.reserve((size_t)pieces);
  for (int t = 0; t < pieces; ++t) {
    const int64_t b = std::min(count, t * per), e = std::min(count, b + per);
    try {
      th.emplace_back([&, t, b, e]() {
        try { fn(t, b, e); } catch (...) { ok[(size_t)t] = 0; }
      });
    } catch (...) {  // no thread to be had: do the piece here
      try { fn(t, b, e); } catch (...) { ok[(size_t)t] = 0; }
    }
  }
  for (auto &x : th) x.join();
  return std::all_of(ok.begin(), ok.end(), [](char c) { return c != 0; });
}

// wall time of the phases of an operator's creation, printed under SLQ_DEBUG (scripts/time_create.py)
struct PhaseClock {
  bool on = env_int("SLQ_DEBUG", 0) != 0;
  std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now(), t0 = t;
  void total(const char *what) {
    if (on) fprintf(stderr, "[slq] create: %-34s %7.1f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  }
  void lap(const char *what) {
    if (!on) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[slq] create: %-34s %7.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t).count());
    t = now;
  }
};

// Uninitialised host storage for the big arrays of an operator's creation (a std::vector zero-fills them on one thread,
// 20 ms per 100 MB, and every page is then touched a second time); whoever fills it writes every byte it will read.
template <typename T> struct RawBuf {
  std::unique_ptr<T[]> p;
  size_t n = 0;
  void alloc(size_t count) {
    p.reset(new T[count]);  // (default-initialised: no fill for arithmetic T)
    n = count;
  }
  T *data() { return p.get(); }
  const T *data() const { return p.get(); }
  size_t size() const { return n; }
};

// Host-to-device copies of an operator's arrays, run by helper threads while the caller builds the next arrays: a copy from
// pageable memory blocks its caller at ~10 GB/s, 40 of the 160 ms a 10^6-row operator took to create. Every source must
// outlive wait() (declare the queue AFTER the buffers it reads: its destructor joins first).
struct UploadQueue {
  struct Job { void *dst; const void *src; size_t bytes; };
  int device;
  std::vector<std::thread> th;
  std::mutex m;
  hipError_t err = hipSuccess;
  explicit UploadQueue(int dev) : device(dev) {}
  void push(std::vector<Job> jobs) {
    auto run = [this](const std::vector<Job> &js) {
      hipError_t e = hipSetDevice(device);
      for (const Job &j : js)
        if (e == hipSuccess && j.bytes) e = hipMemcpy(j.dst, j.src, j.bytes, hipMemcpyHostToDevice);
      if (e != hipSuccess) {
        std::lock_guard<std::mutex> g(m);
        if (err == hipSuccess) err = e;
      }
    };
    try {
      th.emplace_back([run, jobs]() { run(jobs); });
    } catch (const std::system_error &) {  // no thread to be had: the caller copies
      run(jobs);
    }
  }
  hipError_t wait() {
    for (auto &t : th)
      if (t.joinable()) t.join();
    th.clear();
    return err;
  }
  ~UploadQueue() { wait(); }
};

// ---------------------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------------------
extern "C" int slq_device_count(int *count) {
  if (!count) return fail(SLQ_EINVAL, "count is NULL");
  int c = 0;
  hipError_t e = hipGetDeviceCount(&c);
  if (e != hipSuccess) {
    *count = 0;
    return fail(SLQ_ENODEV, "hipGetDeviceCount: %s", hipGetErrorString(e));
  }
  *count = c;
  return SLQ_OK;
}

extern "C" int slq_context_create(int device, void *hip_stream, slq_context **out) {
  if (!out) return fail(SLQ_EINVAL, "out is NULL");
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
    return fail(SLQ_ENODEV, "no HIP device visible: the SLQ engine has no CPU fallback");
  if (device < 0) HIP_TRY(hipGetDevice(&device));
  if (device >= count) return fail(SLQ_EINVAL, "device %d out of range (%d visible)", device, count);
  HIP_TRY(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(SLQ_ENODEV, "device %d is %s; libslq is built for gfx950 only", device,
                prop.gcnArchName);
  slq_context *ctx = new (std::nothrow) slq_context();
  if (!ctx) return fail(SLQ_ENOMEM, "host allocation failed");
  ctx->device = device;
  ctx->num_cus = prop.multiProcessorCount;
  if (hip_stream) {
    ctx->stream = (hipStream_t)hip_stream;
    ctx->own_stream = false;
  } else {
    hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
      delete ctx;
      return fail(SLQ_EHIP, "hipStreamCreate: %s", hipGetErrorString(e));
    }
    ctx->own_stream = true;
  }
  *out = ctx;
  return SLQ_OK;
}

extern "C" int slq_context_destroy(slq_context *ctx) {
  if (!ctx) return SLQ_OK;
  ctx->dead = true;
  if (ctx->refs == 0) ctx_free(ctx);
  return SLQ_OK;
}

extern "C" int slq_context_synchronize(slq_context *ctx) {
  if (!ctx) return fail(SLQ_EINVAL, "ctx is NULL");
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return SLQ_OK;
}

extern "C" int slq_context_device(slq_context *ctx, int *device) {
  if (!ctx || !device) return fail(SLQ_EINVAL, "ctx/device is NULL");
  *device = ctx->device;
  return SLQ_OK;
}

extern "C" int slq_context_meminfo(slq_context *ctx, size_t *free_bytes, size_t *total_bytes) {
  if (!ctx) return fail(SLQ_EINVAL, "ctx is NULL");
  HIP_TRY(hipSetDevice(ctx->device));
  size_t f = 0, t = 0;
  HIP_TRY(hipMemGetInfo(&f, &t));
  if (free_bytes) *free_bytes = f;
  if (total_bytes) *total_bytes = t;
  return SLQ_OK;
}

// ---------------------------------------------------------------------------------------------------
// operators
// ---------------------------------------------------------------------------------------------------
static size_t esize(int dtype) { return dtype == SLQ_F64 ? 8 : 4; }

static int check_dtype(int dtype) {
  if (dtype != SLQ_F32 && dtype != SLQ_F64)
    return fail(SLQ_EINVAL, "Only 32- or 64-bit floats are supported.");
  return SLQ_OK;
}


// ---------------------------------------------------------------------------------------------------
// XCD-aware row reordering (speed only; results are permutation-invariant up to rounding)
// ---------------------------------------------------------------------------------------------------
// k_spmm_3term / k_csr_pass give XCD x the contiguous row range [x*n/8, (x+1)*n/8) and sweep it
// with all of the XCD's waves in lock-step, so a gathered panel row stays useful only while the
// sweep front is within the matrix bandwidth of it. With 1 KiB panel rows and a 4 MiB L2 the
// natural order of a 1000 x 1000 grid (bandwidth 1000 -> 2 MiB of halo) no longer fits beside the
// rows in flight: the alpha pass fetched 6.5 GB per launch against 4.2 GB algorithmic
// (profiles/r01b_pmc_per_kernel.csv). Reverse Cuthill-McKee INSIDE each XCD's chunk shrinks the
// bandwidth to the chunk's short dimension (125 for that grid; build/rcm_test in round 1). perm[new] = old.
// MEASURED RESULT: slower, see slq_csr_create. The L2 behaviour of this kernel is not explained by
// the reuse-distance model above (fewer resident workgroups also fetch MORE, not less).
// sub: second-level pieces per chunk (below). avg_level: if not null, receives the mean size of the breadth-first level sets of
// the final order - what a tile sweep has to keep in L2 between a row and its neighbours in the next level.
// first_level: the order a call with sub = 1 returned for this matrix (null: computed here) - the sweep over sub = 4, 16, 64 of
// csr_create_body does the chunks' own Cuthill-McKee once (r04: it was redone per attempt, 12 ms of a 100^3 operator's creation).
static void xcd_rcm_permutation(int64_t n, const int32_t *rowptr, const int32_t *colind, std::vector<int32_t> &perm, int sub,
                                double *avg_level, const std::vector<int32_t> *first_level = nullptr) {
  perm.resize((size_t)n);
  const int64_t chunk = (n + 7) / 8;
  // The eight chunks are independent: one worker each. deg / part / seen are indexed by node and a worker touches the
  // entries of its own chunk only (a neighbour's entry is read only after its index has been found inside the chunk).
  std::vector<int32_t> deg((size_t)n), part((size_t)n, -1);
  std::vector<char> seen((size_t)n, 0);
  sub = std::max(1, sub);
  int64_t levels_x[8] = {0, 0, 0, 0, 0, 0, 0, 0}, levelled_x[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  auto do_chunk = [&](int x) {
    const int64_t lo = x * chunk, hi = std::min<int64_t>(n, lo + chunk);
    if (lo >= hi) return;
    int64_t levels = 0, levelled = 0;  // of the committed searches since the last reset
    std::vector<int32_t> nbrs, order, members, first, piece, second;
    auto inside = [&](int32_t v, int32_t id) { return v >= lo && v < hi && part[(size_t)v] == id; };
    // Reverse Cuthill-McKee of the subgraph induced by `members` (all with part[v] == id), appended to `out`.
    auto rcm = [&](const std::vector<int32_t> &mem, int32_t id, std::vector<int32_t> &out) {
      for (int32_t v : mem) {
        int d = 0;
        for (int32_t p = rowptr[v]; p < rowptr[v + 1]; ++p) d += (colind[p] != v && inside(colind[p], id));
        deg[(size_t)v] = d;
      }
      order.clear();
      // candidates in increasing degree: starting points of the components
      std::vector<int32_t> cand(mem);
      std::stable_sort(cand.begin(), cand.end(), [&](int32_t a, int32_t b) { return deg[(size_t)a] < deg[(size_t)b]; });
      auto bfs = [&](int32_t start, bool commit, int32_t *last_min) {
        // breadth-first numbering with neighbours in increasing degree (Cuthill-McKee)
        const size_t base = order.size();
        order.push_back(start);
        seen[(size_t)start] = 1;
        size_t head = base, level_begin = base;
        while (head < order.size()) {
          const size_t level_end = order.size();
          level_begin = head;
          if (commit) {
            ++levels;
            levelled += (int64_t)(level_end - head);
          }
          for (; head < level_end; ++head) {
            const int32_t u = order[head];
            nbrs.clear();
            for (int32_t p = rowptr[u]; p < rowptr[u + 1]; ++p) {
              const int32_t v = colind[p];
              if (inside(v, id) && !seen[(size_t)v]) {
                seen[(size_t)v] = 1;
                nbrs.push_back(v);
              }
            }
            std::sort(nbrs.begin(), nbrs.end(), [&](int32_t a, int32_t b) { return deg[(size_t)a] < deg[(size_t)b] || (deg[(size_t)a] == deg[(size_t)b] && a < b); });
            order.insert(order.end(), nbrs.begin(), nbrs.end());
          }
        }
        // min-degree node of the last level: a pseudo-peripheral candidate
        int32_t far = order[level_begin];
        for (size_t q = level_begin; q < order.size(); ++q)
          if (deg[(size_t)order[q]] < deg[(size_t)far]) far = order[q];
        if (last_min) *last_min = far;
        if (!commit) {
          for (size_t q = base; q < order.size(); ++q) seen[(size_t)order[q]] = 0;
          order.resize(base);
        }
      };
      for (int32_t c : cand) {
        if (seen[(size_t)c]) continue;
        int32_t far = c;
        bfs(c, false, &far);      // one pseudo-peripheral refinement
        bfs(far, true, nullptr);
      }
      for (int32_t v : mem) seen[(size_t)v] = 0;
      out.insert(out.end(), order.rbegin(), order.rend());  // reversed (RCM)
    };
    if (first_level && sub > 1) {
      first.assign(first_level->begin() + lo, first_level->begin() + hi);
    } else {
      members.resize((size_t)(hi - lo));
      for (int64_t i = lo; i < hi; ++i) {
        members[(size_t)(i - lo)] = (int32_t)i;
        part[(size_t)i] = x;
      }
      rcm(members, x, first);
    }
    // Second level (SLQ_RCM_SUB = K > 1): the chunk's RCM order is cut into K consecutive pieces of equal size - runs of
    // BFS levels, i.e. slices ACROSS the chunk's longest direction - and each piece is reordered on its own. A piece is
    // short along the old sweep direction, so its own Cuthill-McKee levels run along another one and are K times
    // smaller: the gather halo an XCD's L2 has to hold shrinks accordingly, at the price of the edges cut between pieces.
    if (sub > 1 && (int64_t)first.size() >= 64 * sub) {
      levels = levelled = 0;
      const size_t len = (first.size() + sub - 1) / sub;
      for (int k = 0; k < sub; ++k) {
        const size_t b0 = std::min(first.size(), k * len), b1 = std::min(first.size(), b0 + len);
        piece.assign(first.begin() + b0, first.begin() + b1);
        const int32_t id = 8 + x * sub + k;
        for (int32_t v : piece) part[(size_t)v] = id;
        rcm(piece, id, second);
      }
      first.swap(second);
    }
    levels_x[x] = levels;
    levelled_x[x] = levelled;
    for (int64_t q = 0; q < hi - lo; ++q) perm[(size_t)(lo + q)] = first[(size_t)q];
  };
  if (host_threads() > 1) {
    if (!parallel_pieces(8, 8, [&](int, int64_t x0, int64_t x1) { for (int64_t x = x0; x < x1; ++x) do_chunk((int)x); })) throw std::bad_alloc();
  } else {
    for (int x = 0; x < 8; ++x) do_chunk(x);
  }
  int64_t levels_all = 0, levelled_all = 0;
  for (int x = 0; x < 8; ++x) levels_all += levels_x[x], levelled_all += levelled_x[x];
  if (avg_level) *avg_level = levels_all > 0 ? (double)levelled_all / (double)levels_all : 0.0;
}


// Workgroup tiles for k_csr_tile_pass (SLQ_TILES). The rows of every XCD chunk are regrouped into compact clusters:
// seeds are taken in the chunk's current order (natural or Cuthill-McKee), a cluster grows breadth-first by the
// unassigned in-chunk neighbour with the most links into it (ties: first discovered), up to kTileRows rows and as long
// as its rows and columns together stay within kTileCols distinct indices. Clusters follow one another in seed order,
// so the sweep of the chunk keeps its locality. order_in: stored row -> caller row; inv_in: caller row -> stored row
// (null: identity). order_out: the new stored order; tile_row: first stored row of every tile; xcd_tile: tile range of
// every chunk. Returns false when a single row already needs more than kTileCols indices (no tiling for this operator).
// *lines_total (if not null): the sum over the clusters of their distinct indices (rows and columns) - the panel rows a sweep of the tiles lands.
static bool build_clusters(int64_t n, const int32_t *rowptr, const int32_t *colind, const int32_t *order_in, const int32_t *inv_in,
                           std::vector<int32_t> &order_out, std::vector<int32_t> &tile_row, int32_t xcd_tile[9], int64_t *lines_total = nullptr) {
  const int64_t chunk = (n + 7) / 8;
  const bool ringed = tiles_mode() == 2;  // tiles of the ring-fed kernel (k_csr_ring_pass): smaller, fixed caps
  const int tmax = ringed ? kRingTileRows : std::max(1, std::min(env_int("SLQ_TILE_ROWS", kTileRows), 64));
  const int dcap = ringed ? kRingTileCols : std::max(8, std::min(env_int("SLQ_TILE_COLS", kTileCols), kTileCols));
  const int nzcap = ringed ? kRingTileNnz : std::numeric_limits<int>::max();  // the ring kernel's tile record is bounded
  // The chunks are independent (a cluster never leaves its chunk): one worker each, with its own order, its own tile
  // boundaries (counted from the chunk's first row) and its own stamp array; `assigned` is shared, but a worker reads and
  // writes the entries of its own chunk's rows only.
  // (r03: every chunk is clustered as kClusterPieces independent halves of its order - a fixed split, so the tiles do not
  // depend on the number of host threads - because the greedy growth is sequential and was 25-50 ms of an operator's creation
  // with one worker per chunk; a cluster never crosses the middle of a chunk either: one short tile per 60,000 rows.)
  constexpr int kClusterPieces = 2, NX = 8 * kClusterPieces;
  std::vector<char> assigned((size_t)n, 0);
  std::vector<int32_t> order_x[NX], rows_x[NX];  // per piece: the new order, and the row count of every cluster
  int64_t lines_x[NX] = {};
  char failed[NX] = {};
  auto do_chunk = [&](int x) {
    const int64_t clo = (x / kClusterPieces) * chunk, chi = std::min<int64_t>(n, clo + chunk);
    if (clo >= chi) return;
    const int64_t plen = (chi - clo + kClusterPieces - 1) / kClusterPieces;
    const int64_t lo = clo + (x % kClusterPieces) * plen, hi = std::min<int64_t>(chi, lo + plen);
    if (lo >= hi) return;
    std::vector<int32_t> stamp((size_t)n, -1);
    struct Cand { int32_t node, cnt, disc; };
    std::vector<Cand> cand;
    // where a node of this piece sits in `cand` while it is a candidate of the current cluster (r04: the list was searched linearly for every
    // neighbour of every added row - 20 of the 35 ms this took on a 100^3 grid); indexed by position in the piece
    std::vector<int32_t> slot_of((size_t)(hi - lo), -1);
    std::vector<int32_t> &order = order_x[x];
    order.reserve((size_t)(hi - lo));
    int32_t cid = 0;
    auto in_chunk = [&](int32_t v) {
      const int64_t b = inv_in ? inv_in[v] : v;
      return b >= lo && b < hi;
    };
    for (int64_t b = lo; b < hi; ++b) {
      const int32_t seed = order_in ? order_in[b] : (int32_t)b;
      if (assigned[(size_t)seed]) continue;
      int D = 0, ndisc = 0, nz = 0;
      cand.clear();
      const size_t first_member = order.size();
      auto new_cols = [&](int32_t v) {
        int c = stamp[(size_t)v] != cid;
        for (int32_t p = rowptr[v]; p < rowptr[v + 1]; ++p) c += (stamp[(size_t)colind[p]] != cid && colind[p] != v);
        return c;
      };
      auto add = [&](int32_t v) {
        assigned[(size_t)v] = 1;
        order.push_back(v);
        nz += rowptr[v + 1] - rowptr[v];
        if (stamp[(size_t)v] != cid) { stamp[(size_t)v] = cid; ++D; }
        for (int32_t p = rowptr[v]; p < rowptr[v + 1]; ++p) {
          const int32_t c = colind[p];
          if (stamp[(size_t)c] != cid) { stamp[(size_t)c] = cid; ++D; }
          if (c != v && in_chunk(c) && !assigned[(size_t)c]) {
            int32_t &slot = slot_of[(size_t)((inv_in ? inv_in[c] : c) - lo)];
            if (slot >= 0) ++cand[(size_t)slot].cnt;
            else slot = (int32_t)cand.size(), cand.push_back(Cand{c, 1, ndisc++});
          }
        }
      };
      if (new_cols(seed) > dcap || rowptr[seed + 1] - rowptr[seed] > nzcap) { failed[x] = 1; return; }
      add(seed);
      while ((int)(order.size() - first_member) < tmax && !cand.empty()) {
        size_t best = 0;
        for (size_t q = 1; q < cand.size(); ++q)
          if (cand[q].cnt > cand[best].cnt || (cand[q].cnt == cand[best].cnt && cand[q].disc < cand[best].disc)) best = q;
        const int32_t v = cand[best].node;
        slot_of[(size_t)((inv_in ? inv_in[v] : v) - lo)] = -1;
        cand[best] = cand.back();
        cand.pop_back();
        if (best < cand.size()) slot_of[(size_t)((inv_in ? inv_in[cand[best].node] : cand[best].node) - lo)] = (int32_t)best;
        if (assigned[(size_t)v]) continue;
        if (D + new_cols(v) > dcap || nz + rowptr[v + 1] - rowptr[v] > nzcap) continue;  // would not fit: leave it for a later cluster
        add(v);
      }
      for (const Cand &k : cand) slot_of[(size_t)((inv_in ? inv_in[k.node] : k.node) - lo)] = -1;  // (what the cluster leaves behind)
      lines_x[x] += D;
      rows_x[x].push_back((int32_t)(order.size() - first_member));
      ++cid;
    }
  };
  if (host_threads() > 1) {
    if (!parallel_pieces(NX, NX, [&](int, int64_t x0, int64_t x1) { for (int64_t x = x0; x < x1; ++x) do_chunk((int)x); })) return false;
  } else {
    for (int x = 0; x < NX; ++x) do_chunk(x);
  }
  order_out.clear();
  order_out.reserve((size_t)n);
  tile_row.assign(1, 0);
  for (int x = 0; x < NX; ++x) {
    if (failed[x]) return false;
    if (x % kClusterPieces == 0) xcd_tile[x / kClusterPieces] = (int32_t)tile_row.size() - 1;
    order_out.insert(order_out.end(), order_x[x].begin(), order_x[x].end());
    for (int32_t r : rows_x[x]) tile_row.push_back(tile_row.back() + r);
  }
  xcd_tile[8] = (int32_t)tile_row.size() - 1;
  for (int x = 7; x >= 0; --x) xcd_tile[x] = std::min(xcd_tile[x], xcd_tile[x + 1]);
  if (lines_total) {
    *lines_total = 0;
    for (int x = 0; x < NX; ++x) *lines_total += lines_x[x];
  }
  return (int64_t)order_out.size() == n;
}

// A cheap look before the expensive one: grow one cluster from each of 256 evenly spaced seeds with build_clusters' rule (most
// links first, same caps) on the caller's numbering, and return the distinct panel rows per tile row of that sample. Operators
// whose rows share nothing (random graphs, bands with scattered far entries) show it here, in microseconds, and are spared the
// reorderings and the full clustering (tens of seconds at n = 10^7).
static double sample_tile_quality(int64_t n, const int32_t *rowptr, const int32_t *colind, int tmax, int dcap, int nzcap) {
  const int64_t chunk = (n + 7) / 8;
  int64_t rows = 0, cols = 0;
  std::vector<int32_t> members, seen;
  struct Cand { int32_t node, cnt; };
  std::vector<Cand> cand;
  for (int sidx = 0; sidx < 256; ++sidx) {
    const int32_t seed = (int32_t)(((int64_t)sidx * n) / 256);
    const int64_t lo = (seed / chunk) * chunk, hi = std::min<int64_t>(n, lo + chunk);
    members.clear();
    seen.clear();
    cand.clear();
    int nz = 0;
    auto is_in = [](const std::vector<int32_t> &v, int32_t x) { return std::find(v.begin(), v.end(), x) != v.end(); };
    auto new_cols = [&](int32_t v) {
      int c = !is_in(seen, v);
      for (int32_t p = rowptr[v]; p < rowptr[v + 1]; ++p) c += (colind[p] != v && !is_in(seen, colind[p]));
      return c;
    };
    auto add = [&](int32_t v) {
      members.push_back(v);
      nz += rowptr[v + 1] - rowptr[v];
      if (!is_in(seen, v)) seen.push_back(v);
      for (int32_t p = rowptr[v]; p < rowptr[v + 1]; ++p) {
        const int32_t c = colind[p];
        if (!is_in(seen, c)) seen.push_back(c);
        if (c != v && c >= lo && c < hi && !is_in(members, c)) {
          bool found = false;
          for (auto &k : cand) if (k.node == c) { ++k.cnt; found = true; break; }
          if (!found) cand.push_back(Cand{c, 1});
        }
      }
    };
    if (new_cols(seed) > dcap || rowptr[seed + 1] - rowptr[seed] > nzcap) return 1e9;
    add(seed);
    while ((int)members.size() < tmax && !cand.empty()) {
      size_t best = 0;
      for (size_t q = 1; q < cand.size(); ++q) if (cand[q].cnt > cand[best].cnt) best = q;
      const int32_t v = cand[best].node;
      cand[best] = cand.back();
      cand.pop_back();
      if (is_in(members, v)) continue;
      if ((int)seen.size() + new_cols(v) > dcap || nz + rowptr[v + 1] - rowptr[v] > nzcap) continue;
      add(v);
    }
    rows += (int64_t)members.size();
    cols += (int64_t)seen.size();
  }
  return rows > 0 ? (double)cols / (double)rows : 1e9;
}

// Tile lists of the STORED CSR: per tile the distinct indices of its rows and their columns (ascending unless SLQ_RING_ORDER
// says otherwise), per nonzero the position of its column in that list, per row the position of the row itself.
static void build_tile_meta(int64_t n, const int32_t *rowptr, const int32_t *colind, const std::vector<int32_t> &tile_row,
                            std::vector<int32_t> &tile_ptr, std::vector<int32_t> &tile_cols, std::vector<int32_t> &lcol,
                            std::vector<int32_t> &self_idx, int *max_cols) {
  const size_t ntiles = tile_row.size() - 1;
  tile_ptr.assign(ntiles + 1, 0);
  tile_cols.clear();
  lcol.assign((size_t)rowptr[n] + kCsrPad, 0);
  self_idx.assign((size_t)n, 0);
  const int line_order = env_int("SLQ_RING_ORDER", 0);
  const int pieces = host_threads();
  std::vector<std::vector<int32_t>> local((size_t)pieces);  // every piece's lists, in tile order
  std::vector<int> mx_piece((size_t)pieces, 0);
  const bool ok = parallel_pieces(pieces, (int64_t)ntiles, [&](int piece, int64_t t0, int64_t t1) {
    std::vector<int32_t> u, pos, ordered;
    std::vector<int32_t> &mine = local[(size_t)piece];
    int mx = 0;
    for (int64_t t = t0; t < t1; ++t) {
      const int64_t r0 = tile_row[(size_t)t], r1 = tile_row[(size_t)t + 1];
      u.clear();
      for (int64_t r = r0; r < r1; ++r) {
        u.push_back((int32_t)r);
        for (int32_t p = rowptr[r]; p < rowptr[r + 1]; ++p) u.push_back(colind[p]);
      }
      std::sort(u.begin(), u.end());
      u.erase(std::unique(u.begin(), u.end()), u.end());
      mx = std::max(mx, (int)u.size());
      // position of every distinct index in the tile's list = the order its panel rows are landed in. Ascending by default;
      // line_order 1: the tile's own rows first, then the rows below them, then the rows above (experiments, SLQ_RING_ORDER)
      pos.resize(u.size());
      if (line_order == 0) {
        for (size_t q = 0; q < u.size(); ++q) pos[q] = (int32_t)q;
      } else {
        const size_t lo = (size_t)(std::lower_bound(u.begin(), u.end(), (int32_t)r0) - u.begin());
        const size_t own = (size_t)(r1 - r0);
        for (size_t q = 0; q < u.size(); ++q) pos[q] = (int32_t)(q < lo ? own + q : (q < lo + own ? q - lo : q));
      }
      ordered.resize(u.size());
      for (size_t q = 0; q < u.size(); ++q) ordered[(size_t)pos[q]] = u[q];
      for (int64_t r = r0; r < r1; ++r) {
        self_idx[(size_t)r] = pos[(size_t)(std::lower_bound(u.begin(), u.end(), (int32_t)r) - u.begin())];
        for (int32_t p = rowptr[r]; p < rowptr[r + 1]; ++p)
          lcol[(size_t)p] = pos[(size_t)(std::lower_bound(u.begin(), u.end(), colind[p]) - u.begin())];
      }
      mine.insert(mine.end(), ordered.begin(), ordered.end());
      tile_ptr[(size_t)t + 1] = (int32_t)ordered.size();  // (the list's length for now; offsets below)
    }
    mx_piece[(size_t)piece] = mx;
  });
  if (!ok) throw std::bad_alloc();
  for (size_t t = 0; t < ntiles; ++t) tile_ptr[t + 1] += tile_ptr[t];
  tile_cols.reserve((size_t)tile_ptr[ntiles] + kCsrPad);
  for (auto &v : local) tile_cols.insert(tile_cols.end(), v.begin(), v.end());
  tile_cols.insert(tile_cols.end(), kCsrPad, 0);
  *max_cols = *std::max_element(mx_piece.begin(), mx_piece.end());
}

// What the ring-fed passes read (SLQ_TILES=2; layouts in slq_kernels.hpp / slq_ring.hpp): per tile a descriptor of R blocks
// of 64 words and a record of its CSR in the tile's own numbering, every record at a 16-byte boundary of one blob that ends
// in a spare record's worth of zeros (a record is fetched in whole KiB). R = 1: the tiles as clustered (k_csr_ring_pass and
// k_ring_pass<LPR = 64>); R = 2, 4: tiles of R merged base tiles for panels of 64 / R lanes per row - block b of the
// descriptor lists the lines b, R + b, 2R + b, ... (the lines lane group b lands), the last one repeated to the end of its DMA.
template <typename F>
static void build_ring_stream(int R, const int32_t *rowptr, const F *vals, const std::vector<int32_t> &tile_row, const std::vector<int32_t> &tile_ptr,
                              const std::vector<int32_t> &tile_cols, const std::vector<int32_t> &lcol, const std::vector<int32_t> &self_idx,
                              RawBuf<int32_t> &desc, RawBuf<char> &rec, bool *pad_rows = nullptr) {
  const size_t ntiles = tile_row.size() - 1;
  const size_t dw = (size_t)64 * R, head_bytes = (size_t)kRecHeadBytes * R;
  const int valoff_w = 16 * R - 1, self_w = 16 * R;
  desc.alloc(ntiles * dw);  // (zeroed tile by tile below, by the thread that fills the tile)
  // *pad_rows (the alpha-only pass's upper-triangle streams): every row's entries padded to a multiple of four, at least four, with
  // {the row's own line, 0} - its consumer then reads a row's entries four at a time with aligned 16-byte LDS reads and
  // without a single per-entry condition (slq_ring.hpp: do_alpha_padded). Given up (*pad_rows = false) if some tile's record
  // would outgrow its slot.
  bool pad = pad_rows && *pad_rows;
  auto padded = [](int32_t cnt) { return std::max<int32_t>(4, (cnt + 3) / 4 * 4); };
  if (pad) {
    for (size_t t = 0; t < ntiles && pad; ++t) {
      size_t e = 0;
      for (int32_t r = tile_row[t]; r < tile_row[t + 1]; ++r) e += (size_t)padded(rowptr[r + 1] - rowptr[r]);
      if (head_bytes + e * (4 + sizeof(F)) > (size_t)((kRingRecStride * R + 1023) / 1024 * 1024)) pad = false;
    }
  }
  if (pad_rows) *pad_rows = pad;
  // where every record starts (its size follows from the tile's entry count alone), then the tiles in parallel
  std::vector<size_t> off(ntiles + 1, 0);
  for (size_t t = 0; t < ntiles; ++t) {
    const int32_t r0 = tile_row[t];
    size_t nz = (size_t)(rowptr[tile_row[t + 1]] - rowptr[r0]);
    if (pad) {
      nz = 0;
      for (int32_t r = r0; r < tile_row[t + 1]; ++r) nz += (size_t)padded(rowptr[r + 1] - rowptr[r]);
    }
    const size_t nzp = (nz + 3) / 4 * 4;
    off[t + 1] = off[t] + (head_bytes + nzp * 4 + nzp * sizeof(F) + 15) / 16 * 16;
  }
  rec.alloc(off[ntiles] + (size_t)kRingMetaBytes * R);
  memset(rec.data() + off[ntiles], 0, (size_t)kRingMetaBytes * R);  // the spare record behind the last one
  const bool ok = parallel_pieces(host_threads(), (int64_t)ntiles, [&](int, int64_t t0, int64_t t1) {
    for (int64_t tt = t0; tt < t1; ++tt) {
      const size_t t = (size_t)tt;
      const int32_t r0 = tile_row[t], rows = tile_row[t + 1] - r0, p0 = rowptr[r0];
      int32_t nz = rowptr[r0 + rows] - p0;
      if (pad) {
        nz = 0;
        for (int32_t i = 0; i < rows; ++i) nz += padded(rowptr[r0 + i + 1] - rowptr[r0 + i]);
      }
      const int32_t D = tile_ptr[t + 1] - tile_ptr[t];
      const size_t nzp = ((size_t)nz + 3) / 4 * 4, valoff = head_bytes + nzp * 4, bytes = off[t + 1] - off[t];
      memset(rec.data() + off[t], 0, bytes);
      memset(desc.data() + t * dw, 0, dw * 4);
      int32_t *head = (int32_t *)(rec.data() + off[t]);
      head[valoff_w] = (int32_t)valoff;
      for (int32_t i = 0; i < rows; ++i) head[self_w + i] = self_idx[(size_t)(r0 + i)];
      if (!pad) {
        for (int32_t i = 0; i <= rows; ++i) head[i] = rowptr[r0 + i] - p0;
        memcpy(rec.data() + off[t] + head_bytes, lcol.data() + p0, (size_t)nz * 4);
        memcpy(rec.data() + off[t] + valoff, vals + p0, (size_t)nz * sizeof(F));
      } else {
        int32_t *lc_out = (int32_t *)(rec.data() + off[t] + head_bytes);
        F *va_out = (F *)(rec.data() + off[t] + valoff);
        int32_t w = 0;
        for (int32_t i = 0; i < rows; ++i) {
          const int32_t q0 = rowptr[r0 + i], cnt = rowptr[r0 + i + 1] - q0, pc = padded(cnt);
          head[i] = w;
          for (int32_t q = 0; q < pc; ++q) {
            lc_out[w + q] = q < cnt ? lcol[(size_t)(q0 + q)] : self_idx[(size_t)(r0 + i)];
            va_out[w + q] = q < cnt ? vals[q0 + q] : (F)0;
          }
          w += pc;
        }
        head[rows] = w;
      }
      int32_t *d = desc.data() + t * dw;
      d[kDescCols] = D;
      d[kDescRecOff] = (int32_t)(off[t] / 16);
      d[kDescRecChunks] = (int32_t)((bytes + 1023) / 1024);
      d[kDescRow0] = r0;
      d[kDescRows] = rows;
      const int32_t nd = (D + R - 1) / R;
      if (R == 1) {  // (de-interleaved: even lines, then odd ones - slq_common.hpp: ring1_list_pos)
        for (int32_t c = 0; c < D; ++c) d[kDescList + ring1_list_pos(c)] = tile_cols[(size_t)tile_ptr[t] + c];
      } else {
        for (int32_t c = 0; c < nd * R; ++c) d[(size_t)(c % R) * 64 + kDescList + c / R] = tile_cols[(size_t)tile_ptr[t] + std::min(c, D - 1)];
      }
    }
  });
  if (!ok) throw std::bad_alloc();
}

// Tiles of the upper-triangle stream (the alpha-only pass, r04). That pass lands 31-32 GB/s per CU by LDS-DMA whatever the operator (configs[1]: 1.64 KiB per row,
// 0.40 ms; 100^3: 2.65 KiB per row, 0.65 ms) - the DMA path's own cadence - so what shortens it is fewer landed lines per row. The base tiles are cut to what a slot
// holds of FULL rows; over the upper triangle the same rows need two thirds of the lines, so consecutive base tiles of a chunk - neighbours in the sweep, which share
// halo - are joined while the run keeps to kRingTileRows rows, kRingTileCols distinct lines (rows and upper columns) and kRingTileNnz padded entries: 100^3, 118,940 ->
// 107,848 tiles, alpha pass 0.652 -> 0.607 ms. (Cutting the chunk's rows anew, row by row, to the same caps gives 12.9-row tiles that straddle cluster boundaries and
// land MORE lines per row, 2.65 against 2.47: 0.82 ms. Not kept.) Tiles stay contiguous row ranges of one XCD chunk; kernel and stream format do not change.
// each_upper(r, consider): calls consider(c) for every column c >= r of stored row r and returns how many there were (the upper
// triangle's CSR, or - before that exists - the caller's CSR seen through the permutation: the columns' order does not matter)
template <typename EachUpper>
static void regroup_upper_tiles_impl(EachUpper each_upper, const std::vector<int32_t> &tile_row, const int32_t xcd_tile[9],
                                     std::vector<int32_t> &tile_row_u, int32_t xcd_tile_u[9]) {
  auto padded = [](int32_t cnt) { return std::max<int32_t>(4, (cnt + 3) / 4 * 4); };
  // every chunk on its own (in parallel): consecutive base tiles - neighbours in the sweep - joined while the run keeps to the caps
  std::vector<int32_t> cuts[8];
  const bool ok = parallel_pieces(8, 8, [&](int, int64_t x0, int64_t x1) {
    for (int64_t x = x0; x < x1; ++x) {
      std::vector<int32_t> &out = cuts[x];
      if (xcd_tile[x] >= xcd_tile[x + 1]) continue;
      // membership by stamps (r04: the lists were searched linearly - 25 ms of a 100^3 operator's creation): in_run[c - base] == run: c is a line of
      // the current run; in_tile[c - base] == stamp: c was counted for the base tile under consideration. Every column of a chunk's rows is >= base.
      const int32_t base = tile_row[(size_t)xcd_tile[x]], n_all = tile_row.back();
      std::vector<int32_t> in_run((size_t)(n_all - base), -1), in_tile((size_t)(n_all - base), -1);
      int32_t run = 0, stamp = 0;
      int nl = 0, rows = 0, nz = 0;
      for (int32_t t = xcd_tile[x]; t < xcd_tile[x + 1]; ++t) {
        const int32_t r0 = tile_row[(size_t)t], r1 = tile_row[(size_t)t + 1];
        // what this base tile lists - its rows and their upper columns, each once - and how much of that the run does not list yet
        int32_t all[kRingTileCols + 16];
        int na = 0, nf = 0;
        int32_t pz = 0;
        auto consider = [&](int32_t c) {
          const size_t k = (size_t)(c - base);
          if (in_tile[k] == stamp) return;
          in_tile[k] = stamp;
          if (na < kRingTileCols + 16) all[na++] = c, nf += in_run[k] != run;
        };
        for (int32_t r = r0; r < r1; ++r) {
          consider(r);
          pz += padded(each_upper(r, consider));
        }
        ++stamp;
        if (rows > 0 && (rows + (r1 - r0) > kRingTileRows || nl + nf > kRingTileCols || nz + pz > kRingTileNnz)) {
          nl = rows = nz = 0;  // cut: this base tile opens the next run (its own lines: everything it lists)
          ++run;
        }
        if (rows == 0) out.push_back(r0);
        for (int q = 0; q < na && nl < 2 * kRingTileCols + 16; ++q) {
          int32_t &m = in_run[(size_t)(all[q] - base)];
          if (m != run) m = run, ++nl;
        }
        rows += r1 - r0;
        nz += pz;
      }
    }
  });
  if (!ok) throw std::bad_alloc();
  tile_row_u.clear();
  for (int x = 0; x < 8; ++x) {
    xcd_tile_u[x] = (int32_t)tile_row_u.size();
    tile_row_u.insert(tile_row_u.end(), cuts[x].begin(), cuts[x].end());
  }
  xcd_tile_u[8] = (int32_t)tile_row_u.size();
  tile_row_u.push_back(tile_row.back());
}
static void regroup_upper_tiles(const int32_t *urp, const int32_t *uci, const std::vector<int32_t> &tile_row, const int32_t xcd_tile[9],
                                std::vector<int32_t> &tile_row_u, int32_t xcd_tile_u[9]) {
  regroup_upper_tiles_impl(
      [&](int32_t r, auto &consider) {
        for (int32_t q = urp[r]; q < urp[r + 1]; ++q) consider(uci[q]);
        return urp[r + 1] - urp[r];
      },
      tile_row, xcd_tile, tile_row_u, xcd_tile_u);
}

// If the CSR (rows sorted, no duplicates) is exactly symmetric, emit its upper triangle with the strict
// upper entries doubled and return true. Row ranges in parallel: every off-diagonal entry (i, j) looks its mirror (j, i)
// up by bisection in row j (rows are sorted - checked on the way) and compares the values; the upper entries are then
// counted per row, placed by a prefix sum and written, again by row ranges.
template <typename F>
static bool build_symmetric_upper(int64_t n, const int32_t *rowptr, const int32_t *colind, const F *vals,
                                  std::vector<int32_t> &urp, std::vector<int32_t> &uci, std::vector<char> &uva) {
  urp.assign((size_t)n + 1, 0);
  const int pieces = host_threads();
  std::vector<char> bad((size_t)pieces, 0);
  if (!parallel_pieces(pieces, n, [&](int piece, int64_t i0, int64_t i1) {
        for (int64_t i = i0; i < i1 && !bad[(size_t)piece]; ++i) {
          int32_t up = 0;
          for (int32_t q = rowptr[i]; q < rowptr[i + 1]; ++q) {
            const int32_t j = colind[q];
            if (q > rowptr[i] && colind[q - 1] >= j) { bad[(size_t)piece] = 1; break; }  // unsorted or duplicate
            up += j >= i;
            if (j == i) continue;
            const int32_t *lo = colind + rowptr[j], *hi = colind + rowptr[j + 1];
            const int32_t *m = std::lower_bound(lo, hi, (int32_t)i);
            if (m == hi || *m != (int32_t)i || !(vals[m - colind] == vals[q])) { bad[(size_t)piece] = 1; break; }
          }
          urp[(size_t)i + 1] = up;
        }
      }))
    return false;
  if (std::any_of(bad.begin(), bad.end(), [](char c) { return c != 0; })) return false;
  for (int64_t i = 0; i < n; ++i) urp[(size_t)i + 1] += urp[(size_t)i];
  const size_t nu = (size_t)urp[(size_t)n];
  uci.resize(nu);
  uva.resize(nu * sizeof(F));
  F *uv = (F *)uva.data();
  return parallel_pieces(pieces, n, [&](int, int64_t i0, int64_t i1) {
    for (int64_t i = i0; i < i1; ++i) {
      size_t w = (size_t)urp[(size_t)i];
      for (int32_t q = rowptr[i]; q < rowptr[i + 1]; ++q) {
        const int32_t j = colind[q];
        if (j < i) continue;
        uci[w] = j;
        uv[w] = j == i ? vals[q] : (F)2 * vals[q];
        ++w;
      }
    }
  });
}

// ---- derived data built on the device (slq_build.hpp) ----
struct DevBuf {  // device scratch of a build, freed when the build is left
  void *p = nullptr;
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  ~DevBuf() { release(); }
  void release() {
    if (p) hipFree(p);
    p = nullptr;
  }
  hipError_t alloc(size_t bytes) {
    release();
    return hipMalloc(&p, std::max<size_t>(bytes, 16));
  }
  template <typename T> T *as() const { return (T *)p; }
  void *take() {
    void *q = p;
    p = nullptr;
    return q;
  }
};
// a[0, count) := its inclusive scan (the callers keep a zero in front of it: row pointers, record offsets)
static hipError_t device_scan_inclusive(int32_t *a, int64_t count, hipStream_t st) {
  if (count <= 0) return hipSuccess;
  const int nb = (int)((count + slqb::kScanTile - 1) / slqb::kScanTile);
  DevBuf sums;
  hipError_t e = sums.alloc((size_t)nb * 4);
  if (e != hipSuccess) return e;
  slqb::k_scan_block_sums<<<dim3(nb), dim3(256), 0, st>>>(a, count, sums.as<int32_t>());
  slqb::k_scan_sums<<<dim3(1), dim3(1024), 0, st>>>(sums.as<int32_t>(), nb);
  slqb::k_scan_apply<<<dim3(nb), dim3(256), 0, st>>>(a, count, sums.as<int32_t>());
  e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(st);  // (sums is freed on return)
  return e;
}
struct DeviceStream {  // what device_build_stream hands back (the caller owns desc, rec, tile_ptr)
  int32_t *desc = nullptr;
  char *rec = nullptr;
  int32_t *tile_ptr = nullptr;  // [ntiles + 1] running sum of the lists' lengths (only if asked for)
  size_t desc_bytes = 0, rec_bytes = 0;
  int max_lines = 0;
  int64_t sum_lines = 0;
  bool padded = false;
};
// The descriptor / record stream of R-merged tiles `tile_row_d` (device, ntiles + 1) over the CSR (rp, ci, va) (device): what
// build_tile_meta + build_ring_stream produce on the host, byte for byte. want_pad: rows padded to whole chunks of four entries
// unless some record would outgrow its slot (out.padded tells). max_lines_per_row > 0: nothing is built (return 1) when the tiles
// land more distinct panel rows per row than that; out.max_lines / out.sum_lines are set either way. 0: built; < 0: failed
// (the SLQ status).
static int device_build_stream(slq_context *ctx, int dtype, int R, int64_t n, const int32_t *rp, const int32_t *ci, const void *va, const int32_t *tile_row_d,
                               int ntiles, bool want_pad, double max_lines_per_row, bool keep_tile_ptr, DeviceStream &out) {
  hipStream_t st = ctx->stream;
  const int cap = kRingTileCols * R;
  const int es = (int)esize(dtype);
  const int head_bytes = kRecHeadBytes * R, pad_limit = (kRingRecStride * R + 1023) / 1024 * 1024;
  DevBuf lists, small;
  const size_t cnt = (size_t)ntiles + 1;
  hipError_t e = lists.alloc((size_t)ntiles * cap * 4);
  if (e == hipSuccess) e = small.alloc((3 * cnt + 4) * 4);
  if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? SLQ_ENOMEM : SLQ_EHIP, "tile stream scratch: %s", hipGetErrorString(e));
  int32_t *D = small.as<int32_t>(), *units = D + cnt, *units_pad = units + cnt;
  int *flags = (int *)(units_pad + cnt);
  e = hipMemsetAsync(small.p, 0, (3 * cnt + 4) * 4, st);
  if (e != hipSuccess) return fail(SLQ_EHIP, "tile stream scratch: %s", hipGetErrorString(e));
  const dim3 gl((unsigned)((ntiles + 63) / 64));
  if (R == 1) slqb::k_tile_lists<kRingTileCols><<<gl, dim3(64), 0, st>>>(ntiles, rp, ci, tile_row_d, lists.as<int32_t>(), D, units, units_pad, head_bytes, es, pad_limit, flags);
  else if (R == 2) slqb::k_tile_lists<2 * kRingTileCols><<<gl, dim3(64), 0, st>>>(ntiles, rp, ci, tile_row_d, lists.as<int32_t>(), D, units, units_pad, head_bytes, es, pad_limit, flags);
  else if (R == 4) slqb::k_tile_lists<4 * kRingTileCols><<<gl, dim3(64), 0, st>>>(ntiles, rp, ci, tile_row_d, lists.as<int32_t>(), D, units, units_pad, head_bytes, es, pad_limit, flags);
  else return fail(SLQ_EINVAL, "tiles are merged 1, 2 or 4 at a time");
  int hflags[4] = {0, 0, 0, 0};
  e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(hflags, flags, sizeof hflags, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) return fail(SLQ_EHIP, "tile lists: %s", hipGetErrorString(e));
  if (hflags[0]) return fail(SLQ_EHIP, "a tile lists more than %d distinct panel rows (tiles not built to the ring's caps)", cap);
  out.max_lines = hflags[2];
  out.padded = want_pad && !hflags[1];
  int32_t *u = out.padded ? units_pad : units;
  e = device_scan_inclusive(D + 1, ntiles, st);
  if (e == hipSuccess) e = device_scan_inclusive(u + 1, ntiles, st);
  int32_t totals[2] = {0, 0};
  if (e == hipSuccess) e = hipMemcpyAsync(&totals[0], D + ntiles, 4, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipMemcpyAsync(&totals[1], u + ntiles, 4, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) return fail(SLQ_EHIP, "tile stream offsets: %s", hipGetErrorString(e));
  out.sum_lines = totals[0];
  if (max_lines_per_row > 0.0 && (double)totals[0] / (double)n > max_lines_per_row) return 1;
  DevBuf desc, rec, tptr;
  out.desc_bytes = (size_t)ntiles * 64 * R * 4;
  out.rec_bytes = (size_t)totals[1] * 16 + (size_t)kRingMetaBytes * R;
  e = desc.alloc(out.desc_bytes);
  if (e == hipSuccess) e = rec.alloc(out.rec_bytes);
  if (e == hipSuccess && keep_tile_ptr) e = tptr.alloc(cnt * 4);
  if (e == hipSuccess && keep_tile_ptr) e = hipMemcpyAsync(tptr.p, D, cnt * 4, hipMemcpyDeviceToDevice, st);
  if (e == hipSuccess) e = hipMemsetAsync(rec.as<char>() + (size_t)totals[1] * 16, 0, (size_t)kRingMetaBytes * R, st);  // the spare record behind the last one
  if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? SLQ_ENOMEM : SLQ_EHIP, "tile stream: %s", hipGetErrorString(e));
  const dim3 gs((unsigned)((ntiles + 3) / 4));
  const int pad = out.padded ? 1 : 0;
#define SLQ_STREAM_LAUNCH(F, CAP) \
  slqb::k_tile_stream<F, CAP><<<gs, dim3(256), 0, st>>>(ntiles, R, pad, rp, ci, (const F *)va, tile_row_d, lists.as<int32_t>(), D, u, desc.as<int32_t>(), rec.as<char>())
  if (dtype == SLQ_F64) {
    if (R == 1) SLQ_STREAM_LAUNCH(double, kRingTileCols);
    else if (R == 2) SLQ_STREAM_LAUNCH(double, 2 * kRingTileCols);
    else SLQ_STREAM_LAUNCH(double, 4 * kRingTileCols);
  } else {
    if (R == 1) SLQ_STREAM_LAUNCH(float, kRingTileCols);
    else if (R == 2) SLQ_STREAM_LAUNCH(float, 2 * kRingTileCols);
    else SLQ_STREAM_LAUNCH(float, 4 * kRingTileCols);
  }
#undef SLQ_STREAM_LAUNCH
  e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(st);  // (the scratch goes when this returns)
  if (e != hipSuccess) return fail(SLQ_EHIP, "tile stream: %s", hipGetErrorString(e));
  out.desc = (int32_t *)desc.take();
  out.rec = (char *)rec.take();
  out.tile_ptr = keep_tile_ptr ? (int32_t *)tptr.take() : nullptr;
  return 0;
}
static bool device_equals_host(const void *dev, const void *host, size_t bytes) {
  std::vector<char> tmp(bytes);
  if (hipMemcpy(tmp.data(), dev, bytes, hipMemcpyDeviceToHost) != hipSuccess) return false;
  return memcmp(tmp.data(), host, bytes) == 0;
}

// plain != 0: rows stay in the caller's order and no derived copy (upper triangle, tiles) is built - for operators whose
// values change after creation (the affine operator)
// dev_*: the same CSR already on the device (slq_csr_create_device): the device-side build reads it in place, and `vals` may then be null
// (the values come back to the host only if the host has to build the operator after all)
struct DeviceCsr { const int32_t *rp = nullptr, *ci = nullptr; const void *va = nullptr; };
static int csr_create_impl(slq_context *ctx, int dtype, int64_t n, int64_t nnz, const int32_t *rowptr, const int32_t *colind,
                           const void *vals, slq_operator **out, int plain, int host_build = 0, DeviceCsr dev = DeviceCsr());

extern "C" int slq_csr_create(slq_context *ctx, int dtype, int64_t n, int64_t nnz,
                              const int32_t *rowptr, const int32_t *colind, const void *vals,
                              slq_operator **out) {
  return csr_create_impl(ctx, dtype, n, nnz, rowptr, colind, vals, out, 0);
}

static int csr_create_body(slq_context *ctx, int dtype, int64_t n, int64_t nnz, const int32_t *rowptr, const int32_t *colind,
                           const void *vals, slq_operator **out, int plain, int host_build, DeviceCsr dev);
static int csr_create_impl(slq_context *ctx, int dtype, int64_t n, int64_t nnz, const int32_t *rowptr, const int32_t *colind,
                           const void *vals, slq_operator **out, int plain, int host_build, DeviceCsr dev) {
  try {  // (host-side allocations of the analysis: no C++ exception crosses the C boundary)
    return csr_create_body(ctx, dtype, n, nnz, rowptr, colind, vals, out, plain, host_build, dev);
  } catch (const std::bad_alloc &) {
    if (out) *out = nullptr;
    return fail(SLQ_ENOMEM, "host allocation failed while analysing the operator");
  } catch (const std::exception &e) {  // (e.g. no thread to be had)
    if (out) *out = nullptr;
    return fail(SLQ_EHIP, "operator analysis failed: %s", e.what());
  }
}

static int csr_finish_on_device(slq_context *ctx, slq_operator *op, int dtype, int64_t n, int64_t nnz, const int32_t *rowptr0, const int32_t *colind0,
                                const int32_t *o_rp, const int32_t *o_ci, const void *o_va, UploadQueue &early, const std::vector<int32_t> &tile_row,
                                const int32_t xcd_tile[9], PhaseClock &clk);
static int operators_differ(const slq_operator *a, const slq_operator *b);
static int csr_create_body(slq_context *ctx, int dtype, int64_t n, int64_t nnz, const int32_t *rowptr, const int32_t *colind,
                           const void *vals, slq_operator **out, int plain, int host_build, DeviceCsr dev) {
  if (!ctx || !out) return fail(SLQ_EINVAL, "ctx/out is NULL");
  *out = nullptr;
  SLQ_TRY(check_dtype(dtype));
  if (n <= 0 || n >= (int64_t)1 << 31 || nnz < 0 || nnz >= (int64_t)1 << 31)
    return fail(SLQ_EINVAL, "CSR shape out of range for int32 indices (n=%lld, nnz=%lld)",
                (long long)n, (long long)nnz);
  if (!rowptr || (nnz > 0 && (!colind || (!vals && !dev.va)))) return fail(SLQ_EINVAL, "CSR arrays are NULL");
  if (rowptr[0] != 0 || rowptr[n] != nnz)
    return fail(SLQ_EINVAL, "rowptr[0] must be 0 and rowptr[n] must equal nnz");
  PhaseClock clk;
  std::vector<char> vals_back;  // the values of a device-resident CSR, fetched when the host needs them
  auto need_host_vals = [&]() -> hipError_t {
    if (vals || nnz == 0) return hipSuccess;
    vals_back.resize((size_t)nnz * esize(dtype));
    const hipError_t he = hipMemcpy(vals_back.data(), dev.va, vals_back.size(), hipMemcpyDeviceToHost);
    vals = vals_back.data();
    return he;
  };
  for (int64_t i = 0; i < n; ++i)
    if (rowptr[i + 1] < rowptr[i]) return fail(SLQ_EINVAL, "rowptr is not non-decreasing at %lld", (long long)i);
  {
    // every column index inside [0, n): ranges of the array in parallel, the first offender (lowest position) reported
    const int pieces = host_threads();
    std::vector<int64_t> bad((size_t)pieces, -1);
    if (!parallel_pieces(pieces, nnz, [&](int piece, int64_t p0, int64_t p1) {
          for (int64_t p = p0; p < p1; ++p)
            if (colind[p] < 0 || colind[p] >= n) { bad[(size_t)piece] = p; break; }
        }))
      return fail(SLQ_ENOMEM, "host worker failed while validating the column indices");
    for (int64_t b : bad)
      if (b >= 0) return fail(SLQ_EINVAL, "column index %d out of range at position %lld", colind[b], (long long)b);
  }
  HIP_TRY(hipSetDevice(ctx->device));
  clk.lap("validation");
  if (ctx->dead) return fail(SLQ_EINVAL, "the context has been destroyed");
  slq_operator *op = new (std::nothrow) slq_operator();
  if (!op) return fail(SLQ_ENOMEM, "host allocation failed");
  ctx_retain(ctx);
  *op = slq_operator{ctx, OP_CSR, dtype, n, nnz, nullptr, nullptr, nullptr, 0, true, nullptr, nullptr, nullptr, nullptr, TileMeta{}};
  // whatever way this function is left without handing `op` out - an error return below or an exception of the host-side
  // analysis - the operator and what it owns on the device go with it (declared before the upload queue: that one joins first)
  struct OpGuard {
    slq_operator *op;
    ~OpGuard() {
      if (op) slq_operator_destroy(op);
    }
  } guard{op};
  const size_t es = esize(dtype);
  // optional XCD-aware reordering: A' = P A P^T stored, vectors live in the permuted row space
  std::vector<int32_t> rp2;
  RawBuf<int32_t> ci2;
  RawBuf<char> va2;
  // SLQ_REORDER: 0 never, 1 operators with n >= 65536, 2 always, unset = automatic. Measured (DESIGN.md
  // §5.3): on the 2-D grid of configs[1] (rms |i-j| of the nonzeros = 632 rows) it RAISED the alpha pass's
  // fetch traffic from 6.5 to 8.9 GB and the step time by 10 %; on 3-D grids (100^3: rms |i-j| = 5345,
  // 126^3: 8486) whose natural-order halo no longer fits any cache level it is 9-12 % FASTER. Automatic
  // mode therefore reorders only when the rms index distance exceeds 2048 rows AND the permutation cuts
  // it to 60 % or less (random graphs gain nothing and are left alone).
  const int reorder_mode = plain ? 0 : env_int("SLQ_REORDER", -1);
  // rms index distance of the nonzeros whose two ends lie in the same XCD chunk (links that cross
  // chunks are served by another XCD's L2 whatever the order inside the chunks)
  const int64_t rchunk = (n + 7) / 8;
  auto mean_dist = [&](const std::vector<int32_t> *inv) {
    const int pieces = 8;  // (a fixed partition: the sum does not depend on how many threads ran it)
    std::vector<double> acc((size_t)pieces, 0.0);
    std::vector<int64_t> cnt((size_t)pieces, 0);
    if (!parallel_pieces(pieces, n, [&](int piece, int64_t i0, int64_t i1) {
      double a = 0.0;
      int64_t c = 0;
      for (int64_t i = i0; i < i1; ++i) {
        const int64_t ii = inv ? (*inv)[(size_t)i] : i;
        for (int32_t q = rowptr[i]; q < rowptr[i + 1]; ++q) {
          if (colind[q] / rchunk != i / rchunk) continue;
          const int64_t jj = inv ? (*inv)[(size_t)colind[q]] : colind[q];
          const double dd = (double)(ii > jj ? ii - jj : jj - ii);
          a += dd * dd;
          ++c;
        }
      }
      acc[(size_t)piece] = a;
      cnt[(size_t)piece] = c;
    })) throw std::bad_alloc();
    double a = 0.0;
    int64_t c = 0;
    for (int t = 0; t < pieces; ++t) a += acc[(size_t)t], c += cnt[(size_t)t];  // (piece order: the same value whatever the timing)
    return std::sqrt(a / (double)std::max<int64_t>(c, 1));
  };
  // Workgroup tiles (SLQ_TILES, tiles_mode()): the rows are regrouped into compact clusters = the tiles of k_csr_tile_pass /
  // k_csr_ring_pass, on top of a base order. Kept only if the tiles actually share rows: at most kTileMaxColsPerRow distinct
  // panel rows per tile row (5-point grid: 2.1, 7-point grid: 3.9 with the ring kernel's 36-row images, random graph: 10+).
  // Unasked (SLQ_TILES unset) only operators of 65536 rows and more are tried - below that a pass is launch-bound anyway.
  const int tmode = plain ? 0 : tiles_mode();
  const bool tiles_forced = getenv("SLQ_TILES") != nullptr;
  bool try_tiles = tmode != 0 && nnz > 0 && n >= (tiles_forced ? 4096 : 65536);
  if (try_tiles) {
    // (a sample cluster grows on the caller's numbering, the real ones on the reordered chunk: 4.3 against 3.9 on a 7-point grid,
    // 2.3 against 2.1 on a 5-point one, 12-16 on the operators this is meant to turn away. 25 % of margin keeps it a filter for
    // those only)
    const bool ringed = tmode == 2;
    const double q = sample_tile_quality(n, rowptr, colind, ringed ? kRingTileRows : std::max(1, std::min(env_int("SLQ_TILE_ROWS", kTileRows), 64)),
                                         ringed ? kRingTileCols : std::max(8, std::min(env_int("SLQ_TILE_COLS", kTileCols), kTileCols)),
                                         ringed ? kRingTileNnz : std::numeric_limits<int>::max());
    if (env_int("SLQ_DEBUG", 0) != 0) fprintf(stderr, "[slq] tiles: sample of 256 clusters: %.2f distinct panel rows per row\n", q);
    if (q > 1.25 * kTileMaxColsPerRow) try_tiles = false;
  }
  clk.lap("tile sample");
  // SLQ_DEVICE_BUILD (r04, slq_build.hpp): 1 (default) - an operator that gets ring-sized tiles has its stored CSR, upper triangle and
  // tile streams built on the device from the caller's CSR, which starts its way up now, while the host orders and clusters the rows;
  // 0 - everything on the host, as before; 2 - both, compared array by array (tests)
  const int dev_mode = (plain || host_build) ? 0 : env_int("SLQ_DEVICE_BUILD", 1);
  DevBuf o_rp, o_ci, o_va;       // the caller's CSR on the device (scratch of the build)
  UploadQueue early(ctx->device);  // (declared after what it fills: joined first)
  bool early_started = false;
  const int32_t *src_rp = nullptr, *src_ci = nullptr;  // where the device-side build reads the caller's CSR
  const void *src_va = nullptr;
  if (dev_mode != 0 && try_tiles && tmode == 2 && reorder_mode != 0 && env_int("SLQ_RING_ORDER", 0) == 0 && dev.va) {
    src_rp = dev.rp, src_ci = dev.ci, src_va = dev.va;  // in place (slq_csr_create_device)
    early_started = true;
  } else if (dev_mode != 0 && try_tiles && tmode == 2 && reorder_mode != 0 && env_int("SLQ_RING_ORDER", 0) == 0) {
    hipError_t ee = o_rp.alloc((size_t)(n + 1) * 4);
    if (ee == hipSuccess) ee = o_ci.alloc((size_t)nnz * 4);
    if (ee == hipSuccess) ee = o_va.alloc((size_t)nnz * esize(dtype));
    if (ee != hipSuccess) return fail(ee == hipErrorOutOfMemory ? SLQ_ENOMEM : SLQ_EHIP, "CSR upload: %s", hipGetErrorString(ee));
    early.push({{o_rp.p, rowptr, (size_t)(n + 1) * 4}, {o_ci.p, colind, (size_t)nnz * 4}, {o_va.p, vals, (size_t)nnz * esize(dtype)}});
    early_started = true;
    src_rp = o_rp.as<int32_t>(), src_ci = o_ci.as<int32_t>(), src_va = o_va.p;
  }
  const double tile_limit = kTileMaxColsPerRow;
  std::vector<int32_t> tile_row;
  int32_t xcd_tile[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  bool have_tiles = false;
  // clusters on top of `base` (stored row -> caller row; null: the caller's order); on success `order` is the new order
  auto cluster_tiles = [&](const std::vector<int32_t> *base, std::vector<int32_t> &order) -> bool {
    std::vector<int32_t> inv0;
    if (base) {
      inv0.resize((size_t)n);
      for (int64_t i = 0; i < n; ++i) inv0[(size_t)(*base)[(size_t)i]] = (int32_t)i;
    }
    int64_t dsum = 0;  // distinct indices (rows and columns) summed over the tiles: the clusters count them as they grow
    if (!build_clusters(n, rowptr, colind, base ? base->data() : nullptr, base ? inv0.data() : nullptr, order, tile_row, xcd_tile, &dsum)) return false;
    clk.lap("  clusters");
    const double per_row = (double)dsum / (double)n;
    if (env_int("SLQ_DEBUG", 0) != 0)
      fprintf(stderr, "[slq] tiles: %zu clusters, %.2f rows each, %.2f distinct panel rows per row (limit %.1f)\n", tile_row.size() - 1,
              (double)n / (double)(tile_row.size() - 1), per_row, tile_limit);
    return per_row <= tile_limit;
  };
  auto adopt = [&](std::vector<int32_t> &order) -> bool {
    if (!op->perm_h) op->perm_h = new (std::nothrow) std::vector<int32_t>();
    if (!op->perm_h) return false;
    op->perm_h->swap(order);
    return true;
  };
  std::vector<int32_t> rcm_perm;  // the in-chunk Cuthill-McKee order, computed at most once
  const int sub_env = env_int("SLQ_RCM_SUB", 0);  // 0: 1 piece, except for the tile sweep below
  auto rcm_order = [&]() -> const std::vector<int32_t> & {
    if (rcm_perm.empty()) xcd_rcm_permutation(n, rowptr, colind, rcm_perm, std::max(1, sub_env), nullptr);
    return rcm_perm;
  };
  // Ring-fed tiles sweep a chunk tile after tile, 32 CUs abreast, and re-read a neighbour tile's rows from L2 only if the
  // neighbour is at most a few dozen tiles away: the BASE order must have short level sets, whatever the index distances
  // are. On the 2-D grid of configs[1] in its natural order (grid rows of 1000 = 270 tiles) every vertical neighbour was
  // fetched again (7.5 GB per dots pass against 6.3 algorithmic); on the in-chunk Cuthill-McKee order (levels of <= 125
  // nodes = 34 tiles) the pass fetches 6.37 GB. So mode 2 clusters the Cuthill-McKee order unless SLQ_REORDER=0 forbids it.
  // ... and level sets no longer than about one round of the sweep (32 CUs x 10 rows): a 12.5-plane slab of a 100^3 grid has
  // level sets of 590 rows on average - its tiles then fetch 8.7 GB per dots pass against 6.1 algorithmic - so the chunk's
  // order is cut into 4, 16, 64 runs of levels, each reordered on its own (xcd_rcm_permutation), until they are: 16 pieces
  // there (level sets of ~200 rows, 7.5 GB). SLQ_RCM_SUB fixes the number of pieces.
  if (try_tiles && tmode == 2 && reorder_mode != 0) {
    if (sub_env <= 0) {
      double w = 0.0;
      std::vector<int32_t> level1;  // the chunks' own order (k = 1), which every finer attempt starts from
      for (int k = 1; k <= 64; k *= 4) {
        if (k == 4) level1 = rcm_perm;
        xcd_rcm_permutation(n, rowptr, colind, rcm_perm, k, &w, k > 1 ? &level1 : nullptr);
        clk.lap("  Cuthill-McKee in the chunks");
        if (env_int("SLQ_DEBUG", 0) != 0) fprintf(stderr, "[slq] tiles: %d piece(s) per chunk: level sets of %.0f rows on average\n", k, w);
        if (w <= kTileLevelRows) break;
      }
    }
    std::vector<int32_t> order;
    if (cluster_tiles(&rcm_order(), order)) {
      have_tiles = true;
      if (!adopt(order)) { return fail(SLQ_ENOMEM, "host allocation failed"); }
    } else if (sub_env <= 0) {
      rcm_perm.clear();  // declined: the generic passes keep their own (one-piece) order, decided below
    }
  }
  clk.lap("base order + clusters");
  bool want = false;
  if (nnz > 0 && !have_tiles) {
    if (reorder_mode == 2) want = true;
    else if (reorder_mode == 1) want = n >= 65536;
    else if (reorder_mode < 0) want = n >= 65536 && mean_dist(nullptr) > 2048.0;
  }
  if (want) {
    op->perm_h = new (std::nothrow) std::vector<int32_t>();
    if (!op->perm_h) { return fail(SLQ_ENOMEM, "host allocation failed"); }
    std::vector<int32_t> &perm = *op->perm_h;
    perm = rcm_order();
    std::vector<int32_t> inv((size_t)n);
    for (int64_t i = 0; i < n; ++i) inv[(size_t)perm[(size_t)i]] = (int32_t)i;
    const double d_new = mean_dist(&inv);
    if (reorder_mode < 0 && d_new > 0.6 * mean_dist(nullptr)) {
      delete op->perm_h;  // no locality to gain: keep the caller's order
      op->perm_h = nullptr;
    } else {
      op->rms_dist = d_new;
    }
  }
  // tiles on top of whatever order was chosen above: mode 1, and mode 2 when SLQ_REORDER=0 kept it from its own base order
  if (try_tiles && !have_tiles && (tmode == 1 || reorder_mode == 0)) {
    std::vector<int32_t> order;
    if (cluster_tiles(op->perm_h, order)) {
      have_tiles = true;
      if (!adopt(order)) { return fail(SLQ_ENOMEM, "host allocation failed"); }
    }
  }
  if (have_tiles) {
    std::vector<int32_t> inv((size_t)n);
    for (int64_t i = 0; i < n; ++i) inv[(size_t)(*op->perm_h)[(size_t)i]] = (int32_t)i;
    op->rms_dist = mean_dist(&inv);
  }
  if (op->rms_dist < 0.0 && nnz > 0) op->rms_dist = mean_dist(nullptr);
  clk.lap("reorder decision");
  if (early_started && have_tiles && op->perm_h) {
    const int rc = csr_finish_on_device(ctx, op, dtype, n, nnz, rowptr, colind, src_rp, src_ci, src_va, early, tile_row, xcd_tile, clk);
    if (rc != SLQ_OK) return rc;
    if (dev_mode == 2) {  // the same operator built on the host: every array must be the same
      if (need_host_vals() != hipSuccess) return fail(SLQ_EHIP, "CSR values: copy back failed");
      slq_operator *ref = nullptr;
      const int rr = csr_create_impl(ctx, dtype, n, nnz, rowptr, colind, vals, &ref, plain, 1);
      if (rr != SLQ_OK) return rr;
      const int diff = operators_differ(op, ref);
      slq_operator_destroy(ref);
      if (diff) return fail(SLQ_EHIP, "SLQ_DEVICE_BUILD=2: the device-built operator differs from the host-built one (item %d)", diff);
    }
    clk.total("all of slq_csr_create");
    guard.op = nullptr;
    *out = op;
    return SLQ_OK;
  }
  if (early_started) {  // no tiles after all: the host path below uploads what it builds
    early.wait();
    o_rp.release(), o_ci.release(), o_va.release();
  }
  if (need_host_vals() != hipSuccess) return fail(SLQ_EHIP, "CSR values: copy back failed");
  // From here on every array goes to the device through `up` while the next one is being built; the buffers it reads are
  // declared before it and nothing returns without up.wait() (its destructor, at the latest).
  std::vector<int32_t> inv_keep;                  // stored row of every caller row (reordered operators)
  std::vector<int32_t> urp, uci;                  // the upper triangle (stored order), also the source of the alpha-only tile stream
  std::vector<char> uva;
  std::vector<int32_t> tp, tc, lc, si, tpu, tcu, lcu, siu;
  RawBuf<int32_t> desc, desc_u;  // (the upper stream in buffers of its own: the full stream's upload is still running when it is built)
  RawBuf<char> rec, rec_u;
  UploadQueue up(ctx->device);
  auto bail = [&](int code, const char *what, hipError_t e) {
    up.wait();
    hipStreamSynchronize(ctx->stream);
    return fail(e == hipErrorOutOfMemory ? SLQ_ENOMEM : code, "%s: %s", what, hipGetErrorString(e));
  };
  if (op->perm_h) {
    std::vector<int32_t> &perm = *op->perm_h;
    inv_keep.resize((size_t)n);
    std::vector<int32_t> &inv = inv_keep;
    for (int64_t i = 0; i < n; ++i) inv[(size_t)perm[(size_t)i]] = (int32_t)i;
    hipError_t pe = hipMalloc((void **)&op->perm_d, (size_t)n * 4);
    if (pe == hipSuccess) pe = hipMalloc((void **)&op->inv_perm_d, (size_t)n * 4);
    if (pe != hipSuccess) return bail(SLQ_EHIP, "permutation upload", pe);
    up.push({{op->perm_d, perm.data(), (size_t)n * 4}, {op->inv_perm_d, inv.data(), (size_t)n * 4}});
    rp2.resize((size_t)n + 1);
    ci2.alloc((size_t)nnz);
    va2.alloc((size_t)nnz * es);
    rp2[0] = 0;
    for (int64_t i = 0; i < n; ++i) {
      const int32_t o = perm[(size_t)i];
      rp2[(size_t)i + 1] = rp2[(size_t)i] + (rowptr[o + 1] - rowptr[o]);
    }
    const bool pok = parallel_pieces(host_threads(), n, [&](int, int64_t i0, int64_t i1) {
      std::vector<std::pair<int32_t, int32_t>> rowbuf;
      for (int64_t i = i0; i < i1; ++i) {
        const int32_t o = perm[(size_t)i];
        rowbuf.clear();
        for (int32_t q = rowptr[o]; q < rowptr[o + 1]; ++q) rowbuf.emplace_back(inv[(size_t)colind[q]], q);
        std::sort(rowbuf.begin(), rowbuf.end());
        int32_t w = rp2[(size_t)i];
        for (auto &e2 : rowbuf) {
          ci2.data()[(size_t)w] = e2.first;
          memcpy(va2.data() + (size_t)w * es, (const char *)vals + (size_t)e2.second * es, es);
          ++w;
        }
      }
    });
    if (!pok) {
      return fail(SLQ_ENOMEM, "host allocation failed");
    }
    rowptr = rp2.data();
    colind = ci2.data();
    vals = va2.data();
  }
  clk.lap("permuted CSR");
  // colind/vals carry kCsrPad spare entries: the batched row gather (slq_kernels.hpp: gather_row_uniform)
  // loads indices and values 8 at a time and may read (never use) up to 7 entries past a row's end
  hipError_t e = hipMalloc((void **)&op->rowptr, (size_t)(n + 1) * 4);
  if (e == hipSuccess) e = hipMalloc((void **)&op->colind, ((size_t)nnz + kCsrPad) * 4);
  if (e == hipSuccess) e = hipMalloc(&op->vals, ((size_t)nnz + kCsrPad) * es);
  if (e == hipSuccess) e = hipMemsetAsync(op->colind + nnz, 0, kCsrPad * 4, ctx->stream);
  if (e == hipSuccess) e = hipMemsetAsync((char *)op->vals + (size_t)nnz * es, 0, kCsrPad * es, ctx->stream);
  if (e != hipSuccess) return bail(SLQ_EHIP, "CSR upload", e);
  up.push({{op->rowptr, rowptr, (size_t)(n + 1) * 4}, {op->colind, colind, (size_t)nnz * 4}, {op->vals, vals, (size_t)nnz * es}});
  {
    // gathers per row that reach further than any cache-resident halo (|i - j| > 4096 rows in the stored
    // order): what decides between the recompute passes and the store-and-revisit sweeps (enqueue_run)
    std::vector<int64_t> farp((size_t)host_threads(), 0);
    if (!parallel_pieces((int)farp.size(), n, [&](int piece, int64_t i0, int64_t i1) {
      int64_t f = 0;
      for (int64_t i = i0; i < i1; ++i)
        for (int32_t q = rowptr[i]; q < rowptr[i + 1]; ++q) f += std::llabs((long long)colind[q] - (long long)i) > 4096;
      farp[(size_t)piece] = f;
    })) return bail(SLQ_ENOMEM, "host worker failed (far-gather count)", hipSuccess);
    int64_t far = 0;
    for (int64_t f : farp) far += f;
    op->far_per_row = (double)far / (double)n;
  }
  if (env_int("SLQ_DEBUG", 0) != 0)
    fprintf(stderr, "[slq] csr n=%lld nnz=%lld reordered=%d rms in-chunk |i-j| = %.1f, far gathers per row %.2f\n", (long long)n,
            (long long)nnz, op->perm_h ? 1 : 0, op->rms_dist, op->far_per_row);
  clk.lap("far count");
  // Symmetric operators (what Lanczos assumes; the reference never checks): the alpha pass only needs the
  // scalar q^T A q, so it can run on the upper triangle with doubled off-diagonals and gather half the
  // panel rows. Built only when the stored CSR is EXACTLY symmetric (pattern and values, sorted rows
  // without duplicates); anything else keeps the full rows. SLQ_SYM_ALPHA=0 disables it.
  bool sym = false;
  if (!plain && env_int("SLQ_SYM_ALPHA", 1) != 0 && nnz > 0) {
    sym = dtype == SLQ_F64 ? build_symmetric_upper<double>(n, rowptr, colind, (const double *)vals, urp, uci, uva)
                                : build_symmetric_upper<float>(n, rowptr, colind, (const float *)vals, urp, uci, uva);
    if (sym) {
      const size_t nu = uci.size();
      op->nnz_u = (int64_t)nu;
      hipError_t ue = hipMalloc((void **)&op->rowptr_u, (size_t)(n + 1) * 4);
      if (ue == hipSuccess) ue = hipMalloc((void **)&op->colind_u, (nu + kCsrPad) * 4);
      if (ue == hipSuccess) ue = hipMalloc(&op->vals_u, (nu + kCsrPad) * es);
      if (ue == hipSuccess) ue = hipMemsetAsync(op->colind_u + nu, 0, kCsrPad * 4, ctx->stream);
      if (ue == hipSuccess) ue = hipMemsetAsync((char *)op->vals_u + nu * es, 0, kCsrPad * es, ctx->stream);
      if (ue != hipSuccess) return bail(SLQ_EHIP, "upper-triangle upload", ue);
      up.push({{op->rowptr_u, urp.data(), (size_t)(n + 1) * 4}, {op->colind_u, uci.data(), nu * 4}, {op->vals_u, uva.data(), nu * es}});
    }
  }
  clk.lap("upper triangle");
  // workgroup tiles (SLQ_TILES): lists of the stored CSR, uploaded next to it
  if (have_tiles) {
    int mx = 0;
    build_tile_meta(n, rowptr, colind, tile_row, tp, tc, lc, si, &mx);
    clk.lap("  tile lists");
    int32_t *d_tr = nullptr, *d_tp = nullptr, *d_tc = nullptr, *d_lc = nullptr, *d_si = nullptr;
    // (the per-nonzero lists are read by k_csr_tile_pass only: ring-sized tiles carry them inside their records instead)
    const bool lists_on_device = tiles_mode() != 2;
    hipError_t te = hipMalloc((void **)&d_tr, tile_row.size() * 4);
    if (te == hipSuccess) te = hipMalloc((void **)&d_tp, tp.size() * 4);
    if (te == hipSuccess && lists_on_device) te = hipMalloc((void **)&d_tc, tc.size() * 4);
    if (te == hipSuccess && lists_on_device) te = hipMalloc((void **)&d_lc, lc.size() * 4);
    if (te == hipSuccess && lists_on_device) te = hipMalloc((void **)&d_si, si.size() * 4);
    op->tiles.tile_row = d_tr;
    op->tiles.tile_ptr = d_tp;
    op->tiles.tile_cols = d_tc;
    op->tiles.lcol = d_lc;
    op->tiles.self_idx = d_si;
    if (te == hipSuccess) {
      std::vector<UploadQueue::Job> jobs = {{d_tr, tile_row.data(), tile_row.size() * 4}, {d_tp, tp.data(), tp.size() * 4}};
      if (lists_on_device) {
        jobs.push_back({d_tc, tc.data(), tc.size() * 4});
        jobs.push_back({d_lc, lc.data(), lc.size() * 4});
        jobs.push_back({d_si, si.data(), si.size() * 4});
      }
      up.push(std::move(jobs));
    }
    for (int x = 0; x < 9; ++x) op->tiles.xcd_tile[x] = xcd_tile[x];
    op->tiles.max_cols = mx;
    op->tiles_ringed = tiles_mode() == 2;
    if (op->tiles_ringed) op->merged_lock = new (std::nothrow) std::mutex();
    if (te == hipSuccess && op->tiles_ringed) {
      if (dtype == SLQ_F64) build_ring_stream<double>(1, rowptr, (const double *)vals, tile_row, tp, tc, lc, si, desc, rec);
      else build_ring_stream<float>(1, rowptr, (const float *)vals, tile_row, tp, tc, lc, si, desc, rec);
      clk.lap("  tile stream");
      te = hipMalloc((void **)&op->tile_desc, desc.size() * 4);
      if (te == hipSuccess) te = hipMalloc((void **)&op->tile_rec, rec.size());
      op->tile_desc_bytes = desc.size() * 4, op->tile_rec_bytes = rec.size();
      if (te == hipSuccess) up.push({{op->tile_desc, desc.data(), desc.size() * 4}, {op->tile_rec, rec.data(), rec.size()}});
      if (te == hipSuccess && sym) {
        // the same tiles over the upper triangle (doubled off-diagonals), for the alpha-only pass: a tile's image then holds its
        // own rows and the neighbours of HIGHER index only - about half the halo, and the pass is bound by what it lands by DMA
        int mxu = 0;
        // (its own, longer tiles: runs of the base tiles - the pass pays per tile, regroup_upper_tiles; SLQ_RING_UPPER_REGROUP=0 keeps the base tiles)
        std::vector<int32_t> tile_row_u;
        // (not where the base tiles are as tall as a tile gets - a 5-point grid's 13.9 of 14 rows: nothing to join, 10-20 ms of host time saved)
        const bool tall_already = (double)n / (double)(tile_row.size() - 1) > 0.8 * kRingTileRows;
        if (env_int("SLQ_RING_UPPER_REGROUP", 1) != 0 && !tall_already) {
          regroup_upper_tiles(urp.data(), uci.data(), tile_row, xcd_tile, tile_row_u, op->xcd_tile_u);
        } else {
          tile_row_u = tile_row;
          for (int x = 0; x < 9; ++x) op->xcd_tile_u[x] = xcd_tile[x];
        }
        if (env_int("SLQ_DEBUG", 0) != 0)
          fprintf(stderr, "[slq] tiles: upper-triangle stream on %zu tiles of %.2f rows (base: %zu of %.2f)\n", tile_row_u.size() - 1, (double)n / (double)(tile_row_u.size() - 1),
                  tile_row.size() - 1, (double)n / (double)(tile_row.size() - 1));
        build_tile_meta(n, urp.data(), uci.data(), tile_row_u, tpu, tcu, lcu, siu, &mxu);
        clk.lap("  upper tile lists");
        // Worth it while the tiles land at most kTileAlphaColsPerRow panel rows per row (r03, scalar-descriptor loaders and the
        // padded-row consumer of slq_ring.hpp: 5-point grid, 1.5 rows per row: 0.40 against 0.51 ms for the generic pass; 7-point
        // grid, 2.5: 0.66 against 0.81 ms)
        const double upper_per_row = (double)(tcu.size() - kCsrPad) / (double)n;
        if (env_int("SLQ_DEBUG", 0) != 0) fprintf(stderr, "[slq] tiles: upper triangle: %.2f distinct panel rows per row, longest list %d (full rows: %d)\n", upper_per_row, mxu, mx);
        op->tile_max_lines_u = mxu;
        op->upper_per_row = upper_per_row;
        // (built up to kTileAlphaMergedColsPerRow: wide panels take it up to kTileAlphaColsPerRow, slq_plan_create; the merged
        // tiles of narrow panels share more of their halo and gain from it on 7-point grids too - 100^3, 64 probes: alpha pass
        // 0.25 against 0.35 ms for the generic upper-triangle pass)
        if (upper_per_row <= kTileAlphaMergedColsPerRow) {
          bool pad = env_int("SLQ_RING_PAD_ROWS", 1) != 0;
          if (dtype == SLQ_F64) build_ring_stream<double>(1, urp.data(), (const double *)uva.data(), tile_row_u, tpu, tcu, lcu, siu, desc_u, rec_u, &pad);
          else build_ring_stream<float>(1, urp.data(), (const float *)uva.data(), tile_row_u, tpu, tcu, lcu, siu, desc_u, rec_u, &pad);
          op->tile_u_padded = pad;
          clk.lap("  upper tile stream");
          te = hipMalloc((void **)&op->tile_desc_u, desc_u.size() * 4);
          if (te == hipSuccess) te = hipMalloc((void **)&op->tile_rec_u, rec_u.size());
          op->tile_desc_u_bytes = desc_u.size() * 4, op->tile_rec_u_bytes = rec_u.size();
          if (te == hipSuccess) up.push({{op->tile_desc_u, desc_u.data(), desc_u.size() * 4}, {op->tile_rec_u, rec_u.data(), rec_u.size()}});
        }
      }
    }
    if (te != hipSuccess) return bail(SLQ_EHIP, "tile upload", te);
  }
  e = up.wait();
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) return bail(SLQ_EHIP, "operator upload", e);
  clk.lap("uploads drained");
  clk.total("all of slq_csr_create");
  guard.op = nullptr;
  *out = op;
  return SLQ_OK;
}

// The second half of slq_csr_create for operators with ring-sized tiles (SLQ_DEVICE_BUILD, slq_build.hpp): given the order and the
// tiles (host), everything stored with the operator is built on the device from the caller's CSR (o_rp, o_ci, o_va: on their way up
// through `early`): the permuted CSR, the far-gather count, the symmetry check and the upper triangle, both tile streams. The one
// sequential piece left - the runs of base tiles the upper-triangle stream's tiles are made of (regroup_upper_tiles) - runs on a host
// thread meanwhile, on the caller's CSR seen through the permutation.
static int csr_finish_on_device(slq_context *ctx, slq_operator *op, int dtype, int64_t n, int64_t nnz, const int32_t *rowptr0, const int32_t *colind0,
                                const int32_t *o_rp, const int32_t *o_ci, const void *o_va, UploadQueue &early, const std::vector<int32_t> &tile_row,
                                const int32_t xcd_tile[9], PhaseClock &clk) {
  hipStream_t st = ctx->stream;
  const size_t es = esize(dtype);
  const std::vector<int32_t> &perm = *op->perm_h;
  const int ntiles = (int)tile_row.size() - 1;
  std::vector<int32_t> inv((size_t)n), rp2((size_t)n + 1);
  for (int64_t i = 0; i < n; ++i) inv[(size_t)perm[(size_t)i]] = (int32_t)i;
  rp2[0] = 0;
  for (int64_t i = 0; i < n; ++i) {
    const int32_t o = perm[(size_t)i];
    rp2[(size_t)i + 1] = rp2[(size_t)i] + (rowptr0[o + 1] - rowptr0[o]);
  }
  const bool want_sym = env_int("SLQ_SYM_ALPHA", 1) != 0;
  const bool tall_already = (double)n / (double)ntiles > 0.8 * kRingTileRows;
  const bool regroup = want_sym && env_int("SLQ_RING_UPPER_REGROUP", 1) != 0 && !tall_already;
  std::vector<int32_t> tile_row_u;
  int32_t xcd_u[9];
  for (int x = 0; x < 9; ++x) xcd_u[x] = xcd_tile[x];
  std::atomic<int> rg_failed{0};
  auto do_regroup = [&]() {
    try {
      regroup_upper_tiles_impl(
          [&](int32_t r, auto &consider) {
            const int32_t o = perm[(size_t)r];
            int32_t cnt = 0;
            for (int32_t q = rowptr0[o]; q < rowptr0[o + 1]; ++q) {
              const int32_t c = inv[(size_t)colind0[q]];
              if (c >= r) consider(c), ++cnt;
            }
            return cnt;
          },
          tile_row, xcd_tile, tile_row_u, xcd_u);
    } catch (...) {
      rg_failed = 1;
    }
  };
  std::thread rg;
  struct Joiner {
    std::thread &t;
    ~Joiner() {
      if (t.joinable()) t.join();
    }
  } joiner{rg};
  if (regroup) {
    try {
      rg = std::thread(do_regroup);
    } catch (const std::system_error &) {
      do_regroup();
    }
  }
  auto hip_fail = [&](const char *what, hipError_t e) { return fail(e == hipErrorOutOfMemory ? SLQ_ENOMEM : SLQ_EHIP, "%s: %s", what, hipGetErrorString(e)); };
  int32_t *d_tr = nullptr;
  hipError_t e = hipMalloc((void **)&op->perm_d, (size_t)n * 4);
  if (e == hipSuccess) e = hipMalloc((void **)&op->inv_perm_d, (size_t)n * 4);
  if (e == hipSuccess) e = hipMalloc((void **)&op->rowptr, (size_t)(n + 1) * 4);
  if (e == hipSuccess) e = hipMalloc((void **)&op->colind, ((size_t)nnz + kCsrPad) * 4);
  if (e == hipSuccess) e = hipMalloc(&op->vals, ((size_t)nnz + kCsrPad) * es);
  if (e == hipSuccess) e = hipMalloc((void **)&d_tr, tile_row.size() * 4);
  op->tiles.tile_row = d_tr;
  if (e == hipSuccess) e = hipMemcpy(op->perm_d, perm.data(), (size_t)n * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(op->inv_perm_d, inv.data(), (size_t)n * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(op->rowptr, rp2.data(), (size_t)(n + 1) * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_tr, tile_row.data(), tile_row.size() * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemsetAsync(op->colind + nnz, 0, kCsrPad * 4, st);
  if (e == hipSuccess) e = hipMemsetAsync((char *)op->vals + (size_t)nnz * es, 0, kCsrPad * es, st);
  if (e == hipSuccess) e = early.wait();
  if (e != hipSuccess) return hip_fail("CSR upload", e);
  clk.lap("  order, tile boundaries and the caller's CSR on the device");
  const dim3 grow((unsigned)((n + 255) / 256)), brow(256);
  if (dtype == SLQ_F64) slqb::k_permute_csr<double><<<grow, brow, 0, st>>>((int)n, o_rp, o_ci, (const double *)o_va, op->perm_d, op->inv_perm_d, op->rowptr, op->colind, (double *)op->vals);
  else slqb::k_permute_csr<float><<<grow, brow, 0, st>>>((int)n, o_rp, o_ci, (const float *)o_va, op->perm_d, op->inv_perm_d, op->rowptr, op->colind, (float *)op->vals);
  DevBuf misc;  // [0..1] far gathers (u64), [2] "not symmetric"
  e = misc.alloc(16);
  if (e == hipSuccess) e = hipMemsetAsync(misc.p, 0, 16, st);
  if (e != hipSuccess) return hip_fail("operator analysis", e);
  slqb::k_far_count<<<grow, brow, 0, st>>>((int)n, op->rowptr, op->colind, misc.as<unsigned long long>());
  if (want_sym) {
    e = hipMalloc((void **)&op->rowptr_u, (size_t)(n + 1) * 4);
    if (e == hipSuccess) e = hipMemsetAsync(op->rowptr_u, 0, 4, st);
    if (e != hipSuccess) return hip_fail("upper triangle", e);
    if (dtype == SLQ_F64) slqb::k_sym_count<double><<<grow, brow, 0, st>>>((int)n, op->rowptr, op->colind, (const double *)op->vals, op->rowptr_u, misc.as<int>() + 2);
    else slqb::k_sym_count<float><<<grow, brow, 0, st>>>((int)n, op->rowptr, op->colind, (const float *)op->vals, op->rowptr_u, misc.as<int>() + 2);
  }
  struct { unsigned long long far; int bad, spare; } h = {0, 0, 0};
  e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(&h, misc.p, 16, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) return hip_fail("operator analysis", e);
  op->far_per_row = (double)h.far / (double)n;
  bool sym = want_sym && !h.bad;
  if (want_sym && !sym) {
    hipFree(op->rowptr_u);
    op->rowptr_u = nullptr;
  }
  if (sym) {
    e = device_scan_inclusive(op->rowptr_u + 1, n, st);
    int32_t nu32 = 0;
    if (e == hipSuccess) e = hipMemcpy(&nu32, op->rowptr_u + n, 4, hipMemcpyDeviceToHost);
    const size_t nu = (size_t)nu32;
    op->nnz_u = (int64_t)nu;
    if (e == hipSuccess) e = hipMalloc((void **)&op->colind_u, (nu + kCsrPad) * 4);
    if (e == hipSuccess) e = hipMalloc(&op->vals_u, (nu + kCsrPad) * es);
    if (e == hipSuccess) e = hipMemsetAsync(op->colind_u + nu, 0, kCsrPad * 4, st);
    if (e == hipSuccess) e = hipMemsetAsync((char *)op->vals_u + nu * es, 0, kCsrPad * es, st);
    if (e != hipSuccess) return hip_fail("upper triangle", e);
    if (dtype == SLQ_F64) slqb::k_upper_fill<double><<<grow, brow, 0, st>>>((int)n, op->rowptr, op->colind, (const double *)op->vals, op->rowptr_u, op->colind_u, (double *)op->vals_u);
    else slqb::k_upper_fill<float><<<grow, brow, 0, st>>>((int)n, op->rowptr, op->colind, (const float *)op->vals, op->rowptr_u, op->colind_u, (float *)op->vals_u);
  }
  if (env_int("SLQ_DEBUG", 0) != 0)
    fprintf(stderr, "[slq] csr n=%lld nnz=%lld reordered=1 rms in-chunk |i-j| = %.1f, far gathers per row %.2f (built on the device)\n", (long long)n, (long long)nnz,
            op->rms_dist, op->far_per_row);
  clk.lap("  device: stored CSR, upper triangle");
  // the tiles' stream over the full rows
  DeviceStream full;
  int rc = device_build_stream(ctx, dtype, 1, n, op->rowptr, op->colind, op->vals, d_tr, ntiles, false, 0.0, true, full);
  if (rc != 0) return rc < 0 ? rc : fail(SLQ_EHIP, "tile stream declined");
  op->tile_desc = full.desc, op->tile_rec = full.rec;
  op->tile_desc_bytes = full.desc_bytes, op->tile_rec_bytes = full.rec_bytes;
  op->tiles.tile_ptr = full.tile_ptr;
  op->tiles.max_cols = full.max_lines;
  for (int x = 0; x < 9; ++x) op->tiles.xcd_tile[x] = xcd_tile[x];
  op->tiles_ringed = true;
  op->merged_lock = new (std::nothrow) std::mutex();
  clk.lap("  device: tile stream");
  if (sym) {
    if (rg.joinable()) rg.join();
    if (rg_failed) return fail(SLQ_ENOMEM, "host worker failed (upper-triangle tiles)");
    if (!regroup) tile_row_u = tile_row;
    for (int x = 0; x < 9; ++x) op->xcd_tile_u[x] = xcd_u[x];
    const int ntu = (int)tile_row_u.size() - 1;
    if (env_int("SLQ_DEBUG", 0) != 0)
      fprintf(stderr, "[slq] tiles: upper-triangle stream on %d tiles of %.2f rows (base: %d of %.2f)\n", ntu, (double)n / (double)ntu, ntiles, (double)n / (double)ntiles);
    DevBuf d_tru;
    e = d_tru.alloc(tile_row_u.size() * 4);
    if (e == hipSuccess) e = hipMemcpy(d_tru.p, tile_row_u.data(), tile_row_u.size() * 4, hipMemcpyHostToDevice);
    if (e != hipSuccess) return hip_fail("upper-triangle tiles", e);
    DeviceStream us;
    rc = device_build_stream(ctx, dtype, 1, n, op->rowptr_u, op->colind_u, op->vals_u, d_tru.as<int32_t>(), ntu, env_int("SLQ_RING_PAD_ROWS", 1) != 0,
                             kTileAlphaMergedColsPerRow, false, us);
    if (rc < 0) return rc;
    op->tile_max_lines_u = us.max_lines;
    op->upper_per_row = (double)us.sum_lines / (double)n;
    if (env_int("SLQ_DEBUG", 0) != 0)
      fprintf(stderr, "[slq] tiles: upper triangle: %.2f distinct panel rows per row, longest list %d (full rows: %d)\n", op->upper_per_row, us.max_lines, full.max_lines);
    if (rc == 0) {
      op->tile_desc_u = us.desc, op->tile_rec_u = us.rec;
      op->tile_desc_u_bytes = us.desc_bytes, op->tile_rec_u_bytes = us.rec_bytes;
      op->tile_u_padded = us.padded;
    }
    clk.lap("  device: upper tile stream");
  }
  e = hipStreamSynchronize(st);
  if (e != hipSuccess) return hip_fail("operator build", e);
  return SLQ_OK;
}

// SLQ_DEVICE_BUILD=2: 0 when everything two operators over the same matrix keep is the same, else the number of the first item that is not
static int operators_differ(const slq_operator *a, const slq_operator *b) {
  const size_t es = esize(a->dtype);
  auto same = [](const void *x, const void *y, size_t bytes) {
    if (!x || !y) return x == y;
    std::vector<char> hx(bytes), hy(bytes);
    if (hipMemcpy(hx.data(), x, bytes, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(hy.data(), y, bytes, hipMemcpyDeviceToHost) != hipSuccess) return false;
    return memcmp(hx.data(), hy.data(), bytes) == 0;
  };
  const size_t n = (size_t)a->n, nnz = (size_t)a->nnz;
  if (a->n != b->n || a->nnz != b->nnz || a->nnz_u != b->nnz_u || a->dtype != b->dtype) return 1;
  if (!a->perm_h || !b->perm_h || *a->perm_h != *b->perm_h) return 2;
  if (!same(a->perm_d, b->perm_d, n * 4) || !same(a->inv_perm_d, b->inv_perm_d, n * 4)) return 3;
  if (!same(a->rowptr, b->rowptr, (n + 1) * 4)) return 4;
  if (!same(a->colind, b->colind, (nnz + kCsrPad) * 4)) return 5;
  if (!same(a->vals, b->vals, (nnz + kCsrPad) * es)) return 6;
  if (a->far_per_row != b->far_per_row || a->rms_dist != b->rms_dist) return 7;
  if ((a->rowptr_u == nullptr) != (b->rowptr_u == nullptr)) return 8;
  if (a->rowptr_u) {
    const size_t nu = (size_t)a->nnz_u;
    if (!same(a->rowptr_u, b->rowptr_u, (n + 1) * 4)) return 9;
    if (!same(a->colind_u, b->colind_u, (nu + kCsrPad) * 4)) return 10;
    if (!same(a->vals_u, b->vals_u, (nu + kCsrPad) * es)) return 11;
  }
  for (int x = 0; x < 9; ++x)
    if (a->tiles.xcd_tile[x] != b->tiles.xcd_tile[x] || a->xcd_tile_u[x] != b->xcd_tile_u[x]) return 12;
  const size_t nt = (size_t)a->tiles.xcd_tile[8];
  if (!same(a->tiles.tile_row, b->tiles.tile_row, (nt + 1) * 4) || !same(a->tiles.tile_ptr, b->tiles.tile_ptr, (nt + 1) * 4)) return 13;
  if (a->tiles.max_cols != b->tiles.max_cols || a->tiles_ringed != b->tiles_ringed) return 14;
  if (a->tile_desc_bytes != b->tile_desc_bytes || a->tile_rec_bytes != b->tile_rec_bytes) return 15;
  if (!same(a->tile_desc, b->tile_desc, a->tile_desc_bytes)) return 16;
  if (!same(a->tile_rec, b->tile_rec, a->tile_rec_bytes)) return 17;
  if (a->tile_max_lines_u != b->tile_max_lines_u || a->upper_per_row != b->upper_per_row || a->tile_u_padded != b->tile_u_padded) return 18;
  if ((a->tile_desc_u == nullptr) != (b->tile_desc_u == nullptr)) return 19;
  if (a->tile_desc_u) {
    if (a->tile_desc_u_bytes != b->tile_desc_u_bytes || a->tile_rec_u_bytes != b->tile_rec_u_bytes) return 20;
    if (!same(a->tile_desc_u, b->tile_desc_u, a->tile_desc_u_bytes)) return 21;
    if (!same(a->tile_rec_u, b->tile_rec_u, a->tile_rec_u_bytes)) return 22;
  }
  return 0;
}

extern "C" int slq_csr_create_device(slq_context *ctx, int dtype, int64_t n, int64_t nnz,
                                     const int32_t *d_rowptr, const int32_t *d_colind,
                                     const void *d_vals, slq_operator **out) {
  if (!ctx || !out) return fail(SLQ_EINVAL, "ctx/out is NULL");
  *out = nullptr;
  SLQ_TRY(check_dtype(dtype));
  if (n <= 0 || n >= (int64_t)1 << 31 || nnz < 0 || nnz >= (int64_t)1 << 31)
    return fail(SLQ_EINVAL, "CSR shape out of range for int32 indices");
  if (!d_rowptr || (nnz > 0 && (!d_colind || !d_vals))) return fail(SLQ_EINVAL, "CSR arrays are NULL");
  HIP_TRY(hipSetDevice(ctx->device));
  if (ctx->dead) return fail(SLQ_EINVAL, "the context has been destroyed");
  // The operator is built the way slq_csr_create builds it - validated, reordered, with its upper triangle and its tiles. Order and
  // tiles are decided on the host, from the index arrays (4 bytes per nonzero come back, once); what the operator stores is then built
  // on the device straight from the caller's arrays (slq_build.hpp) - the values come back only for operators without ring tiles, whose
  // storage the host builds. The caller's arrays are not referenced after the call.
  std::vector<int32_t> rp, ci;
  try {
    rp.resize((size_t)n + 1);
    ci.resize((size_t)nnz);
  } catch (const std::bad_alloc &) {
    return fail(SLQ_ENOMEM, "host allocation failed");
  }
  HIP_TRY(hipMemcpy(rp.data(), d_rowptr, ((size_t)n + 1) * 4, hipMemcpyDeviceToHost));
  if (nnz) HIP_TRY(hipMemcpy(ci.data(), d_colind, (size_t)nnz * 4, hipMemcpyDeviceToHost));
  DeviceCsr dev;
  dev.rp = d_rowptr, dev.ci = d_colind, dev.va = d_vals;
  return csr_create_impl(ctx, dtype, n, nnz, rp.data(), ci.data(), nullptr, out, 0, 0, dev);
}

extern "C" int slq_csr_gram_create(slq_context *ctx, int dtype, int64_t mrows, int64_t ncols, int64_t nnz, const int32_t *rowptr,
                                   const int32_t *colind, const void *vals, slq_operator **out) {
  if (!ctx || !out) return fail(SLQ_EINVAL, "ctx/out is NULL");
  *out = nullptr;
  SLQ_TRY(check_dtype(dtype));
  if (mrows <= 0 || ncols <= 0 || mrows >= (int64_t)1 << 31 || ncols >= (int64_t)1 << 31 || nnz < 0 || nnz >= (int64_t)1 << 31)
    return fail(SLQ_EINVAL, "CSR shape out of range for int32 indices");
  if (!rowptr || (nnz > 0 && (!colind || !vals))) return fail(SLQ_EINVAL, "CSR arrays are NULL");
  if (rowptr[0] != 0 || rowptr[mrows] != nnz) return fail(SLQ_EINVAL, "rowptr[0] must be 0 and rowptr[mrows] must equal nnz");
  for (int64_t i = 0; i < mrows; ++i)
    if (rowptr[i + 1] < rowptr[i]) return fail(SLQ_EINVAL, "rowptr is not non-decreasing at %lld", (long long)i);
  for (int64_t q = 0; q < nnz; ++q)
    if (colind[q] < 0 || colind[q] >= ncols) return fail(SLQ_EINVAL, "column index %d out of range at position %lld", colind[q], (long long)q);
  HIP_TRY(hipSetDevice(ctx->device));
  if (ctx->dead) return fail(SLQ_EINVAL, "the context has been destroyed");
  slq_operator *op = new (std::nothrow) slq_operator();
  if (!op) return fail(SLQ_ENOMEM, "host allocation failed");
  ctx_retain(ctx);
  *op = slq_operator{ctx, OP_GRAM, dtype, ncols, nnz, nullptr, nullptr, nullptr, 0, true, nullptr, nullptr, nullptr, nullptr, TileMeta{}};
  op->mrows = mrows;
  const size_t es = esize(dtype);
  // transpose on the host (counting sort by column; rows of A^T come out with ascending indices)
  std::vector<int32_t> tp((size_t)ncols + 1, 0), tc((size_t)nnz);
  std::vector<char> tv((size_t)nnz * es);
  for (int64_t q = 0; q < nnz; ++q) ++tp[(size_t)colind[q] + 1];
  for (int64_t c = 0; c < ncols; ++c) tp[(size_t)c + 1] += tp[(size_t)c];
  {
    std::vector<int32_t> cur(tp.begin(), tp.end() - 1);
    for (int64_t i = 0; i < mrows; ++i)
      for (int32_t q = rowptr[i]; q < rowptr[i + 1]; ++q) {
        const int32_t w = cur[(size_t)colind[q]]++;
        tc[(size_t)w] = (int32_t)i;
        memcpy(tv.data() + (size_t)w * es, (const char *)vals + (size_t)q * es, es);
      }
  }
  hipError_t e = hipMalloc((void **)&op->rowptr, (size_t)(mrows + 1) * 4);
  if (e == hipSuccess) e = hipMalloc((void **)&op->colind, ((size_t)nnz + kCsrPad) * 4);
  if (e == hipSuccess) e = hipMalloc(&op->vals, ((size_t)nnz + kCsrPad) * es);
  if (e == hipSuccess) e = hipMalloc((void **)&op->rowptr_t, (size_t)(ncols + 1) * 4);
  if (e == hipSuccess) e = hipMalloc((void **)&op->colind_t, ((size_t)nnz + kCsrPad) * 4);
  if (e == hipSuccess) e = hipMalloc(&op->vals_t, ((size_t)nnz + kCsrPad) * es);
  if (e == hipSuccess) e = hipMemcpyAsync(op->rowptr, rowptr, (size_t)(mrows + 1) * 4, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess && nnz) e = hipMemcpyAsync(op->colind, colind, (size_t)nnz * 4, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess && nnz) e = hipMemcpyAsync(op->vals, vals, (size_t)nnz * es, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(op->rowptr_t, tp.data(), (size_t)(ncols + 1) * 4, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess && nnz) e = hipMemcpyAsync(op->colind_t, tc.data(), (size_t)nnz * 4, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess && nnz) e = hipMemcpyAsync(op->vals_t, tv.data(), (size_t)nnz * es, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) {
    slq_operator_destroy(op);
    return fail(e == hipErrorOutOfMemory ? SLQ_ENOMEM : SLQ_EHIP, "Gram operator upload: %s", hipGetErrorString(e));
  }
  *out = op;
  return SLQ_OK;
}

// Affine sparse operator A + t B (eigen_operators.h:106-137, SparseEigenAffineOperator): both n x n CSR. The operator is an
// ordinary CSR operator on the UNION pattern - every fused pass applies - whose values are va + t vb, recomputed on the
// device by slq_operator_set_parameter (t = 0 at creation, like the reference's _param).
extern "C" int slq_csr_affine_create(slq_context *ctx, int dtype, int64_t n, int64_t nnz_a, const int32_t *rp_a, const int32_t *ci_a,
                                     const void *va, int64_t nnz_b, const int32_t *rp_b, const int32_t *ci_b, const void *vb,
                                     slq_operator **out) {
  if (!ctx || !out) return fail(SLQ_EINVAL, "ctx/out is NULL");
  *out = nullptr;
  SLQ_TRY(check_dtype(dtype));
  if (n <= 0 || n >= (int64_t)1 << 31 || nnz_a < 0 || nnz_b < 0 || nnz_a >= (int64_t)1 << 31 || nnz_b >= (int64_t)1 << 31)
    return fail(SLQ_EINVAL, "CSR shape out of range for int32 indices");
  if (!rp_a || !rp_b || (nnz_a > 0 && (!ci_a || !va)) || (nnz_b > 0 && (!ci_b || !vb))) return fail(SLQ_EINVAL, "CSR arrays are NULL");
  if (rp_a[0] != 0 || rp_b[0] != 0 || rp_a[n] != nnz_a || rp_b[n] != nnz_b)
    return fail(SLQ_EINVAL, "rowptr[0] must be 0 and rowptr[n] must equal nnz");
  for (int64_t i = 0; i < n; ++i)
    if (rp_a[i + 1] < rp_a[i] || rp_b[i + 1] < rp_b[i]) return fail(SLQ_EINVAL, "rowptr is not non-decreasing at %lld", (long long)i);
  const size_t es = esize(dtype);
  std::vector<int32_t> rp, ci;
  std::vector<char> ua, ub;
  std::vector<std::pair<int32_t, int>> a_row, b_row;
  const char zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  try {
  rp.assign((size_t)n + 1, 0);
  ci.reserve((size_t)std::max(nnz_a, nnz_b));
  for (int64_t i = 0; i < n; ++i) {
    a_row.clear();
    b_row.clear();
    for (int32_t q = rp_a[i]; q < rp_a[i + 1]; ++q) a_row.emplace_back(ci_a[q], q);
    for (int32_t q = rp_b[i]; q < rp_b[i + 1]; ++q) b_row.emplace_back(ci_b[q], q);
    std::sort(a_row.begin(), a_row.end());
    std::sort(b_row.begin(), b_row.end());
    size_t x = 0, y = 0;
    while (x < a_row.size() || y < b_row.size()) {
      const int32_t ca = x < a_row.size() ? a_row[x].first : INT32_MAX, cb = y < b_row.size() ? b_row[y].first : INT32_MAX;
      const int32_t c = std::min(ca, cb);
      if (c < 0 || c >= n) return fail(SLQ_EINVAL, "column index out of range in the affine operator");
      ci.push_back(c);
      ua.insert(ua.end(), zero, zero + es);
      ub.insert(ub.end(), zero, zero + es);
      if (ca == c) { memcpy(ua.data() + ua.size() - es, (const char *)va + (size_t)a_row[x].second * es, es); ++x; }
      if (cb == c) { memcpy(ub.data() + ub.size() - es, (const char *)vb + (size_t)b_row[y].second * es, es); ++y; }
    }
    if (ci.size() >= ((size_t)1 << 31)) return fail(SLQ_EINVAL, "the union pattern exceeds int32 indices");
    rp[(size_t)i + 1] = (int32_t)ci.size();
  }
  } catch (const std::bad_alloc &) {  // no C++ exception crosses the C boundary
    return fail(SLQ_ENOMEM, "host allocation failed");
  }
  const int64_t nnz = (int64_t)ci.size();
  // the union pattern with A's values is an ordinary CSR operator; keep its rows as given (no reordering, no upper-triangle
  // alpha pass: both would have to follow every parameter change)
  slq_operator *op = nullptr;
  const int rc = csr_create_impl(ctx, dtype, n, nnz, rp.data(), ci.data(), ua.data(), &op, 1);
  if (rc != SLQ_OK) return rc;
  hipError_t e = hipMalloc(&op->vals_a, std::max<size_t>((size_t)nnz * es, 8));
  if (e == hipSuccess) e = hipMalloc(&op->vals_b, std::max<size_t>((size_t)nnz * es, 8));
  if (e == hipSuccess && nnz) e = hipMemcpyAsync(op->vals_a, ua.data(), (size_t)nnz * es, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess && nnz) e = hipMemcpyAsync(op->vals_b, ub.data(), (size_t)nnz * es, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) {
    slq_operator_destroy(op);
    return fail(e == hipErrorOutOfMemory ? SLQ_ENOMEM : SLQ_EHIP, "affine operator upload: %s", hipGetErrorString(e));
  }
  *out = op;
  return SLQ_OK;
}

// t of an affine operator A + t B (SparseEigenAffineOperator::set_parameter, eigen_operators.h:134-136)
extern "C" int slq_operator_set_parameter(slq_operator *op, double t) {
  if (!op) return fail(SLQ_EINVAL, "op is NULL");
  if (!op->vals_a || !op->vals_b) return fail(SLQ_EINVAL, "not an affine operator");
  HIP_TRY(hipSetDevice(op->ctx->device));
  const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(4096, (op->nnz + 255) / 256));
  if (op->dtype == SLQ_F64)
    k_affine_vals<double><<<grid, 256, 0, op->ctx->stream>>>(op->nnz, (const double *)op->vals_a, (const double *)op->vals_b, t, (double *)op->vals);
  else
    k_affine_vals<float><<<grid, 256, 0, op->ctx->stream>>>(op->nnz, (const float *)op->vals_a, (const float *)op->vals_b, t, (float *)op->vals);
  HIP_TRY(hipGetLastError());
  return SLQ_OK;
}

extern "C" int slq_dense_create(slq_context *ctx, int dtype, int64_t n, const void *A, int64_t lda,
                                slq_operator **out) {
  if (!ctx || !out) return fail(SLQ_EINVAL, "ctx/out is NULL");
  *out = nullptr;
  SLQ_TRY(check_dtype(dtype));
  if (n <= 0 || n >= (int64_t)1 << 31 || !A || lda < n) return fail(SLQ_EINVAL, "bad dense operator shape");
  HIP_TRY(hipSetDevice(ctx->device));
  if (ctx->dead) return fail(SLQ_EINVAL, "the context has been destroyed");
  slq_operator *op = new (std::nothrow) slq_operator();
  if (!op) return fail(SLQ_ENOMEM, "host allocation failed");
  ctx_retain(ctx);
  *op = slq_operator{ctx, OP_DENSE, dtype, n, n * n, nullptr, nullptr, nullptr, n, true, nullptr, nullptr, nullptr, nullptr, TileMeta{}};
  const size_t es = esize(dtype);
  // Y = A X for whatever is given (eigen_operators.h:24-30 does not ask for symmetry either). k_dense_mfma_3term reads
  // A(row, k) and is right for any A; k_dense_panel walks row `row` of A as the contiguous COLUMN `row`, which is A^T:
  // equal for the symmetric operators Lanczos is meant for. An O(n^2) host pass checks that; a non-symmetric input
  // gets its transpose uploaded next to it for that kernel.
  bool symmetric = true;
  for (int64_t j = 0; j < n && symmetric; ++j)
    for (int64_t i = j + 1; i < n; ++i) {
      const bool same = dtype == SLQ_F64 ? ((const double *)A)[j * lda + i] == ((const double *)A)[i * lda + j]
                                         : ((const float *)A)[j * lda + i] == ((const float *)A)[i * lda + j];
      if (!same) { symmetric = false; break; }
    }
  hipError_t e = hipMalloc(&op->vals, (size_t)n * n * es);
  if (e == hipSuccess)
    e = hipMemcpy2DAsync(op->vals, (size_t)n * es, A, (size_t)lda * es, (size_t)n * es, (size_t)n,
                         hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess && !symmetric) {
    std::vector<char> T;
    try {
      T.resize((size_t)n * n * es);
    } catch (const std::bad_alloc &) {
      slq_operator_destroy(op);
      return fail(SLQ_ENOMEM, "host allocation failed");
    }
    for (int64_t j = 0; j < n; ++j)
      for (int64_t i = 0; i < n; ++i)
        memcpy(T.data() + ((size_t)j * n + i) * es, (const char *)A + ((size_t)i * lda + j) * es, es);  // T(i,j) = A(j,i)
    e = hipMalloc(&op->vals_t, (size_t)n * n * es);
    if (e == hipSuccess) e = hipMemcpyAsync(op->vals_t, T.data(), (size_t)n * n * es, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) {
    slq_operator_destroy(op);
    return fail(e == hipErrorOutOfMemory ? SLQ_ENOMEM : SLQ_EHIP, "dense upload: %s", hipGetErrorString(e));
  }
  *out = op;
  return SLQ_OK;
}

extern "C" int slq_callback_create(slq_context *ctx, int dtype, int64_t n, slq_matvec_fn fn,
                                   void *user, slq_operator **out) {
  if (!ctx || !out) return fail(SLQ_EINVAL, "ctx/out is NULL");
  *out = nullptr;
  SLQ_TRY(check_dtype(dtype));
  if (n <= 0 || n >= (int64_t)1 << 31) return fail(SLQ_EINVAL, "bad operator shape");
  if (!fn) return fail(SLQ_EINVAL, "Supplied object is missing 'matvec' attribute.");
  if (ctx->dead) return fail(SLQ_EINVAL, "the context has been destroyed");
  slq_operator *op = new (std::nothrow) slq_operator();
  if (!op) return fail(SLQ_ENOMEM, "host allocation failed");
  ctx_retain(ctx);
  *op = slq_operator{ctx, OP_CALLBACK, dtype, n, 0, nullptr, nullptr, nullptr, 0, false, fn, user, nullptr, nullptr, TileMeta{}};
  *out = op;
  return SLQ_OK;
}

extern "C" int slq_device_callback_create(slq_context *ctx, int dtype, int64_t n, slq_matmat_device_fn fn, void *user,
                                          slq_operator **out) {
  if (!ctx || !out || !fn) return fail(SLQ_EINVAL, "ctx/out/fn is NULL");
  *out = nullptr;
  SLQ_TRY(check_dtype(dtype));
  if (n <= 0 || n >= (int64_t)1 << 31) return fail(SLQ_EINVAL, "bad operator size");
  if (ctx->dead) return fail(SLQ_EINVAL, "the context has been destroyed");
  slq_operator *op = new (std::nothrow) slq_operator();
  if (!op) return fail(SLQ_ENOMEM, "host allocation failed");
  ctx_retain(ctx);
  *op = slq_operator{ctx, OP_DEVICE_CALLBACK, dtype, n, 0, nullptr, nullptr, nullptr, 0, false, nullptr, user, nullptr, nullptr, TileMeta{}};
  op->dev_fn = fn;
  *out = op;
  return SLQ_OK;
}

extern "C" int slq_operator_destroy(slq_operator *op) {
  if (!op) return SLQ_OK;
  if (op->owns) {
    hipSetDevice(op->ctx->device);
    if (op->rowptr) hipFree(op->rowptr);
    if (op->colind) hipFree(op->colind);
    if (op->vals) hipFree(op->vals);
  }
  if (op->vals_t) hipFree(op->vals_t);
  if (op->rowptr_t) hipFree(op->rowptr_t);
  if (op->colind_t) hipFree(op->colind_t);
  if (op->vals_a) hipFree(op->vals_a);
  if (op->vals_b) hipFree(op->vals_b);
  if (op->perm_d) hipFree(op->perm_d);
  if (op->inv_perm_d) hipFree(op->inv_perm_d);
  delete op->perm_h;
  if (op->rowptr_u) hipFree(op->rowptr_u);
  if (op->colind_u) hipFree(op->colind_u);
  if (op->vals_u) hipFree(op->vals_u);
  if (op->tiles.tile_row) hipFree((void *)op->tiles.tile_row);
  if (op->tiles.tile_ptr) hipFree((void *)op->tiles.tile_ptr);
  if (op->tiles.tile_cols) hipFree((void *)op->tiles.tile_cols);
  if (op->tiles.lcol) hipFree((void *)op->tiles.lcol);
  if (op->tiles.self_idx) hipFree((void *)op->tiles.self_idx);
  if (op->tile_desc) hipFree(op->tile_desc);
  if (op->tile_rec) hipFree(op->tile_rec);
  if (op->tile_desc_u) hipFree(op->tile_desc_u);
  if (op->tile_rec_u) hipFree(op->tile_rec_u);
  for (void *q : {(void *)op->fa.desc, (void *)op->fa.rec, (void *)op->fa.er, (void *)op->fa.ec, op->fa.ev})
    if (q) hipFree(q);
  for (auto &m : op->merged) {
    if (m.desc) hipFree(m.desc);
    if (m.rec) hipFree(m.rec);
    if (m.desc_u) hipFree(m.desc_u);
    if (m.rec_u) hipFree(m.rec_u);
  }
  delete op->merged_lock;
  ctx_release(op->ctx);
  delete op;
  return SLQ_OK;
}

// Tiles of R = 2 or 4 merged base tiles for the narrow-panel form of the ring-fed passes (slq_ring.hpp), built the first time
// a plan needs them and kept with the operator. A merged tile is R consecutive tiles of one XCD chunk - neighbours in the
// sweep, so most of what they read they share - with ONE line list (build_tile_meta on the merged row ranges), hence
// R x 14 rows, at most R x 36 lines and R x 112 nonzeros: exactly what a slot of that kernel holds. Everything comes
// from what the operator already keeps on the device (its CSR in stored order, the tile boundaries); nothing of the
// caller's is needed again. Returns false when the operator has no ring-sized tiles (or the build failed: the plan then
// takes the generic passes).
static bool ensure_ring_stream(slq_operator *op, int R) {
  if (R == 1) return op->tile_desc != nullptr;
  if (!op->tile_desc || !op->tiles_ringed || !op->merged_lock || (R != 2 && R != 4)) return false;
  slq_operator::MergedStream &m = op->merged[R == 2 ? 0 : 1];
  std::lock_guard<std::mutex> guard(*op->merged_lock);
  if (m.tried) return m.desc != nullptr;
  m.tried = true;
  const int64_t n = op->n, nnz = op->nnz;
  const size_t es = esize(op->dtype);
  const int32_t ntiles = op->tiles.xcd_tile[8];
  auto drop = [](slq_operator::MergedStream &mm) {
    for (void **q : {(void **)&mm.desc, (void **)&mm.rec, (void **)&mm.desc_u, (void **)&mm.rec_u}) {
      if (*q) hipFree(*q);
      *q = nullptr;
    }
  };
  try {
    std::vector<int32_t> tr((size_t)ntiles + 1);
    if (hipMemcpy(tr.data(), op->tiles.tile_row, ((size_t)ntiles + 1) * 4, hipMemcpyDeviceToHost) != hipSuccess) return false;
    // merged tile boundaries, chunk by chunk (a merged tile never straddles two XCD chunks; a chunk's last one may be short)
    std::vector<int32_t> mrow;
    for (int x = 0; x < 8; ++x) {
      m.xcd_tile[x] = (int32_t)mrow.size();
      for (int32_t t = op->tiles.xcd_tile[x]; t < op->tiles.xcd_tile[x + 1]; t += R) mrow.push_back(tr[(size_t)t]);
    }
    m.xcd_tile[8] = (int32_t)mrow.size();
    mrow.push_back((int32_t)n);
    // the host's way (SLQ_DEVICE_BUILD=0, and the yardstick of SLQ_DEVICE_BUILD=2): the CSR comes back, lists and stream are built
    // here and uploaded. sizes: bytes of desc, rec, desc_u, rec_u
    auto build_host = [&](slq_operator::MergedStream &mm, size_t sizes[4]) -> bool {
      std::vector<int32_t> rp((size_t)n + 1), ci((size_t)nnz);
      std::vector<char> va((size_t)nnz * es);
      if (hipMemcpy(rp.data(), op->rowptr, ((size_t)n + 1) * 4, hipMemcpyDeviceToHost) != hipSuccess) return false;
      if (hipMemcpy(ci.data(), op->colind, (size_t)nnz * 4, hipMemcpyDeviceToHost) != hipSuccess) return false;
      if (hipMemcpy(va.data(), op->vals, (size_t)nnz * es, hipMemcpyDeviceToHost) != hipSuccess) return false;
      auto upload = [&](const int32_t *rowptr, const int32_t *colind, const void *vals, int32_t **desc_d, char **rec_d, int *max_lines, bool *pad, size_t *sz) -> bool {
        std::vector<int32_t> tp, tc, lc, si;
        RawBuf<int32_t> desc;
        RawBuf<char> rec;
        int mx = 0;
        build_tile_meta(n, rowptr, colind, mrow, tp, tc, lc, si, &mx);
        *max_lines = mx;
        if (mx > kRingTileCols * R) return false;  // (cannot happen: a union of R lists of <= 36)
        if (op->dtype == SLQ_F64) build_ring_stream<double>(R, rowptr, (const double *)vals, mrow, tp, tc, lc, si, desc, rec, pad);
        else build_ring_stream<float>(R, rowptr, (const float *)vals, mrow, tp, tc, lc, si, desc, rec, pad);
        sz[0] = desc.size() * 4, sz[1] = rec.size();
        if (hipMalloc((void **)desc_d, desc.size() * 4) != hipSuccess) return false;
        if (hipMalloc((void **)rec_d, rec.size()) != hipSuccess) return false;
        return hipMemcpy(*desc_d, desc.data(), desc.size() * 4, hipMemcpyHostToDevice) == hipSuccess &&
               hipMemcpy(*rec_d, rec.data(), rec.size(), hipMemcpyHostToDevice) == hipSuccess;
      };
      bool ok = upload(rp.data(), ci.data(), va.data(), &mm.desc, &mm.rec, &mm.max_lines, nullptr, sizes);
      if (ok && op->tile_desc_u && op->rowptr_u) {
        const size_t nu = (size_t)op->nnz_u;
        bool upad = env_int("SLQ_RING_PAD_ROWS", 1) != 0;
        std::vector<int32_t> urp((size_t)n + 1), uci(nu);
        std::vector<char> uva(nu * es);
        ok = hipMemcpy(urp.data(), op->rowptr_u, ((size_t)n + 1) * 4, hipMemcpyDeviceToHost) == hipSuccess &&
             hipMemcpy(uci.data(), op->colind_u, nu * 4, hipMemcpyDeviceToHost) == hipSuccess &&
             hipMemcpy(uva.data(), op->vals_u, nu * es, hipMemcpyDeviceToHost) == hipSuccess &&
             upload(urp.data(), uci.data(), uva.data(), &mm.desc_u, &mm.rec_u, &mm.max_lines_u, &upad, sizes + 2);
        mm.u_padded = ok && upad;
      }
      if (!ok) drop(mm);
      return ok;
    };
    // on the device (slq_build.hpp): the operator's CSR never leaves it
    size_t dsz[4] = {0, 0, 0, 0};
    auto build_device = [&](slq_operator::MergedStream &mm) -> bool {
      if (hipSetDevice(op->ctx->device) != hipSuccess) return false;
      DevBuf d_mrow;
      if (d_mrow.alloc(mrow.size() * 4) != hipSuccess) return false;
      if (hipMemcpy(d_mrow.p, mrow.data(), mrow.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return false;
      const int nm = (int)mrow.size() - 1;
      DeviceStream f;
      if (device_build_stream(op->ctx, op->dtype, R, n, op->rowptr, op->colind, op->vals, d_mrow.as<int32_t>(), nm, false, 0.0, false, f) != 0) return false;
      mm.desc = f.desc, mm.rec = f.rec, mm.max_lines = f.max_lines;
      dsz[0] = f.desc_bytes, dsz[1] = f.rec_bytes;
      if (op->tile_desc_u && op->rowptr_u) {
        DeviceStream g;
        if (device_build_stream(op->ctx, op->dtype, R, n, op->rowptr_u, op->colind_u, op->vals_u, d_mrow.as<int32_t>(), nm, env_int("SLQ_RING_PAD_ROWS", 1) != 0, 0.0, false, g) != 0) {
          drop(mm);
          return false;
        }
        mm.desc_u = g.desc, mm.rec_u = g.rec, mm.max_lines_u = g.max_lines, mm.u_padded = g.padded;
        dsz[2] = g.desc_bytes, dsz[3] = g.rec_bytes;
      }
      return true;
    };
    const int dev_mode = env_int("SLQ_RING_ORDER", 0) != 0 ? 0 : env_int("SLQ_DEVICE_BUILD", 1);
    size_t hsz[4] = {0, 0, 0, 0};
    if (dev_mode == 0) return build_host(m, hsz);
    if (!build_device(m)) return false;
    if (dev_mode == 2) {  // both, compared
      slq_operator::MergedStream h;
      bool same = build_host(h, hsz);
      auto eq = [](const void *x, const void *y, size_t bytes) {
        if (!x || !y) return x == y;
        std::vector<char> hx(bytes), hy(bytes);
        return hipMemcpy(hx.data(), x, bytes, hipMemcpyDeviceToHost) == hipSuccess && hipMemcpy(hy.data(), y, bytes, hipMemcpyDeviceToHost) == hipSuccess &&
               memcmp(hx.data(), hy.data(), bytes) == 0;
      };
      for (int q = 0; q < 4 && same; ++q) same = hsz[q] == dsz[q];
      same = same && h.max_lines == m.max_lines && h.max_lines_u == m.max_lines_u && h.u_padded == m.u_padded;
      same = same && eq(h.desc, m.desc, dsz[0]) && eq(h.rec, m.rec, dsz[1]) && eq(h.desc_u, m.desc_u, dsz[2]) && eq(h.rec_u, m.rec_u, dsz[3]);
      drop(h);
      if (!same) {
        fprintf(stderr, "[slq] SLQ_DEVICE_BUILD=2: the device-built stream of %d-merged tiles differs from the host-built one\n", R);
        drop(m);
        return false;
      }
    }
    return true;
  } catch (const std::bad_alloc &) {
    return false;
  }
}

// The interior upper-triangle stream of the fused update + alpha pass (slq_ring_fa.hpp): the operator's upper triangle (diagonal +
// doubled strict upper entries, as the alpha-only pass reads it) over the SAME tiles, minus every entry whose column lies in a
// later XCD chunk than its row - those rows of W_{j+1} are written by another XCD, whose L2 this one cannot see inside a launch -
// and those entries as an edge list. Rows padded to whole chunks of four entries (the branch-free consumer); given up when some
// tile's padded record would outgrow its slot. Built from what the operator keeps on the device, once.
static bool ensure_fa_stream(slq_operator *op) {
  if (!op->tile_desc || !op->tiles_ringed || !op->merged_lock || !op->rowptr_u || !op->tile_desc_u) return false;
  std::lock_guard<std::mutex> guard(*op->merged_lock);
  slq_operator::FaStream &f = op->fa;
  if (f.tried) return f.desc != nullptr;
  f.tried = true;
  const int64_t n = op->n;
  const size_t nu = (size_t)op->nnz_u, es = esize(op->dtype);
  const int32_t ntiles = op->tiles.xcd_tile[8];
  try {
    std::vector<int32_t> urp((size_t)n + 1), uci(nu), tr((size_t)ntiles + 1);
    std::vector<char> uva(nu * es);
    if (hipMemcpy(urp.data(), op->rowptr_u, ((size_t)n + 1) * 4, hipMemcpyDeviceToHost) != hipSuccess) return false;
    if (hipMemcpy(uci.data(), op->colind_u, nu * 4, hipMemcpyDeviceToHost) != hipSuccess) return false;
    if (hipMemcpy(uva.data(), op->vals_u, nu * es, hipMemcpyDeviceToHost) != hipSuccess) return false;
    if (hipMemcpy(tr.data(), op->tiles.tile_row, ((size_t)ntiles + 1) * 4, hipMemcpyDeviceToHost) != hipSuccess) return false;
    // chunk of every row: rows [tr[xcd_tile[x]], tr[xcd_tile[x + 1]]) belong to XCD x
    int32_t cend[8];
    for (int x = 0; x < 8; ++x) cend[x] = tr[(size_t)op->tiles.xcd_tile[x + 1]];
    std::vector<int32_t> irp((size_t)n + 1, 0), ici, er, ec;
    std::vector<char> iva, ev;
    ici.reserve(nu + kCsrPad);
    iva.reserve((nu + kCsrPad) * es);
    int x = 0;
    for (int64_t r = 0; r < n; ++r) {
      while (x < 7 && r >= cend[x]) ++x;
      for (int32_t q = urp[(size_t)r]; q < urp[(size_t)r + 1]; ++q) {
        const int32_t c = uci[(size_t)q];
        if (c >= cend[x]) {  // crosses into a later chunk
          er.push_back((int32_t)r);
          ec.push_back(c);
          ev.insert(ev.end(), uva.begin() + (size_t)q * es, uva.begin() + ((size_t)q + 1) * es);
        } else {
          ici.push_back(c);
          iva.insert(iva.end(), uva.begin() + (size_t)q * es, uva.begin() + ((size_t)q + 1) * es);
        }
      }
      irp[(size_t)r + 1] = (int32_t)ici.size();
    }
    ici.resize(ici.size() + kCsrPad, 0);
    iva.resize(iva.size() + (size_t)kCsrPad * es, 0);
    std::vector<int32_t> tp, tc, lc, si;
    RawBuf<int32_t> desc;
    RawBuf<char> rec;
    int mx = 0;
    build_tile_meta(n, irp.data(), ici.data(), tr, tp, tc, lc, si, &mx);
    if (mx > kRingTileCols) return false;
    bool pad = true;
    if (op->dtype == SLQ_F64) build_ring_stream<double>(1, irp.data(), (const double *)iva.data(), tr, tp, tc, lc, si, desc, rec, &pad);
    else build_ring_stream<float>(1, irp.data(), (const float *)iva.data(), tr, tp, tc, lc, si, desc, rec, &pad);
    if (!pad) return false;
    const size_t ne = er.size();
    bool ok = hipMalloc((void **)&f.desc, desc.size() * 4) == hipSuccess && hipMalloc((void **)&f.rec, rec.size()) == hipSuccess &&
              hipMalloc((void **)&f.er, std::max<size_t>(ne, 1) * 4) == hipSuccess && hipMalloc((void **)&f.ec, std::max<size_t>(ne, 1) * 4) == hipSuccess &&
              hipMalloc(&f.ev, std::max<size_t>(ne, 1) * es) == hipSuccess;
    ok = ok && hipMemcpy(f.desc, desc.data(), desc.size() * 4, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(f.rec, rec.data(), rec.size(), hipMemcpyHostToDevice) == hipSuccess;
    if (ok && ne) {
      ok = hipMemcpy(f.er, er.data(), ne * 4, hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(f.ec, ec.data(), ne * 4, hipMemcpyHostToDevice) == hipSuccess &&
           hipMemcpy(f.ev, ev.data(), ne * es, hipMemcpyHostToDevice) == hipSuccess;
    }
    f.nedges = (int)ne;
    if (!ok) {
      for (void **q : {(void **)&f.desc, (void **)&f.rec, (void **)&f.er, (void **)&f.ec, &f.ev}) {
        if (*q) hipFree(*q);
        *q = nullptr;
      }
    }
    if (env_int("SLQ_DEBUG", 0) != 0) fprintf(stderr, "[slq] fused alpha: interior upper stream built, %zu entries cross XCD chunks (of %zu)\n", ne, nu);
    return ok;
  } catch (const std::bad_alloc &) {
    return false;
  }
}

extern "C" int slq_operator_shape(const slq_operator *op, int64_t *nrows, int64_t *ncols,
                                  int64_t *nnz, int *dtype) {
  if (!op) return fail(SLQ_EINVAL, "op is NULL");
  if (nrows) *nrows = op->n;
  if (ncols) *ncols = op->n;
  if (nnz) *nnz = op->nnz;
  if (dtype) *dtype = op->dtype;
  return SLQ_OK;
}

// ---------------------------------------------------------------------------------------------------
// geometry + dispatch
// ---------------------------------------------------------------------------------------------------
static void choose_geometry(int dtype, int nprobes, int *LPR, int *PW, int *NP) {
  const int V = dtype == SLQ_F64 ? 2 : 4;
  int lpr = 8;
  while (lpr < 64 && lpr * V < nprobes) lpr *= 2;
  const int forced = env_int("SLQ_LPR", 0);
  if (forced == 8 || forced == 16 || forced == 32 || forced == 64) lpr = forced;
  *LPR = lpr;
  *PW = lpr * V;
  *NP = (nprobes + *PW - 1) / *PW;
}

template <int L> using LprTag = std::integral_constant<int, L>;
// calls fn(F{}, LprTag<L>{}) for the runtime (dtype, lanes-per-row) pair
template <typename Fn> static inline void dispatch(int dtype, int lpr, Fn &&fn) {
  if (dtype == SLQ_F64) {
    switch (lpr) {
      case 64: fn(double{}, LprTag<64>{}); break;
      case 32: fn(double{}, LprTag<32>{}); break;
      case 16: fn(double{}, LprTag<16>{}); break;
      default: fn(double{}, LprTag<8>{}); break;
    }
  } else {
    switch (lpr) {
      case 64: fn(float{}, LprTag<64>{}); break;
      case 32: fn(float{}, LprTag<32>{}); break;
      case 16: fn(float{}, LprTag<16>{}); break;
      default: fn(float{}, LprTag<8>{}); break;
    }
  }
}
#define DISPATCH(DT, LPRV, BODY)                          \
  dispatch(DT, LPRV, [&](auto _f, auto _l) {              \
    using F = decltype(_f);                               \
    constexpr int L = decltype(_l)::value;                \
    BODY;                                                 \
  })

static inline char *slot_ptr(const slq_plan *p, int slot) {
  return (char *)p->ring + (size_t)slot * (size_t)p->slot_stride * p->esz;
}

// profiling brackets ----------------------------------------------------------------------------------
static int prof_begin(slq_plan *p, int kind, ProfEvent *ev) {
  if (!p->prof) return SLQ_OK;
  if (!p->pool.empty()) {
    *ev = p->pool.back();
    p->pool.pop_back();
  } else {
    HIP_TRY(hipEventCreate(&ev->a));
    HIP_TRY(hipEventCreate(&ev->b));
  }
  ev->kind = kind;
  HIP_TRY(hipEventRecord(ev->a, p->ctx->stream));
  return SLQ_OK;
}
static int prof_end(slq_plan *p, ProfEvent *ev) {
  if (!p->prof) return SLQ_OK;
  HIP_TRY(hipEventRecord(ev->b, p->ctx->stream));
  p->events.push_back(*ev);
  return SLQ_OK;
}
#define PROFILED(plan, kind, launch)        \
  do {                                      \
    ProfEvent _ev;                          \
    SLQ_TRY(prof_begin(plan, kind, &_ev));  \
    launch;                                 \
    SLQ_TRY(prof_end(plan, &_ev));          \
  } while (0)

static int prof_collect(slq_plan *p) {
  if (p->events.empty()) return SLQ_OK;
  HIP_TRY(hipStreamSynchronize(p->ctx->stream));
  for (auto &ev : p->events) {
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, ev.a, ev.b));
    p->acc.ms[ev.kind] += ms;
    p->acc.launches[ev.kind] += 1;
    p->pool.push_back(ev);
  }
  p->events.clear();
  return SLQ_OK;
}

extern "C" int slq_plan_profile_enable(slq_plan *plan, int enable) {
  if (!plan) return fail(SLQ_EINVAL, "plan is NULL");
  SLQ_TRY(prof_collect(plan));
  plan->prof = enable != 0;
  return SLQ_OK;
}

extern "C" int slq_plan_profile_read(slq_plan *plan, slq_profile *out, int reset) {
  if (!plan || !out) return fail(SLQ_EINVAL, "plan/out is NULL");
  HIP_TRY(hipSetDevice(plan->ctx->device));
  SLQ_TRY(prof_collect(plan));
  *out = plan->acc;
  if (reset) memset(&plan->acc, 0, sizeof(plan->acc));
  return SLQ_OK;
}

// ---------------------------------------------------------------------------------------------------
// plan
// ---------------------------------------------------------------------------------------------------
static int normalise_params(int64_t n, int *deg, int *orth) {
  if (*deg < 1) return fail(SLQ_EINVAL, "Number of steps must be positive!");
  if (*deg > n) *deg = (int)n;                                // lanczos.py:79, operators.py:68
  if (*deg > kMaxDeg) return fail(SLQ_EINVAL, "deg %d exceeds the supported maximum %d", *deg, kMaxDeg);
  if (*orth < 0 || *orth > *deg) *orth = *deg;                 // lanczos.py:88, operators.py:80
  return SLQ_OK;
}

static int ring_slots(int deg, int orth, int keep_basis) {
  if (keep_basis) return deg + 1;
  if (orth == 0) return 2;
  return std::max(orth + 1, 3);
}

static void grid_sizes(int n, int LPR, int NP, int num_cus, int *nblkA, int *nblkS, int *nblkU, bool pipelined) {
  const int RPW = 64 / LPR;
  const int rows_per_block = kWaves * RPW;
  // Tunables: resident workgroups (kBlock threads) per CU, summed over the panels of a launch.
  // Defaults from the MI355X sweeps: in-place read-modify-write sweeps peak at
  // ~2 workgroups per CU (more concurrent writers lose 5-10 %); the SpMM likes 4-8.
  const int per_cu_a = std::max(1, env_int("SLQ_BLOCKS_PER_CU_SPMM", env_int("SLQ_BLOCKS_PER_CU", 4)));
  const int per_cu_s = std::max(1, env_int("SLQ_BLOCKS_PER_CU_STREAM", env_int("SLQ_BLOCKS_PER_CU", 2)));
  // sweep A: a multiple of 8 blocks (XCD-aware chunking), no more than the rows can feed
  const int chunk = (n + 7) / 8;
  int per_xcd = std::min(std::max(8, num_cus * per_cu_a / NP) / 8, (chunk + rows_per_block - 1) / rows_per_block);
  per_xcd = std::max(per_xcd, 1);
  *nblkA = 8 * per_xcd;
  // fused dots/update passes: 2 workgroups resident per CU (LDS padding, enqueue_run) and a grid of 2 per CU
  // per panel. More rows in flight evict each other's gather halo
  // (dots pass, r = 3, per 30 launches: 36.3 ms at 2 resident, 41.3 at 3), and a grid that is not a multiple
  // of what is resident leaves a ragged last round. Panels run one after the other (panel-major dispatch).
  const int per_cu_u = std::max(1, env_int("SLQ_BLOCKS_PER_CU_FUSED", pipelined ? 1 : 2));  // per panel; 1 with the pipelined row loop
  int per_xcd_u = std::min(std::max(8, num_cus * per_cu_u) / 8, (chunk + rows_per_block - 1) / rows_per_block);
  *nblkU = 8 * std::max(per_xcd_u, 1);
  int s = std::min(std::max(1, num_cus * per_cu_s / NP), (n + rows_per_block - 1) / rows_per_block);
  *nblkS = std::max(s, 1);
}

extern "C" int slq_plan_query_bytes(int dtype, int64_t n, int nprobes, int deg, int orth,
                                    int keep_basis, size_t *bytes) {
  if (!bytes) return fail(SLQ_EINVAL, "bytes is NULL");
  SLQ_TRY(check_dtype(dtype));
  if (n <= 0 || nprobes <= 0) return fail(SLQ_EINVAL, "n and nprobes must be positive");
  SLQ_TRY(normalise_params(n, &deg, &orth));
  int LPR, PW, NP;
  choose_geometry(dtype, nprobes, &LPR, &PW, &NP);
  const size_t S = ring_slots(deg, orth, keep_basis);
  *bytes = S * (size_t)NP * (size_t)n * PW * esize(dtype);
  return SLQ_OK;
}

// What a plan on `op` allocates: the ring (slq_plan_query_bytes) plus the product panels of operators that are not applied
// row by row inside the passes - T (dense / callback / Gram: one panel; the fp64 dense MFMA kernel with big tiles adds up to
// 16 split-K slabs) and T2 (Gram: the m-row intermediate). The one-shot entries size their probe chunks from this.
static int plan_bytes_on(const slq_operator *op, int nprobes, int deg, int orth, int keep_basis, size_t *bytes) {
  SLQ_TRY(slq_plan_query_bytes(op->dtype, op->n, nprobes, deg, orth, keep_basis, bytes));
  int LPR, PW, NP;
  choose_geometry(op->dtype, nprobes, &LPR, &PW, &NP);
  const size_t panel = (size_t)NP * PW * esize(op->dtype);
  if (op->kind != OP_CSR) {
    const bool big_tiles = op->kind == OP_DENSE && (op->dtype == SLQ_F32 || PW >= 32);  // K-split slabs of the MFMA dense kernels (<= 16)
    *bytes += (size_t)(1 + (big_tiles ? 16 : 0)) * panel * (size_t)op->n;
  }
  if (op->kind == OP_GRAM) *bytes += panel * (size_t)op->mrows;
  return SLQ_OK;
}

extern "C" int slq_plan_destroy(slq_plan *p) {
  if (!p) return SLQ_OK;
  hipSetDevice(p->ctx->device);
  hipStreamSynchronize(p->ctx->stream);
  for (auto &ev : p->events) { hipEventDestroy(ev.a); hipEventDestroy(ev.b); }
  for (auto &ev : p->pool) { hipEventDestroy(ev.a); hipEventDestroy(ev.b); }
  if (p->graph_exec) hipGraphExecDestroy(p->graph_exec);
  if (p->ring) hipFree(p->ring);
  if (p->ring32) hipFree(p->ring32);
  if (p->T) hipFree(p->T);
  if (p->T2) hipFree(p->T2);
  if (p->stage) hipFree(p->stage);
  if (p->scal) hipFree(p->scal);
  if (p->part) hipFree(p->part);
  if (p->fa_cnt) hipFree(p->fa_cnt);
  if (p->sweep_cols_d) hipFree(p->sweep_cols_d);
  if (p->quad_d) hipFree(p->quad_d);
  if (p->st.active) hipFree(p->st.active);
  ctx_release(p->ctx);
  delete p;
  return SLQ_OK;
}

static int set_kernel_attributes(slq_plan *p);

extern "C" int slq_plan_create(slq_context *ctx, slq_operator *op, int nprobes, int deg, int orth,
                               int keep_basis, slq_plan **out) {
  if (!ctx || !op || !out) return fail(SLQ_EINVAL, "ctx/op/out is NULL");
  *out = nullptr;
  if (op->ctx != ctx) return fail(SLQ_EINVAL, "operator belongs to another context");
  if (nprobes <= 0) return fail(SLQ_EINVAL, "nprobes must be positive");
  SLQ_TRY(normalise_params(op->n, &deg, &orth));
  HIP_TRY(hipSetDevice(ctx->device));
  if (ctx->dead) return fail(SLQ_EINVAL, "the context has been destroyed");
  slq_plan *p = new (std::nothrow) slq_plan();
  if (!p) return fail(SLQ_ENOMEM, "host allocation failed");
  p->ctx = ctx;
  ctx_retain(ctx);
  p->op = op;
  p->dtype = op->dtype;
  p->n = (int)op->n;
  p->nprobes = nprobes;
  p->deg = deg;
  p->orth = orth;
  p->keep_basis = keep_basis != 0;
  p->esz = esize(op->dtype);
  p->sw = Switches{env_int("SLQ_FUSED", 1), env_int("SLQ_NT", 1) != 0, env_int("SLQ_GRAPH", 1) != 0, env_int("SLQ_MGS", 0) != 0,
                   env_int("SLQ_STORED_U", 1) != 0, env_int("SLQ_MERGED", 1) != 0, env_int("SLQ_CROSS", 1) != 0,
                   tiles_mode() != 0, env_int("SLQ_RING_ALPHA", 2), env_int("SLQ_RING_REV", 1) != 0, env_int("SLQ_DENSE_MFMA", 1) != 0, env_int("SLQ_DENSE_TILE16", 0) != 0, env_int("SLQ_DENSE_LDS", 1) != 0, env_int("SLQ_PIPE", -1), env_int("SLQ_RING32", 0) != 0,
                   env_int("SLQ_FUSED_LDS_PAD", -1), env_int("SLQ_SPMM_LDS_PAD", 57344), env_int("SLQ_FUSED_ALPHA", kFusedAlphaDefault) != 0,
                   env_int("SLQ_DEFER_AXPY", 1) != 0};
  choose_geometry(op->dtype, nprobes, &p->LPR, &p->PW, &p->NP);
  p->bpad = p->NP * p->PW;
  p->S = ring_slots(deg, orth, p->keep_basis);
  // Opt-in fp32 archive (SLQ_RING32=1; fp64 plans with reorthogonalisation deeper than the fused passes reach, basis
  // not kept): the fp64 ring shrinks to the three live vectors and every finished vector is also stored as fp32;
  // reorthogonalisation columns i >= 2 are read from the archive and accumulated in fp64. Changes results at the
  // 1e-8 level of the Lanczos coefficients, hence never the default.
  p->ring32 = nullptr;
  p->S32 = 0;
  p->slot_stride = (int64_t)p->NP * p->n * p->PW;
  p->ring32_on = p->sw.ring32 && op->dtype == SLQ_F64 && !p->keep_basis && orth > kFusedMaxR;
  if (p->ring32_on) {
    p->S = 3;
    p->S32 = orth + 1;
  }
  p->rmax = std::max(p->keep_basis ? deg : orth, 1);
  // Row loop of the dots/update passes (slq_kernels.hpp: k_csr_pass). Measured on configs[1] and on the 100^3 grid
  // (DESIGN.md §5.3): rows of up to 5 nonzeros are fastest with the plain loop at 2 resident workgroups per CU (82.9
  // against 91.7 ms per step), 7-point rows with the pipelined loop at ONE resident workgroup per CU (93.7 against
  // 101.0 ms): what bounds both is the traffic a CU's vector-memory pipe has in flight, and the pipelined loop puts
  // the same bytes in flight with half the waves.
  p->pipelined = op->kind == OP_CSR && p->LPR == 64 && (p->sw.pipe >= 0 ? p->sw.pipe != 0 : (double)op->nnz / (double)std::max<int64_t>(op->n, 1) > 5.5);
  grid_sizes(p->n, p->LPR, p->NP, ctx->num_cus, &p->nblkA, &p->nblkS, &p->nblkU, p->pipelined);
  {
    // Fused alpha pass: one read sweep plus the gathers. Two regimes, told apart by the
    // gathers per row of the matrix the pass walks (upper triangle when the operator is symmetric):
    //  * <= 3 (5-point stencil, upper triangle): each wave has too few loads in flight, the pass is bound by
    //    the rowptr -> colind -> gather latency chain: 4 workgroups per CU in total, panels side by side
    //    (configs[1]: 0.88 ms at 2 per CU, 0.51 at 4; one panel at a time with 4 resident 0.61);
    //  * more (full 5-point rows: 0.92 ms at 2 per CU vs 0.96 at 4; 7-point upper triangle: 0.67 vs 0.83;
    //    random graphs): the waves carry enough loads and extra rows in flight only evict each other's halo
    //    from the XCD's L2: 2 resident per CU (LDS padding), 2 per CU *per panel*, panel after panel.
    const double gathers = op->kind != OP_CSR ? 0.0 : (double)(op->rowptr_u ? op->nnz_u : op->nnz) / (double)std::max<int64_t>(op->n, 1);
    const int local = op->kind == OP_CSR && gathers <= 3.2;
    const int per_cu_env = env_int("SLQ_BLOCKS_PER_CU_ALPHA", 0);  // total over the panels
    const int rows_per_block = kWaves * (64 / p->LPR);
    const int chunk = (p->n + 7) / 8;
    const int per_panel = per_cu_env > 0 ? std::max(8, ctx->num_cus * per_cu_env / p->NP)
                                         : (local ? std::max(8, ctx->num_cus * 4 / p->NP) : ctx->num_cus * 2);
    const int per_xcd = std::min(per_panel / 8, (chunk + rows_per_block - 1) / rows_per_block);
    p->nblkF = 8 * std::max(per_xcd, 1);
    p->alpha_pad = (size_t)env_int("SLQ_ALPHA_LDS_PAD", (per_cu_env > 0 || local) ? 0 : 65536);
  }
  {
    // which tile stream, if any (plan_tiled): wide panels take the tiles as clustered; panels of 32 / 16 lanes per row the
    // merged tiles of the narrow-panel ring kernel (built on first use; nontemporal streams only - the one form instantiated)
    p->ringR = 0;
    p->ring_gen = p->ring_deep = p->gram = false;
    p->rs_desc = p->rs_desc_u = nullptr;
    p->rs_rec = p->rs_rec_u = nullptr;
    p->rs_u_padded = false;
    p->ring_staged = false;
    for (int x = 0; x < 9; ++x) p->rs_xcd[x] = op->tiles.xcd_tile[x], p->rs_xcd_u[x] = op->xcd_tile_u[x];
    if (op->kind == OP_CSR && op->tiles.tile_ptr && p->sw.tiles) {
      if (p->LPR == 64) {
        p->ringR = 1;
        if (op->tiles_ringed) {
          p->rs_desc = op->tile_desc, p->rs_rec = op->tile_rec;
          if (op->tile_desc_u && op->upper_per_row <= (double)env_int("SLQ_RING_ALPHA_MAX_X100", (int)(100 * kTileAlphaColsPerRow)) / 100.0)
            p->rs_desc_u = op->tile_desc_u, p->rs_rec_u = op->tile_rec_u, p->rs_u_padded = op->tile_u_padded;
          p->ring_gen = p->sw.nt && env_int("SLQ_RING_GEN", 1) != 0;
          p->ring_deep = p->sw.nt && env_int("SLQ_RING_DEEP", 1) != 0;
        }
      } else if ((p->LPR == 32 || p->LPR == 16) && op->tiles_ringed && p->sw.nt && env_int("SLQ_RING_NARROW", 1) != 0 &&
                 ensure_ring_stream(op, 64 / p->LPR)) {
        const slq_operator::MergedStream &m = op->merged[p->LPR == 32 ? 0 : 1];
        p->ringR = 64 / p->LPR;
        p->ring_gen = true;
        p->ring_deep = env_int("SLQ_RING_DEEP", 1) != 0;
        p->rs_desc = m.desc, p->rs_rec = m.rec, p->rs_desc_u = m.desc_u, p->rs_rec_u = m.rec_u, p->rs_u_padded = m.u_padded;
        for (int x = 0; x < 9; ++x) p->rs_xcd[x] = m.xcd_tile[x], p->rs_xcd_u[x] = m.xcd_tile[x];  // (merged streams: one partition for both)
      }
      // alpha-only pass: LDS-DMA loaders everywhere since their r03 rewrite (merged tiles: a lane reads its lines' sources straight
      // out of the staged descriptor - 100^3, 64 probes 0.230 -> 0.187 ms against the register-staged loaders that had been the
      // faster form there, configs[1] 0.129 -> 0.113). SLQ_RING_STAGED=1 takes the loaders through registers (GEO 1) again.
      p->ring_staged = env_int("SLQ_RING_STAGED", 0) != 0;
      // the Gram sequence needs every step of the window on k_ring_pass (PASS_UPDATEG), i.e. the deep form too
      // (and an EXACTLY symmetric operator: the sequence rewrites W_t . (A W_j) as (A W_t) . W_j - §4.6 - while the reference's recurrence never
      // looks at symmetry, lanczos.h:127-136; rowptr_u is the record of that check. r04: r03 took the sequence on any tiled operator)
      p->gram = p->ring_gen && p->ring_deep && p->sw.merged && !p->sw.mgs && op->rowptr_u != nullptr && env_int("SLQ_GRAM", 1) != 0;
      // fused alpha: whole-row panels on the Gram sequence with at most kRingMaxR ring columns per step, descending update sweeps,
      // the alpha-only pass on the padded upper-triangle stream (step 0 still runs it), and the interior stream buildable
      p->fa_on = p->sw.fused_alpha && !g_fa_broken.load() && p->gram && p->ringR == 1 && p->sw.ring_rev && p->rs_desc_u != nullptr && p->rs_u_padded &&
                 p->sw.ring_alpha == 2 && orth >= 1 && orth <= kRingMaxR && ensure_fa_stream(op);
    }
  }
  // the Gram sequence on the generic passes (no tiles, or a plan whose panels the tiles do not serve): the dots pass - a third to a half of every
  // step's bytes - is gone there as well. Not for operators whose gathers are the cost (random graphs keep the stored-u sequence: it gathers once,
  // the Gram sequence twice) - enqueue_run decides that per step exactly as before.
  p->last_nostore = env_int("SLQ_LAST_STORE", 0) == 0;
  p->sweep_skip = env_int("SLQ_SWEEP_SKIP", 1) != 0;
  p->gram_csr = op->kind == OP_CSR && p->ringR == 0 && op->rowptr_u != nullptr && p->sw.merged && !p->sw.mgs && p->sw.nt && env_int("SLQ_GRAM", 1) != 0 && env_int("SLQ_GRAM_CSR", 1) != 0;
  {
    // tiled passes: as many workgroups resident per CU as their LDS images admit (2 x 72 KiB by default), the same number
    // per CU and panel in the grid, panel after panel
    const int img_kib = op->tiles.tile_ptr ? (op->tiles.max_cols + 16) * (SLQ_TILE_DB ? 2 : 1) : 160;
    const int per_cu_t = op->tiles_ringed ? 1 : std::max(1, env_int("SLQ_BLOCKS_PER_CU_TILED", std::max(1, std::min(4, 160 / std::max(img_kib, 1)))));
    int per_xcd_t = std::max(1, ctx->num_cus * per_cu_t / 8);
    per_xcd_t = std::max(1, std::min(per_xcd_t, env_int("SLQ_TILED_WGS_PER_XCD", per_xcd_t)));  // (experiments: fewer CUs sweeping a chunk)
    if (op->tiles.tile_ptr) {
      int mn = 1 << 30;
      for (int x = 0; x < 8; ++x) mn = std::min(mn, std::max(1, p->rs_xcd[x + 1] - p->rs_xcd[x]));
      per_xcd_t = std::min(per_xcd_t, mn);
    }
    p->nblkT = 8 * per_xcd_t;
  }
  memset(&p->acc, 0, sizeof(p->acc));
  memset(&p->st, 0, sizeof(p->st));

  const size_t ring_bytes = (size_t)p->S * (size_t)p->slot_stride * p->esz;
  const size_t bp = p->bpad;
  // alpha[deg+1], nu[orth margin for stale vectors t < 0 | deg+1], vnorm2, coefA[2], coefB, cross, gram[2][kFusedMaxR+1], gamma[rmax]
  const size_t nscal = ((size_t)(deg + 1) * 2 + (size_t)orth + 1 + 2 + 1 + 1 + 2 * (kFusedMaxR + 1) + (size_t)p->rmax) * bp;
  p->part_maxblk = std::max(std::max(std::max(std::max(p->nblkA, p->nblkF), p->nblkU), p->nblkS), p->nblkT);
  const size_t npart = (size_t)kReorthChunk * p->part_maxblk * bp;
  if (p->fa_on) {
    int mxt = 1;
    for (int x = 0; x < 8; ++x) mxt = std::max(mxt, p->rs_xcd[x + 1] - p->rs_xcd[x]);
    p->fa_rounds = (mxt + p->nblkT / 8 - 1) / (p->nblkT / 8) + 2;
  }
  hipError_t e = hipMalloc(&p->ring, ring_bytes);
  const size_t ring32_bytes = p->ring32_on ? (size_t)p->S32 * (size_t)p->slot_stride * sizeof(float) : 0;
  if (e == hipSuccess && p->ring32_on) e = hipMalloc((void **)&p->ring32, ring32_bytes);
  if (e == hipSuccess) e = hipMalloc((void **)&p->scal, nscal * 8);
  if (e == hipSuccess) e = hipMalloc((void **)&p->part, npart * 8);
  if (e == hipSuccess) e = hipMalloc((void **)&p->sweep_cols_d, 2 * sizeof(unsigned long long));
  if (e == hipSuccess) e = hipMemset(p->sweep_cols_d, 0, 2 * sizeof(unsigned long long));
  if (e == hipSuccess && p->fa_on) e = hipMalloc((void **)&p->fa_cnt, ((size_t)p->NP * 8 * p->fa_rounds + 8) * sizeof(int));
  if (e == hipSuccess) e = hipMalloc((void **)&p->st.active, bp * 2 * sizeof(int) + 16);
  if (e == hipSuccess) e = hipMalloc((void **)&p->quad_d, (bp + 2 * bp * (size_t)deg) * 8);
  // dense fp64 operator on the matrix cores with 32-row tiles: n/32 workgroups per panel rarely fill 256 CUs, so K is
  // also split over dense_ks workgroups whose raw products land in dense_ks slabs behind T. ks minimises the number
  // of workgroup rounds times the work per workgroup, plus a small cost per slab.
  p->dense_ks = 0;  // 0: the 16-row kernel with its fused epilogue
  const bool dense32 = op->kind == OP_DENSE && p->dtype == SLQ_F32 && p->sw.dense_mfma;  // fp32: k_dense_mfma32_lds, 256-row tiles, every panel width
  if (dense32 || (op->kind == OP_DENSE && p->dtype == SLQ_F64 && p->sw.dense_mfma && p->PW >= 32 && !p->sw.dense_tile16)) {
    const int rw = dense32 ? kDense32BM : 32 * (kWaves / (p->PW >= 64 ? 2 : 1));
    const double wgs = (double)((p->n + rw - 1) / rw) * p->NP;
    double best = 1e30;
    const int forced = env_int("SLQ_DENSE_KSPLIT", 0);
    for (int ks = 1; ks <= 16; ++ks) {
      const double cost = std::ceil(wgs * ks / ctx->num_cus) / ks + 0.005 * ks;
      if ((forced > 0 && ks == forced) || (forced <= 0 && cost < best - 1e-12)) { best = cost; p->dense_ks = ks; }
    }
  }
  const size_t t_slabs = op->kind == OP_CSR ? 0 : (size_t)(1 + p->dense_ks);
  if (e == hipSuccess && t_slabs) e = hipMalloc(&p->T, t_slabs * (size_t)p->slot_stride * p->esz);
  const size_t t2_bytes = op->kind == OP_GRAM ? (size_t)p->NP * (size_t)op->mrows * p->PW * p->esz : 0;
  if (e == hipSuccess && t2_bytes) e = hipMalloc(&p->T2, t2_bytes);
  if (e != hipSuccess) {
    slq_plan_destroy(p);
    return fail(e == hipErrorOutOfMemory ? SLQ_ENOMEM : SLQ_EHIP,
                "plan workspace (%zu bytes of Lanczos panels): %s", ring_bytes, hipGetErrorString(e));
  }
  p->bytes = ring_bytes + ring32_bytes + nscal * 8 + npart * 8 + (bp + 2 * bp * deg) * 8 + t_slabs * (size_t)p->slot_stride * p->esz + t2_bytes;
  double *s = p->scal;
  p->st.alpha = s; s += (size_t)(deg + 1) * bp;
  s += (size_t)orth * bp;  // nu rows for t = -orth .. -1 (zero unless the drop-in entry preloads stale columns)
  p->st.nu = s; s += (size_t)(deg + 1) * bp;
  p->st.vnorm2 = s; s += bp;
  p->st.coefA = s; s += 2 * bp;
  p->st.coefB = s; s += bp;
  p->st.cross = s; s += bp;
  p->st.gram = s; s += (size_t)2 * (kFusedMaxR + 1) * bp;
  p->st.gamma = s;
  p->st.steps = p->st.active + bp;
  p->fail_d = p->st.steps + bp;
  p->ring_fail_d = p->fail_d + 1;  // raised by k_csr_ring_pass when a bounded spin ran out (never cleared: the plan is dead)
  if (hipMemset(p->ring_fail_d, 0, sizeof(int)) != hipSuccess) {
    slq_plan_destroy(p);
    return fail(SLQ_EHIP, "hipMemset failed");
  }
  p->st.bpad = p->bpad;
  p->st.nprobes = nprobes;
  p->st.deg = deg;
  p->nodes_d = p->quad_d + bp;
  p->weights_d = p->nodes_d + bp * (size_t)deg;
  p->graph_exec = nullptr;
  p->graph_rtol = 0.0;
  p->graph_variant = 0;
  p->nstale = 0;
  {
    hipError_t ze = hipMemsetAsync(p->scal, 0, nscal * 8, ctx->stream);
    if (ze != hipSuccess) {
      slq_plan_destroy(p);
      return fail(SLQ_EHIP, "workspace clear: %s", hipGetErrorString(ze));
    }
  }
  {
    const int rc = set_kernel_attributes(p);
    if (rc != SLQ_OK) {
      slq_plan_destroy(p);
      return rc;
    }
  }
  *out = p;
  return SLQ_OK;
}

// Kernels that may be launched with more than the default 64 KiB of dynamic LDS (gamma staging):
// raise their limit once, outside any stream capture.
template <typename F, int L> static hipError_t raise_lds_limits() {
  hipError_t e = hipFuncSetAttribute((const void *)k_reorth_update<F, L>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if constexpr (std::is_same<F, double>::value)
    if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_reorth_update32<L>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  // fused passes: LDS padding caps their residency (SLQ_ALPHA_LDS_PAD experiments; dots/update: 2 per CU)
  std::vector<const void *> fused_fns = {
      (const void *)k_csr_pass<F, L, PASS_ALPHA, 0, 0, 0>,
      (const void *)k_csr_pass<F, L, PASS_ALPHA, 1, 0, 0>,
      (const void *)k_csr_pass<F, L, PASS_DOTS, 0, 1, 0>,
      (const void *)k_csr_pass<F, L, PASS_DOTS, 1, 1, 0>,
      (const void *)k_csr_pass<F, L, PASS_DOTS, 0, 2, 0>,
      (const void *)k_csr_pass<F, L, PASS_DOTS, 1, 2, 0>,
      (const void *)k_csr_pass<F, L, PASS_DOTS, 0, 3, 0>,
      (const void *)k_csr_pass<F, L, PASS_DOTS, 1, 3, 0>,
      (const void *)k_csr_pass<F, L, PASS_DOTS, 0, 4, 0>,
      (const void *)k_csr_pass<F, L, PASS_DOTS, 1, 4, 0>,
      (const void *)k_csr_pass<F, L, PASS_DOTS, 0, 5, 0>,
      (const void *)k_csr_pass<F, L, PASS_DOTS, 1, 5, 0>,
      (const void *)k_csr_pass<F, L, PASS_DOTS, 0, 6, 0>,
      (const void *)k_csr_pass<F, L, PASS_DOTS, 1, 6, 0>,
      (const void *)k_csr_pass<F, L, PASS_DOTS, 0, 7, 0>,
      (const void *)k_csr_pass<F, L, PASS_DOTS, 1, 7, 0>,
      (const void *)k_csr_pass<F, L, PASS_DOTS, 0, 8, 0>,
      (const void *)k_csr_pass<F, L, PASS_DOTS, 1, 8, 0>,
      (const void *)k_csr_pass<F, L, PASS_ADOTS, 0, 1, 0>,
      (const void *)k_csr_pass<F, L, PASS_ADOTS, 1, 1, 0>,
      (const void *)k_csr_pass<F, L, PASS_ADOTS, 0, 2, 0>,
      (const void *)k_csr_pass<F, L, PASS_ADOTS, 1, 2, 0>,
      (const void *)k_csr_pass<F, L, PASS_ADOTS, 0, 3, 0>,
      (const void *)k_csr_pass<F, L, PASS_ADOTS, 1, 3, 0>,
      (const void *)k_csr_pass<F, L, PASS_ADOTS, 0, 4, 0>,
      (const void *)k_csr_pass<F, L, PASS_ADOTS, 1, 4, 0>,
      (const void *)k_csr_pass<F, L, PASS_ADOTS, 0, 5, 0>,
      (const void *)k_csr_pass<F, L, PASS_ADOTS, 1, 5, 0>,
      (const void *)k_csr_pass<F, L, PASS_ADOTS, 0, 6, 0>,
      (const void *)k_csr_pass<F, L, PASS_ADOTS, 1, 6, 0>,
      (const void *)k_csr_pass<F, L, PASS_ADOTS, 0, 7, 0>,
      (const void *)k_csr_pass<F, L, PASS_ADOTS, 1, 7, 0>,
      (const void *)k_csr_pass<F, L, PASS_ADOTS, 0, 8, 0>,
      (const void *)k_csr_pass<F, L, PASS_ADOTS, 1, 8, 0>,
      (const void *)k_csr_pass<F, L, PASS_UPDATE, 0, 0, 0>,
      (const void *)k_csr_pass<F, L, PASS_UPDATE, 1, 0, 0>,
      (const void *)k_csr_pass<F, L, PASS_UPDATE, 0, 1, 0>,
      (const void *)k_csr_pass<F, L, PASS_UPDATE, 1, 1, 0>,
      (const void *)k_csr_pass<F, L, PASS_UPDATE, 0, 2, 0>,
      (const void *)k_csr_pass<F, L, PASS_UPDATE, 1, 2, 0>,
      (const void *)k_csr_pass<F, L, PASS_UPDATE, 0, 3, 0>,
      (const void *)k_csr_pass<F, L, PASS_UPDATE, 1, 3, 0>,
      (const void *)k_csr_pass<F, L, PASS_UPDATE, 0, 4, 0>,
      (const void *)k_csr_pass<F, L, PASS_UPDATE, 1, 4, 0>,
      (const void *)k_csr_pass<F, L, PASS_UPDATE, 0, 5, 0>,
      (const void *)k_csr_pass<F, L, PASS_UPDATE, 1, 5, 0>,
      (const void *)k_csr_pass<F, L, PASS_UPDATE, 0, 6, 0>,
      (const void *)k_csr_pass<F, L, PASS_UPDATE, 1, 6, 0>,
      (const void *)k_csr_pass<F, L, PASS_UPDATE, 0, 7, 0>,
      (const void *)k_csr_pass<F, L, PASS_UPDATE, 1, 7, 0>,
      (const void *)k_csr_pass<F, L, PASS_UPDATE, 0, 8, 0>,
      (const void *)k_csr_pass<F, L, PASS_UPDATE, 1, 8, 0>,
      (const void *)k_csr_pass<F, L, PASS_UPDATEG, 1, 1, 0>,
      (const void *)k_csr_pass<F, L, PASS_UPDATEG, 1, 2, 0>,
      (const void *)k_csr_pass<F, L, PASS_UPDATEG, 1, 3, 0>,
      (const void *)k_csr_pass<F, L, PASS_UPDATEG, 1, 4, 0>,
      (const void *)k_csr_pass<F, L, PASS_UPDATEG, 1, 5, 0>,
      (const void *)k_csr_pass<F, L, PASS_UPDATEG, 1, 6, 0>,
      (const void *)k_csr_pass<F, L, PASS_UPDATEG, 1, 7, 0>,
      (const void *)k_csr_pass<F, L, PASS_UPDATEG, 1, 8, 0>};
  if constexpr (L == 64) {
    fused_fns.push_back((const void *)k_csr_pass<F, L, PASS_UPDATEG, 1, 1, 1>);
    fused_fns.push_back((const void *)k_csr_pass<F, L, PASS_UPDATEG, 1, 2, 1>);
    fused_fns.push_back((const void *)k_csr_pass<F, L, PASS_UPDATEG, 1, 3, 1>);
    fused_fns.push_back((const void *)k_csr_pass<F, L, PASS_UPDATEG, 1, 4, 1>);
    fused_fns.push_back((const void *)k_csr_pass<F, L, PASS_UPDATEG, 1, 5, 1>);
    fused_fns.push_back((const void *)k_csr_pass<F, L, PASS_UPDATEG, 1, 6, 1>);
    fused_fns.push_back((const void *)k_csr_pass<F, L, PASS_UPDATEG, 1, 7, 1>);
    fused_fns.push_back((const void *)k_csr_pass<F, L, PASS_UPDATEG, 1, 8, 1>);
    fused_fns.push_back((const void *)k_csr_pass<F, L, PASS_DOTS, 0, 1, 1>);
    fused_fns.push_back((const void *)k_csr_pass<F, L, PASS_DOTS, 1, 1, 1>);
    fused_fns.push_back((const void *)k_csr_pass<F, L, PASS_DOTS, 0, 2, 1>);
    fused_fns.push_back((const void *)k_csr_pass<F, L, PASS_DOTS, 1, 2, 1>);
    fused_fns.push_back((const void *)k_csr_pass<F, L, PASS_DOTS, 0, 3, 1>);
    fused_fns.push_back((const void *)k_csr_pass<F, L, PASS_DOTS, 1, 3, 1>);
    fused_fns.push_back((const void *)k_csr_pass<F, L, PASS_DOTS, 0, 4, 1>);
    fused_fns.push_back((const void *)k_csr_pass<F, L, PASS_DOTS, 1, 4, 1>);
    fused_fns.push_back((const void *)k_csr_pass<F, L, PASS_DOTS, 0, 5, 1>);
    fused_fns.push_back((const void *)k_csr_pass<F, L, PASS_DOTS, 1, 5, 1>);
    fused_fns.push_back((const void *)k_csr_pass<F, L, PASS_DOTS, 0, 6, 1>);
    fused_fns.push_back((const void *)k_csr_pass<F, L, PASS_DOTS, 1, 6, 1>);
    fused_fns.push_back((const void *)k_csr_pass<F, L, PASS_DOTS, 0, 7, 1>);
    fused_fns.push_back((const void *)k_csr_pass<F, L, PASS_DOTS, 1, 7, 1>);
    fused_fns.push_back((const void *)k_csr_pass<F, L, PASS_DOTS, 0, 8, 1>);
    fused_fns.push_back((const void *)k_csr_pass<F, L, PASS_DOTS, 1, 8, 1>);
    fused_fns.push_back((const void *)k_csr_pass<F, L, PASS_ADOTS, 0, 1, 1>);
    fused_fns.push_back((const void *)k_csr_pass<F, L, PASS_ADOTS, 1, 1, 1>);
    fused_fns.push_back((const void *)k_csr_pass<F, L, PASS_ADOTS, 0, 2, 1>);
    fused_fns.push_back((const void *)k_csr_pass<F, L, PASS_ADOTS, 1, 2, 1>);
    fused_fns.push_back((const void *)k_csr_pass<F, L, PASS_ADOTS, 0, 3, 1>);
    fused_fns.push_back((const void *)k_csr_pass<F, L, PASS_ADOTS, 1, 3, 1>);
    fused_fns.push_back((const void *)k_csr_pass<F, L, PASS_ADOTS, 0, 4, 1>);
    fused_fns.push_back((const void *)k_csr_pass<F, L, PASS_ADOTS, 1, 4, 1>);
    fused_fns.push_back((const void *)k_csr_pass<F, L, PASS_ADOTS, 0, 5, 1>);
    fused_fns.push_back((const void *)k_csr_pass<F, L, PASS_ADOTS, 1, 5, 1>);
    fused_fns.push_back((const void *)k_csr_pass<F, L, PASS_ADOTS, 0, 6, 1>);
    fused_fns.push_back((const void *)k_csr_pass<F, L, PASS_ADOTS, 1, 6, 1>);
    fused_fns.push_back((const void *)k_csr_pass<F, L, PASS_ADOTS, 0, 7, 1>);
    fused_fns.push_back((const void *)k_csr_pass<F, L, PASS_ADOTS, 1, 7, 1>);
    fused_fns.push_back((const void *)k_csr_pass<F, L, PASS_ADOTS, 0, 8, 1>);
    fused_fns.push_back((const void *)k_csr_pass<F, L, PASS_ADOTS, 1, 8, 1>);
    fused_fns.push_back((const void *)k_csr_pass<F, L, PASS_UPDATE, 0, 0, 1>);
    fused_fns.push_back((const void *)k_csr_pass<F, L, PASS_UPDATE, 1, 0, 1>);
    fused_fns.push_back((const void *)k_csr_pass<F, L, PASS_UPDATE, 0, 1, 1>);
    fused_fns.push_back((const void *)k_csr_pass<F, L, PASS_UPDATE, 1, 1, 1>);
    fused_fns.push_back((const void *)k_csr_pass<F, L, PASS_UPDATE, 0, 2, 1>);
    fused_fns.push_back((const void *)k_csr_pass<F, L, PASS_UPDATE, 1, 2, 1>);
    fused_fns.push_back((const void *)k_csr_pass<F, L, PASS_UPDATE, 0, 3, 1>);
    fused_fns.push_back((const void *)k_csr_pass<F, L, PASS_UPDATE, 1, 3, 1>);
    fused_fns.push_back((const void *)k_csr_pass<F, L, PASS_UPDATE, 0, 4, 1>);
    fused_fns.push_back((const void *)k_csr_pass<F, L, PASS_UPDATE, 1, 4, 1>);
    fused_fns.push_back((const void *)k_csr_pass<F, L, PASS_UPDATE, 0, 5, 1>);
    fused_fns.push_back((const void *)k_csr_pass<F, L, PASS_UPDATE, 1, 5, 1>);
    fused_fns.push_back((const void *)k_csr_pass<F, L, PASS_UPDATE, 0, 6, 1>);
    fused_fns.push_back((const void *)k_csr_pass<F, L, PASS_UPDATE, 1, 6, 1>);
    fused_fns.push_back((const void *)k_csr_pass<F, L, PASS_UPDATE, 0, 7, 1>);
    fused_fns.push_back((const void *)k_csr_pass<F, L, PASS_UPDATE, 1, 7, 1>);
    fused_fns.push_back((const void *)k_csr_pass<F, L, PASS_UPDATE, 0, 8, 1>);
    fused_fns.push_back((const void *)k_csr_pass<F, L, PASS_UPDATE, 1, 8, 1>);
  }
  for (const void *fn : fused_fns)
    if (e == hipSuccess) e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if constexpr (L == 64) {
    std::vector<const void *> tiled_fns = {
        (const void *)k_csr_tile_pass<F, PASS_ALPHA, 0, 0>, (const void *)k_csr_tile_pass<F, PASS_ALPHA, 1, 0>,
        (const void *)k_csr_tile_pass<F, PASS_UPDATE, 0, 0>, (const void *)k_csr_tile_pass<F, PASS_UPDATE, 1, 0>,
#define TILE_RC(R)                                                                                              \
  (const void *)k_csr_tile_pass<F, PASS_ADOTS, 0, R>, (const void *)k_csr_tile_pass<F, PASS_ADOTS, 1, R>,       \
      (const void *)k_csr_tile_pass<F, PASS_UPDATE, 0, R>, (const void *)k_csr_tile_pass<F, PASS_UPDATE, 1, R>
        TILE_RC(1), TILE_RC(2), TILE_RC(3), TILE_RC(4), TILE_RC(5), TILE_RC(6), TILE_RC(7), TILE_RC(8)};
#undef TILE_RC
    for (const void *fn : tiled_fns)
      if (e == hipSuccess) e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    std::vector<const void *> ring_fns = {
        (const void *)k_csr_ring_pass<F, PASS_ALPHA, 0, 0>, (const void *)k_csr_ring_pass<F, PASS_ALPHA, 1, 0>,
        (const void *)k_csr_ring_pass<F, PASS_UPDATE, 0, 0>, (const void *)k_csr_ring_pass<F, PASS_UPDATE, 1, 0>,
        (const void *)k_csr_ring_pass<F, PASS_SPMM, 0, 0>, (const void *)k_csr_ring_pass<F, PASS_SPMM, 1, 0>,
#define RING_RC(R)                                                                                              \
  (const void *)k_csr_ring_pass<F, PASS_ADOTS, 0, R>, (const void *)k_csr_ring_pass<F, PASS_ADOTS, 1, R>,       \
      (const void *)k_csr_ring_pass<F, PASS_UPDATE, 0, R>, (const void *)k_csr_ring_pass<F, PASS_UPDATE, 1, R>
        RING_RC(1), RING_RC(2), RING_RC(3)};
#undef RING_RC
    for (const void *fn : ring_fns)
      if (e == hipSuccess) e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  return e;
}
static int set_kernel_attributes(slq_plan *p) {
  hipError_t ae = hipSuccess;
  DISPATCH(p->dtype, p->LPR, (ae = raise_lds_limits<F, L>()));
  HIP_TRY(ae);
  if (p->ring_gen || p->ring_deep) {
    hipError_t re = hipSuccess;
    const bool d = p->dtype == SLQ_F64;
    switch (p->LPR) {
      case 64: re = d ? slq_ring_prepare_f64_l64() : slq_ring_prepare_f32_l64(); break;
      case 32: re = d ? slq_ring_prepare_f64_l32() : slq_ring_prepare_f32_l32(); break;
      default: re = d ? slq_ring_prepare_f64_l16() : slq_ring_prepare_f32_l16(); break;
    }
    HIP_TRY(re);
  }
  // (never from inside a stream capture: launch_dense_mfma runs under one)
  HIP_TRY(hipFuncSetAttribute((const void *)k_dense_mfma_3term<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIP_TRY(hipFuncSetAttribute((const void *)k_dense_mfma_3term<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIP_TRY(hipFuncSetAttribute((const void *)k_dense_mfma_3term<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));

  return SLQ_OK;
}

// the plan's fused passes run on the operator's workgroup tiles (slq_plan_create decides: wide panels, and narrow ones of the ring-fed form)
static bool plan_tiled(const slq_plan *p) { return p->ringR > 0; }

// k_csr_ring_pass raises *ring_fail_d when one of its bounded waits ran out (slq_kernels.hpp: kRingSpinMax): everything the
// plan holds is then undefined. ring_flag_status() is the flag -> status translation (no HIP call in it: a CPU test covers
// it through slq_debug_ring_flag_status); check_ring_flag() reads the device word behind whatever the caller has enqueued
// and is called by EVERY accessor that hands results of a run to the host.
static int ring_flag_status(int flag) {
  if (flag == 2) {
    g_fa_broken.store(true);
    return fail(SLQ_EHIP, "the fused update + alpha pass found workgroups of one XCD slot on two XCDs (its hand-offs go through one XCD's L2): results are "
                          "invalid; plans created from now on use the separate alpha pass (SLQ_FUSED_ALPHA=0)");
  }
  if (flag) return fail(SLQ_EHIP, "the ring-fed tile pass gave up waiting on a tile (SLQ_TILES=2): results are invalid");
  return SLQ_OK;
}
static int check_ring_flag(slq_plan *p) {
  int flag = 0;
  HIP_TRY(hipMemcpyAsync(&flag, p->ring_fail_d, sizeof(int), hipMemcpyDeviceToHost, p->ctx->stream));
  HIP_TRY(hipStreamSynchronize(p->ctx->stream));
  return ring_flag_status(flag);
}
extern "C" int slq_debug_ring_flag_status(int flag) { return ring_flag_status(flag); }
// test hook: set the plan's device flag as an aborting workgroup would (tests/test_gpu_api.py)
extern "C" int slq_debug_plan_poke_ring_flag(slq_plan *p, int value) {
  if (!p) return fail(SLQ_EINVAL, "plan is NULL");
  HIP_TRY(hipSetDevice(p->ctx->device));
  HIP_TRY(hipMemcpyAsync(p->ring_fail_d, &value, sizeof(int), hipMemcpyHostToDevice, p->ctx->stream));
  HIP_TRY(hipStreamSynchronize(p->ctx->stream));
  return SLQ_OK;
}

// which launch sequence the steps with r <= kFusedMaxR take (enqueue_run): 0 sweeps, 1 recompute passes, 2 stored u
static int plan_sequence(const slq_plan *p) {
  const slq_operator *op = p->op;
  if (p->ring32_on) return 3;
  if (op->kind != OP_CSR || p->sw.fused == 0 || p->sw.mgs || p->nstale > 0) return 0;
  if (p->sw.fused == 2 || op->far_per_row <= 4.0) return (((p->gram && plan_tiled(p)) || p->gram_csr) && p->orth >= 1) ? 4 : 1;
  return (p->orth >= 1 && p->sw.stored_u && p->sw.merged && !plan_tiled(p)) ? 2 : 0;
}

extern "C" int slq_plan_describe(const slq_plan *p, slq_plan_info *out) {
  if (!p || !out) return fail(SLQ_EINVAL, "plan/out is NULL");
  out->panel_width = p->PW;
  out->panels = p->NP;
  out->ring_slots = p->S;
  out->sequence = plan_sequence(p);
  out->pipelined = p->pipelined ? 1 : 0;
  out->reordered = p->op->perm_d ? 1 : 0;
  out->upper_alpha = p->op->rowptr_u ? 1 : 0;
  out->far_per_row = p->op->far_per_row;
  out->tiles = plan_tiled(p) ? (p->op->tiles_ringed ? 2 : 1) : 0;
  out->fused_alpha = (p->fa_on && plan_sequence(p) == 4) ? 1 : 0;
  return SLQ_OK;
}

// Byte accounting of the store-and-revisit update sweep, which reads a ring column only when SOME probe of the panel projects on it (k_reorth_update): how many
// columns it read and how many it was offered, summed over launches and panels since the last reset. Synchronises.
extern "C" int slq_plan_sweep_columns(slq_plan *p, uint64_t *read, uint64_t *offered, int reset) {
  if (!p) return fail(SLQ_EINVAL, "plan is NULL");
  HIP_TRY(hipSetDevice(p->ctx->device));
  unsigned long long h[2] = {0, 0};
  HIP_TRY(hipMemcpyAsync(h, p->sweep_cols_d, sizeof(h), hipMemcpyDeviceToHost, p->ctx->stream));
  if (reset) HIP_TRY(hipMemsetAsync(p->sweep_cols_d, 0, sizeof(h), p->ctx->stream));
  HIP_TRY(hipStreamSynchronize(p->ctx->stream));
  if (read) *read = h[0];
  if (offered) *offered = h[1];
  return SLQ_OK;
}

extern "C" int slq_plan_workspace_bytes(const slq_plan *plan, size_t *bytes) {
  if (!plan || !bytes) return fail(SLQ_EINVAL, "plan/bytes is NULL");
  *bytes = plan->bytes;
  return SLQ_OK;
}

// staging buffer of `cols` column-major columns
static int ensure_stage(slq_plan *p, int cols) {
  if (p->stage && p->stage_cols >= cols) return SLQ_OK;
  if (p->stage) { hipFree(p->stage); p->stage = nullptr; }
  HIP_TRY(hipMalloc(&p->stage, (size_t)cols * p->n * p->esz));
  p->stage_cols = cols;
  return SLQ_OK;
}

static int stage_chunk_cols(const slq_plan *p) {
  // ~256 MiB of staging at most
  const size_t per_col = (size_t)p->n * p->esz;
  size_t c = std::max<size_t>(1, ((size_t)256 << 20) / per_col);
  return (int)std::min<size_t>(c, (size_t)p->bpad);
}

// ||v||^2 of slot 0 -> nu_0, activity, first coefficients
// unit_entries: every entry of every probe is +-1 (Rademacher probes drawn by k_gen_probes): ||v||^2 = n is known, the norm sweep - one read of
// the panel, 0.38 ms of configs[1]'s 55 - is skipped (r04; bitwise the same nu_0: the sweep's partial sums are exact integers)
static int init_from_probes(slq_plan *p, int sphere, bool unit_entries = false) {
  hipStream_t st = p->ctx->stream;
  dim3 g(p->nblkS, p->NP);
  if (!unit_entries)
    PROFILED(p, SLQ_K_AXPY_NORM,
             DISPATCH(p->dtype, p->LPR,
                      (k_axpy_norm<F, L, 1><<<g, dim3(kBlock), 0, st>>>(p->n,
                                          (F *)slot_ptr(p, 0), (const F *)nullptr,
                                          (const double *)nullptr, p->part, p->bpad))));
  PROFILED(p, SLQ_K_FINALIZE,
           hipLaunchKernelGGL(k_fin_init, dim3((p->bpad + 63) / 64), dim3(kFinThreads), 0, st, p->st, p->part,
                              unit_entries ? 0 : p->nblkS, sphere, (double)p->n));
  if (p->ring32_on) {  // vector 0 joins the fp32 archive
    switch (p->LPR) {
      case 64: k_archive32<64><<<g, dim3(kBlock), 0, st>>>(p->n, (const double *)slot_ptr(p, 0), p->ring32); break;
      case 32: k_archive32<32><<<g, dim3(kBlock), 0, st>>>(p->n, (const double *)slot_ptr(p, 0), p->ring32); break;
      case 16: k_archive32<16><<<g, dim3(kBlock), 0, st>>>(p->n, (const double *)slot_ptr(p, 0), p->ring32); break;
      default: k_archive32<8><<<g, dim3(kBlock), 0, st>>>(p->n, (const double *)slot_ptr(p, 0), p->ring32); break;
    }
  }
  HIP_TRY(hipGetLastError());
  p->probes_ready = true;
  p->ran = false;
  return SLQ_OK;
}

// two pinned buffers of `bytes` each on the context (kept for its lifetime); false when the host cannot pin that much
static bool ensure_pinned(slq_context *ctx, size_t bytes) {
  if (ctx->pin_bytes >= bytes && ctx->pin[0] && ctx->pin[1]) return true;
  for (int b = 0; b < 2; ++b) {
    if (ctx->pin[b]) hipHostFree(ctx->pin[b]);
    ctx->pin[b] = nullptr;
    ctx->pin_busy[b] = false;
  }
  ctx->pin_bytes = 0;
  for (int b = 0; b < 2; ++b) {
    if (hipHostMalloc(&ctx->pin[b], bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (!ctx->pin_ev[b] && hipEventCreateWithFlags(&ctx->pin_ev[b], hipEventDisableTiming) != hipSuccess) return false;
  }
  ctx->pin_bytes = bytes;
  return true;
}

extern "C" int slq_plan_set_probes(slq_plan *p, const void *X, int64_t ldx) {
  if (!p || !X) return fail(SLQ_EINVAL, "plan/X is NULL");
  if (ldx < p->n) return fail(SLQ_EINVAL, "ldx (%lld) < n (%d)", (long long)ldx, p->n);
  HIP_TRY(hipSetDevice(p->ctx->device));
  hipStream_t st = p->ctx->stream;
  slq_context *ctx = p->ctx;
  // The caller's array is pageable memory: copied straight from there, the runtime bounces it through its own small
  // pinned buffers at ~10 GB/s (configs[1]: 2 GB of probes, 0.2 s). Here the columns are copied by a few host threads
  // into one of two pinned buffers (32 MiB chunks) and sent from there; the copy of chunk i + 1 overlaps the transfer
  // and the transposition kernel of chunk i. SLQ_PINNED_UPLOAD=0 keeps the direct copy.
  const size_t col_bytes = (size_t)p->n * p->esz;
  const bool pinned = env_int("SLQ_PINNED_UPLOAD", 1) != 0 && (size_t)p->nprobes * col_bytes >= ((size_t)4 << 20);
  int cc = stage_chunk_cols(p);
  if (pinned) cc = (int)std::max<size_t>(1, std::min<size_t>((size_t)cc, ((size_t)32 << 20) / col_bytes));
  const bool use_pin = pinned && ensure_pinned(ctx, (size_t)cc * col_bytes);
  if (!use_pin) cc = stage_chunk_cols(p);
  SLQ_TRY(ensure_stage(p, cc));
  // padding columns must be zero
  if (p->nprobes < p->bpad)
    HIP_TRY(hipMemsetAsync(slot_ptr(p, 0), 0, (size_t)p->slot_stride * p->esz, st));
  int buf = 0;
  for (int c0 = 0; c0 < p->nprobes; c0 += cc) {
    const int nc = std::min(cc, p->nprobes - c0);
    const char *src = (const char *)X + (size_t)c0 * (size_t)ldx * p->esz;
    if (use_pin) {
      if (ctx->pin_busy[buf]) HIP_TRY(hipEventSynchronize(ctx->pin_ev[buf]));  // its previous transfer has left the buffer
      char *dst = (char *)ctx->pin[buf];
      // rows of the chunk cut into pieces: every thread copies its row range of every column (contiguous runs)
      if (!parallel_pieces(host_threads(), p->n, [&](int, int64_t r0, int64_t r1) {
            for (int c = 0; c < nc; ++c)
              memcpy(dst + (size_t)c * col_bytes + (size_t)r0 * p->esz, src + (size_t)c * (size_t)ldx * p->esz + (size_t)r0 * p->esz, (size_t)(r1 - r0) * p->esz);
          }))
        return fail(SLQ_ENOMEM, "host worker failed while staging the probes");
      HIP_TRY(hipMemcpyAsync(p->stage, dst, (size_t)nc * col_bytes, hipMemcpyHostToDevice, st));
      HIP_TRY(hipEventRecord(ctx->pin_ev[buf], st));
      ctx->pin_busy[buf] = true;
      buf ^= 1;
    } else {
      HIP_TRY(hipMemcpy2DAsync(p->stage, col_bytes, src, (size_t)ldx * p->esz, col_bytes, (size_t)nc, hipMemcpyHostToDevice, st));
    }
    dim3 g((p->n + 63) / 64, (nc + 63) / 64);
    PROFILED(p, SLQ_K_PROBES, {
      if (p->dtype == SLQ_F64)
        hipLaunchKernelGGL(k_cols_to_panel<double>, g, dim3(256), 0, st, p->n, (const double *)p->stage, c0, nc, (double *)slot_ptr(p, 0), p->PW, p->op->perm_d);
      else
        hipLaunchKernelGGL(k_cols_to_panel<float>, g, dim3(256), 0, st, p->n, (const float *)p->stage, c0, nc, (float *)slot_ptr(p, 0), p->PW, p->op->perm_d);
    });
    if (!use_pin) HIP_TRY(hipStreamSynchronize(st));  // the caller's memory and the staging buffer are reused by the next chunk
  }
  // (pinned path: everything of X has been read when the loop ends; the device staging buffer is reused in stream order)
  p->pdf_sphere = 0;
  return init_from_probes(p, 0);
}

extern "C" int slq_plan_generate_probes(slq_plan *p, int pdf, uint64_t seed, uint64_t probe_offset) {
  if (!p) return fail(SLQ_EINVAL, "plan is NULL");
  if (pdf < 0 || pdf > 2) return fail(SLQ_EINVAL, "Invalid distribution id %d supplied.", pdf);
  HIP_TRY(hipSetDevice(p->ctx->device));
  hipStream_t st = p->ctx->stream;
  const int RPW = 64 / p->LPR;
  const int items = (p->n + (pdf == 0 ? 127 : 1)) / (pdf == 0 ? 128 : 2);
  const int gx = std::max(1, std::min(p->ctx->num_cus * 8, (items + 4 * RPW - 1) / (4 * RPW)));
  dim3 g(gx, p->NP);
  // (an operator stored as P A P^T: the rows are scattered through the inverse permutation - the same (seed, id, row) stream)
  PROFILED(p, SLQ_K_PROBES,
           DISPATCH(p->dtype, p->LPR,
                    (k_gen_probes<F, L><<<g, dim3(256), 0, st>>>(p->n,
                                        (F *)slot_ptr(p, 0), pdf, seed, probe_offset, p->nprobes, p->op->inv_perm_d))));
  p->pdf_sphere = (pdf == SLQ_PDF_SPHERE);
  return init_from_probes(p, p->pdf_sphere, pdf == 0 && env_int("SLQ_KNOWN_NORM", 1) != 0);
}

// copy columns [c0, c0+nc) of `slot` to a host column-major array, optional per-column scale
static int panel_to_host(slq_plan *p, int slot, int c0, int nc, void *X, int64_t ldx,
                         const double *d_scale) {
  hipStream_t st = p->ctx->stream;
  const int cc = stage_chunk_cols(p);
  SLQ_TRY(ensure_stage(p, cc));
  for (int o = 0; o < nc; o += cc) {
    const int m = std::min(cc, nc - o);
    dim3 g((p->n + 63) / 64, (m + 63) / 64);
    if (p->dtype == SLQ_F64)
      hipLaunchKernelGGL(k_panel_to_cols<double>, g, dim3(256), 0, st, p->n, (const double *)slot_ptr(p, slot), c0 + o, m, (double *)p->stage, p->PW, d_scale, p->op->perm_d);
    else
      hipLaunchKernelGGL(k_panel_to_cols<float>, g, dim3(256), 0, st, p->n, (const float *)slot_ptr(p, slot), c0 + o, m, (float *)p->stage, p->PW, d_scale, p->op->perm_d);
    HIP_TRY(hipMemcpy2DAsync((char *)X + (size_t)o * (size_t)ldx * p->esz, (size_t)ldx * p->esz, p->stage,
                             (size_t)p->n * p->esz, (size_t)p->n * p->esz, (size_t)m, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
  }
  return SLQ_OK;
}

extern "C" int slq_plan_get_probes(slq_plan *p, void *X, int64_t ldx) {
  if (!p || !X) return fail(SLQ_EINVAL, "plan/X is NULL");
  if (!p->probes_ready) return fail(SLQ_EINVAL, "no probes set, or they were consumed by a run");
  if (ldx < p->n) return fail(SLQ_EINVAL, "ldx < n");
  HIP_TRY(hipSetDevice(p->ctx->device));
  return panel_to_host(p, 0, 0, p->nprobes, X, ldx, nullptr);
}


// dense operator on MFMA (fp64, and fp32 through k_dense_mfma32_lds): Wn = sc*(A Wc) - cp*Wp with alpha partials (plain = 0), or Wn = A Wc
static int launch_dense_mfma(slq_plan *p, const void *Wc, const void *Wp, void *Wn, int first, int plain, int *nblk_out) {
  const slq_operator *op = p->op;
  hipStream_t st = p->ctx->stream;
  // panels of 32+ columns: big tiles, K split over workgroups, epilogue by k_3term_slabs (k_dense_mfma_tile);
  // 16-column panels: the 16-row kernel with its fused epilogue. SLQ_DENSE_TILE16=1 forces the latter (A/B runs).
  if (p->dense_ks > 0 && p->dtype == SLQ_F32) {
    const int ks = p->dense_ks;
    float *raw = (float *)p->T + p->slot_stride;  // slabs 1..ks of T (slab 0 is the unfused product)
    const dim3 g((p->n + kDense32BM - 1) / kDense32BM, p->NP, ks);
    const int a_vec = op->lda % 4 == 0 && ((uintptr_t)op->vals & 15) == 0;
    for (int col0 = 0; col0 < p->PW; col0 += kDense32BN)  // PW = 128 / 256: A streamed once per 64 columns (32 flop per byte of A: still MFMA-bound)
      k_dense_mfma32_lds<<<g, dim3(kBlock), 0, st>>>(p->n, (const float *)op->vals, op->lda, a_vec, (const float *)Wc, p->PW, col0, raw, p->slot_stride);
    const dim3 gS(p->nblkS, p->NP);
    DISPATCH(SLQ_F32, p->LPR,
             (k_3term_slabs<F, L><<<gS, dim3(kBlock), 0, st>>>(p->n, (const F *)raw, ks, p->slot_stride, (const F *)Wc, (const F *)Wp, (F *)Wn,
                                                             p->st.coefA, p->part, p->bpad, first, plain)));
    if (nblk_out) *nblk_out = p->nblkS;
    return SLQ_OK;
  }
  if (p->dense_ks > 0) {
    const int ks = p->dense_ks;
    double *raw = (double *)p->T + p->slot_stride;  // slabs 1..ks of T (slab 0 is the unfused product)
    const int ncg = p->PW >= 64 ? 2 : 1;
    const int rw = 32 * (kWaves / ncg);
    const dim3 g((p->n + rw - 1) / rw, p->NP, ks);
    const bool lds_form = p->sw.dense_lds && (op->lda % 2 == 0);  // operands staged in LDS once per workgroup (16-byte aligned row pairs)
    for (int col0 = 0; col0 < p->PW; col0 += 32 * ncg) {  // PW = 128: two 64-column halves, A streamed twice
      if (lds_form && ncg == 2)  // (panels of 64+ columns; 32-column panels keep the register form: their 256-row block does not fit static LDS)
        k_dense_mfma_lds<2><<<g, dim3(kBlock), 0, st>>>(p->n, (const double *)op->vals, op->lda, (const double *)Wc, p->PW, col0, raw, p->slot_stride);
      else if (ncg == 2)
        k_dense_mfma_tile<2><<<g, dim3(kBlock), 0, st>>>(p->n, (const double *)op->vals, op->lda, (const double *)Wc, p->PW, col0, raw, p->slot_stride);
      else
        k_dense_mfma_tile<1><<<g, dim3(kBlock), 0, st>>>(p->n, (const double *)op->vals, op->lda, (const double *)Wc, p->PW, col0, raw, p->slot_stride);
    }
    // second stage: sum the slabs in slab order, three-term epilogue and alpha partials (or the plain product)
    const dim3 gS(p->nblkS, p->NP);
    DISPATCH(SLQ_F64, p->LPR,
             (k_3term_slabs<F, L><<<gS, dim3(kBlock), 0, st>>>(p->n, (const F *)raw, ks, p->slot_stride, (const F *)Wc, (const F *)Wp, (F *)Wn,
                                                             p->st.coefA, p->part, p->bpad, first, plain)));
    if (nblk_out) *nblk_out = p->nblkS;
    return SLQ_OK;
  }
  const size_t part_rows = (size_t)kReorthChunk * std::max(std::max(std::max(std::max(p->nblkA, p->nblkF), p->nblkU), p->nblkS), p->nblkT);
  const int nblk = (p->n + 15) / 16;
  if ((size_t)nblk > part_rows) return fail(SLQ_EINVAL, "dense operator too large for the partials buffer");
  const dim3 g(nblk, p->NP);
#define DENSE_LAUNCH(TWV, COL0)                                                                             \
  {                                                                                                         \
    const size_t lds = ((size_t)kWaves * (TWV / 16) * 4 * 64 + (size_t)TWV * 4) * sizeof(double);          \
    k_dense_mfma_3term<TWV><<<g, dim3(kBlock), lds, st>>>(p->n, (const double *)op->vals, op->lda, (const double *)Wc,        \
                                                        (const double *)Wp, (double *)Wn, p->st.coefA, p->part, p->bpad, first, plain, p->PW, COL0); \
  }
  switch (p->PW) {
    case 128: DENSE_LAUNCH(64, 0) DENSE_LAUNCH(64, 64) break;
    case 64: DENSE_LAUNCH(64, 0) break;
    case 32: DENSE_LAUNCH(32, 0) break;
    default: DENSE_LAUNCH(16, 0) break;
  }
#undef DENSE_LAUNCH
  if (nblk_out) *nblk_out = nblk;
  return SLQ_OK;
}

// T = A * (slot c), unscaled, for operators without a fused kernel
static int apply_operator_unfused(slq_plan *p, int slot_c) {
  hipStream_t st = p->ctx->stream;
  slq_operator *op = p->op;
  if (op->kind == OP_DENSE && p->sw.dense_mfma && (p->dtype == SLQ_F64 || p->dense_ks > 0)) {
    PROFILED(p, SLQ_K_SPMM, SLQ_TRY(launch_dense_mfma(p, slot_ptr(p, slot_c), nullptr, p->T, 1, 1, nullptr)));
    return SLQ_OK;
  }
  if (op->kind == OP_DENSE) {
    dim3 g(std::max(1, std::min(p->ctx->num_cus * 2, (p->n + kWaves - 1) / kWaves)), p->NP);
    PROFILED(p, SLQ_K_SPMM,
             DISPATCH(p->dtype, p->LPR,
                      (k_dense_panel<F, L><<<g, dim3(kBlock), 0, st>>>(p->n,
                                          (const F *)(op->vals_t ? op->vals_t : op->vals), op->lda, (const F *)slot_ptr(p, slot_c), (F *)p->T))));
    return SLQ_OK;
  }
  if (op->kind == OP_GRAM) {
    // T = A^T (A W_c): two plain panel SpMMs through an mrows-row panel (eigen_operators.h:66-72, gram = true)
    const dim3 g1(std::max(1, std::min(p->nblkS, (int)((op->mrows + kWaves - 1) / kWaves))), p->NP), g2(p->nblkS, p->NP);
    PROFILED(p, SLQ_K_SPMM,
             DISPATCH(p->dtype, p->LPR,
                      (k_spmm_plain<F, L><<<g1, dim3(kBlock), 0, st>>>((int)op->mrows, op->rowptr, op->colind, (const F *)op->vals,
                                                                   (const F *)slot_ptr(p, slot_c), (F *)p->T2, p->n))));
    PROFILED(p, SLQ_K_SPMM,
             DISPATCH(p->dtype, p->LPR,
                      (k_spmm_plain<F, L><<<g2, dim3(kBlock), 0, st>>>(p->n, op->rowptr_t, op->colind_t, (const F *)op->vals_t,
                                                                   (const F *)p->T2, (F *)p->T, (int)op->mrows))));
    return SLQ_OK;
  }
  if (op->kind == OP_DEVICE_CALLBACK) {
    // device plugin: the panel goes through two column-major staging buffers in HBM (X | Y); the plugin
    // enqueues Y = A X on our stream (or synchronises itself) - nothing crosses PCIe
    const int cc = std::max(1, stage_chunk_cols(p) / 2);
    SLQ_TRY(ensure_stage(p, 2 * cc));
    char *sx = (char *)p->stage, *sy = sx + (size_t)cc * p->n * p->esz;
    HIP_TRY(hipMemsetAsync(p->T, 0, (size_t)p->slot_stride * p->esz, st));
    for (int c0 = 0; c0 < p->nprobes; c0 += cc) {
      const int nc = std::min(cc, p->nprobes - c0);
      dim3 g((p->n + 63) / 64, (nc + 63) / 64);
      if (p->dtype == SLQ_F64)
        hipLaunchKernelGGL(k_panel_to_cols<double>, g, dim3(256), 0, st, p->n, (const double *)slot_ptr(p, slot_c), c0, nc, (double *)sx, p->PW, (const double *)nullptr, (const int32_t *)nullptr);
      else
        hipLaunchKernelGGL(k_panel_to_cols<float>, g, dim3(256), 0, st, p->n, (const float *)slot_ptr(p, slot_c), c0, nc, (float *)sx, p->PW, (const double *)nullptr, (const int32_t *)nullptr);
      if (op->dev_fn(op->user, sx, sy, p->n, nc, (void *)st) != 0)
        return fail(SLQ_ECALLBACK, "device operator callback failed on columns %d..%d", c0, c0 + nc - 1);
      if (p->dtype == SLQ_F64)
        hipLaunchKernelGGL(k_cols_to_panel<double>, g, dim3(256), 0, st, p->n, (const double *)sy, c0, nc, (double *)p->T, p->PW, (const int32_t *)nullptr);
      else
        hipLaunchKernelGGL(k_cols_to_panel<float>, g, dim3(256), 0, st, p->n, (const float *)sy, c0, nc, (float *)p->T, p->PW, (const int32_t *)nullptr);
      HIP_TRY(hipGetLastError());
    }
    return SLQ_OK;
  }
  // host callback: device -> host, one matvec per probe, host -> device
  const size_t colb = (size_t)p->n * p->esz;
  if (p->hbuf.size() < 2 * colb * (size_t)p->nprobes) p->hbuf.resize(2 * colb * (size_t)p->nprobes);
  char *hx = p->hbuf.data(), *hy = hx + colb * p->nprobes;
  SLQ_TRY(panel_to_host(p, slot_c, 0, p->nprobes, hx, p->n, nullptr));
  for (int i = 0; i < p->nprobes; ++i)
    if (op->fn(op->user, hx + colb * i, hy + colb * i) != 0)
      return fail(SLQ_ECALLBACK, "operator callback failed on probe %d", i);
  const int cc = stage_chunk_cols(p);
  HIP_TRY(hipMemsetAsync(p->T, 0, (size_t)p->slot_stride * p->esz, st));
  for (int c0 = 0; c0 < p->nprobes; c0 += cc) {
    const int nc = std::min(cc, p->nprobes - c0);
    HIP_TRY(hipMemcpyAsync(p->stage, hy + colb * c0, colb * nc, hipMemcpyHostToDevice, st));
    dim3 g((p->n + 63) / 64, (nc + 63) / 64);
    if (p->dtype == SLQ_F64)
      hipLaunchKernelGGL(k_cols_to_panel<double>, g, dim3(256), 0, st, p->n, (const double *)p->stage, c0, nc, (double *)p->T, p->PW, p->op->perm_d);
    else
      hipLaunchKernelGGL(k_cols_to_panel<float>, g, dim3(256), 0, st, p->n, (const float *)p->stage, c0, nc, (float *)p->T, p->PW, p->op->perm_d);
    HIP_TRY(hipStreamSynchronize(st));
  }
  return SLQ_OK;
}

static int launch_reorth_update(slq_plan *p, int j, int r, int istart, bool axpy = false, int klass = SLQ_K_REORTH_UPD);
static int launch_reorth_update_range(slq_plan *p, int j, int ibegin, int iend, bool axpy = false, int klass = SLQ_K_REORTH_UPD);
static int update_chunk_cols(const slq_plan *p);

// probes per workgroup of the QL kernel: 3*deg*lanes doubles of LDS, at most 150 KiB
static int quadrature_lanes(int deg) {
  int lanes = (int)((150 * 1024) / ((size_t)3 * deg * 8));
  return std::max(1, std::min(64, lanes));
}

// the generic update pass of the Gram sequence (k_csr_pass<PASS_UPDATEG>, nontemporal streams), r = 1 .. kFusedMaxR ring columns
template <typename F, int L, int RC> static inline void launch_csr_updateg_rc(slq_plan *p, bool pipe_on, dim3 grid, size_t lds, hipStream_t st, int j);
template <typename F, int L> static inline void launch_csr_updateg(slq_plan *p, int r, bool pipe_on, dim3 grid, size_t lds, hipStream_t st, int j) {
  switch (r) {
    case 1: launch_csr_updateg_rc<F, L, 1>(p, pipe_on, grid, lds, st, j); break;
    case 2: launch_csr_updateg_rc<F, L, 2>(p, pipe_on, grid, lds, st, j); break;
    case 3: launch_csr_updateg_rc<F, L, 3>(p, pipe_on, grid, lds, st, j); break;
    case 4: launch_csr_updateg_rc<F, L, 4>(p, pipe_on, grid, lds, st, j); break;
    case 5: launch_csr_updateg_rc<F, L, 5>(p, pipe_on, grid, lds, st, j); break;
    case 6: launch_csr_updateg_rc<F, L, 6>(p, pipe_on, grid, lds, st, j); break;
    case 7: launch_csr_updateg_rc<F, L, 7>(p, pipe_on, grid, lds, st, j); break;
    default: launch_csr_updateg_rc<F, L, 8>(p, pipe_on, grid, lds, st, j); break;
  }
}

// one dots sweep of the store-and-revisit sequence, with or without the fp32 archive
// (deferred: the block-CGS sweeps leave `w -= cB W_c` to the update sweep - every dots chunk applies it in registers and stores
// nothing; the fp32-archive kernels keep the stored form)
template <typename F, int L> static inline void launch_reorth_dot(slq_plan *p, dim3 gS, hipStream_t st, int j, int i0, int rc) {
  if constexpr (std::is_same<F, double>::value) {
    if (p->ring32_on) {
      k_reorth_dot32<L><<<gS, dim3(kBlock), 0, st>>>(p->n, (double *)p->ring, p->slot_stride, p->S, j, i0, rc, (int)(i0 == 0), p->st.coefB,
                                                    p->part, p->bpad, p->ring32, p->slot_stride, p->S32);
      return;
    }
  }
  k_reorth_dot<F, L><<<gS, dim3(kBlock), 0, st>>>(p->n, (F *)p->ring, p->slot_stride, p->S, j, i0, rc, (p->sw.defer_axpy && !p->ring32_on) ? 2 : (int)(i0 == 0), p->st.coefB, p->part,
                                                  p->bpad);
}
template <typename F, int L>
static inline void launch_reorth_update_kernel(slq_plan *p, dim3 gS, size_t lds, hipStream_t st, int j, int i0, int rc, int archive, bool axpy = false) {
  if constexpr (std::is_same<F, double>::value) {
    if (p->ring32_on) {
      const size_t lds32 = sizeof(double) * kWaves * 64 * 4 + (size_t)rc * p->PW * sizeof(double) + (2 * (size_t)rc + 1) * sizeof(int);
      k_reorth_update32<L><<<gS, dim3(kBlock), lds32, st>>>(p->n, (double *)p->ring, p->slot_stride, p->S, j, i0, rc,
                                                          p->st.gamma + (size_t)i0 * p->bpad, p->part, p->bpad, p->ring32, p->slot_stride, p->S32, archive,
                                                          p->sweep_cols_d, p->sweep_skip ? 1 : 0);
      return;
    }
  }
  k_reorth_update<F, L><<<gS, dim3(kBlock), lds, st>>>(p->n, (F *)p->ring, p->slot_stride, p->S, j, i0, rc, p->st.gamma + (size_t)i0 * p->bpad,
                                                     p->part, p->bpad, axpy ? p->st.coefB : nullptr, p->sweep_cols_d, p->sweep_skip ? 1 : 0);
}

// one fused CSR pass; the pipelined row loop exists for one-row-per-wave panels (L == 64) and not for the alpha pass
template <typename F, int L, int PASS, int LP, int RC, typename... Args>
static inline void launch_csr_pass(bool pipe_on, dim3 grid, size_t lds, hipStream_t st, Args... args) {
  if constexpr (L == 64 && PASS != PASS_ALPHA) {
    if (pipe_on) {
      k_csr_pass<F, L, PASS, LP, RC, 1><<<grid, dim3(kBlock), lds, st>>>(args...);
      return;
    }
  }
  k_csr_pass<F, L, PASS, LP, RC, 0><<<grid, dim3(kBlock), lds, st>>>(args...);
}

template <typename F, int L, int RC> static inline void launch_csr_updateg_rc(slq_plan *p, bool pipe_on, dim3 grid, size_t lds, hipStream_t st, int j) {
  const slq_operator *op = p->op;
  const int xt = (j == p->deg - 1 && !p->keep_basis && p->last_nostore) ? 16 : 0;  // (the run's last step: W_deg is not stored)
  launch_csr_pass<F, L, PASS_UPDATEG, 1, RC>(pipe_on, grid, lds, st, p->n, op->rowptr, op->colind, (const F *)op->vals, (F *)p->ring, p->slot_stride, p->S, j, p->st.coefA,
                                             p->st.coefB, p->st.gamma, p->part, p->bpad, xt);
}

// the same pass on workgroup tiles (wide panels only; slq_kernels.hpp: k_csr_tile_pass)
template <typename F, int L, int PASS, int LP, int RC>
static inline void launch_tile_pass(slq_plan *p, dim3 grid, size_t lds, hipStream_t st, int j, int xt) {
  if constexpr (L == 64 && (PASS == PASS_ALPHA || PASS == PASS_ADOTS || PASS == PASS_UPDATE || PASS == PASS_SPMM)) {
    const slq_operator *op = p->op;
    TileRanges xr;
    for (int x = 0; x < 9; ++x) xr.first[x] = op->tiles.xcd_tile[x];
    if constexpr (RC <= kRingMaxR) {
      if (op->tiles_ringed) {
        // the ring-fed variant: flag words and descriptor staging + kRingSlots slots; 16 waves per workgroup
        const size_t lds_ring = kRingHeadBytes + (size_t)kRingSlots * (kRingTileCols * 1024 + kRingMetaBytes);
        const bool upper = PASS == PASS_ALPHA && p->rs_desc_u != nullptr && p->sw.ring_alpha == 2;
        if (upper)
          for (int x = 0; x < 9; ++x) xr.first[x] = p->rs_xcd_u[x];
        k_csr_ring_pass<F, PASS, LP, RC><<<grid, dim3(kRingBlock), lds_ring, st>>>(p->n, upper ? p->rs_desc_u : op->tile_desc, upper ? p->rs_rec_u : op->tile_rec, xr, (F *)p->ring, p->slot_stride, p->S, j, p->st.coefA,
                                                                               p->st.coefB, p->st.gamma, p->part, p->bpad,
                                                                               xt | ((PASS == PASS_UPDATE && p->sw.ring_rev) ? 4 : 0), p->ring_fail_d);
        return;
      }
    }
    if (op->tiles_ringed) {  // never reached (enqueue_run sends ring-sized tiles to the ring-fed kernels or the generic passes)
      p->launch_error = true;  // (no launch: enqueue_run turns this into SLQ_EINVAL instead of handing out numbers of a pass that never ran)
      return;
    }
    if constexpr (PASS != PASS_SPMM)
    k_csr_tile_pass<F, PASS, LP, RC><<<grid, dim3(kBlock), lds, st>>>(p->n, op->rowptr, (const F *)op->vals, op->tiles.tile_row, op->tiles.tile_ptr,
                                                                    op->tiles.tile_cols, op->tiles.lcol, op->tiles.self_idx, xr, op->tiles.max_cols, (F *)p->ring,
                                                                    p->slot_stride, p->S, j, p->st.coefA, p->st.coefB, p->st.gamma, p->part, p->bpad, xt);
  }
}

#ifdef SLQ_DEBUG_TIMES
static unsigned long long *g_dbg_host_handle = nullptr;
static unsigned long long *debug_times_buffer() { return g_dbg_host_handle; }
#else
static unsigned long long *debug_times_buffer() { return nullptr; }
#endif

// one ring-fed pass through k_ring_pass (slq_ring.hpp: any panel width, up to 8 ring columns; nontemporal streams)
static int launch_ring_gen(slq_plan *p, int pass, int rc, dim3 grid, hipStream_t st, int j, int xt) {
  const bool upper = pass == PASS_ALPHA && p->rs_desc_u != nullptr && p->sw.ring_alpha == 2;
  RingArgs a;
  a.pass = pass;
  a.rc = rc;
  a.staged = (pass == PASS_ALPHA && p->ring_staged) ? 1 : 0;
  a.grid = grid;
  a.st = st;
  a.n = p->n;
  a.desc = upper ? p->rs_desc_u : p->rs_desc;
  a.rec = upper ? p->rs_rec_u : p->rs_rec;
  for (int x = 0; x < 9; ++x) a.xr.first[x] = upper ? p->rs_xcd_u[x] : p->rs_xcd[x];
  a.ring = p->ring;
  a.slot_stride = p->slot_stride;
  a.S = p->S;
  a.j = j;
  a.coefA = p->st.coefA;
  a.coefB = p->st.coefB;
  a.gamma = p->st.gamma;
  a.part = p->part;
  a.bpad = p->bpad;
  // bit 4: the LAST step of a run whose basis is not kept stores nothing - W_deg is never read (lanczos.h:139-142 takes its norm and breaks); SLQ_LAST_STORE=1 keeps the store
  const bool last_nostore = (pass == PASS_UPDATE || pass == PASS_UPDATEG) && j == p->deg - 1 && !p->keep_basis && p->last_nostore;
  a.xt = xt | (((pass == PASS_UPDATE || pass == PASS_UPDATEG) && p->sw.ring_rev) ? 4 : 0) | ((upper && p->rs_u_padded) ? 8 : 0) | (last_nostore ? 16 : 0);  // bit 3: padded rows
  a.fail = p->ring_fail_d;
  a.dbg = pass == env_int("SLQ_DEBUG_PASS", PASS_ADOTS) ? debug_times_buffer() : nullptr;  // (diagnostic builds: the pass whose time line is stamped)
  const bool d = p->dtype == SLQ_F64;
  int rc_l = -1;
  switch (p->LPR) {
    case 64: rc_l = d ? slq_ring_launch_f64_l64(a) : slq_ring_launch_f32_l64(a); break;
    case 32: rc_l = d ? slq_ring_launch_f64_l32(a) : slq_ring_launch_f32_l32(a); break;
    case 16: rc_l = d ? slq_ring_launch_f64_l16(a) : slq_ring_launch_f32_l16(a); break;
    default: break;
  }
  if (rc_l != 0) return fail(SLQ_EINVAL, "no ring-fed kernel for pass %d with %d ring columns on %d lanes per row", pass, rc, p->LPR);
  return SLQ_OK;
}

// alpha's share of the upper-triangle entries that cross XCD chunks (k_alpha_edges), after a fused update + alpha pass
template <typename F, int L> static inline void launch_alpha_edges(slq_plan *p, int nblk_e, hipStream_t st, int slot, double *part_e) {
  if constexpr (L == 64) {
    const slq_operator *op = p->op;
    k_alpha_edges<F, L><<<dim3(nblk_e, p->NP), dim3(kBlock), 0, st>>>(p->n, op->fa.nedges, op->fa.er, op->fa.ec, (const F *)op->fa.ev, (const F *)slot_ptr(p, slot), part_e,
                                                                      p->bpad);
  }
}

// the ring-fed update pass of step j with step j + 1's alpha dot fused in (slq_ring_fa.hpp); gen: its number within the run
static int launch_ring_fa(slq_plan *p, int pass, int rc, dim3 grid, hipStream_t st, int j, int gen) {
  RingArgs a;
  a.pass = pass;
  a.rc = rc;
  a.staged = 0;
  a.grid = grid;
  a.st = st;
  a.n = p->n;
  a.desc = p->rs_desc;
  a.rec = p->rs_rec;
  a.desc_a = p->op->fa.desc;
  a.rec_a = p->op->fa.rec;
  for (int x = 0; x < 9; ++x) a.xr.first[x] = p->rs_xcd[x];
  a.ring = p->ring;
  a.slot_stride = p->slot_stride;
  a.S = p->S;
  a.j = j;
  a.coefA = p->st.coefA;
  a.coefB = p->st.coefB;
  a.gamma = p->st.gamma;
  a.part = p->part;
  a.bpad = p->bpad;
  a.xt = env_int("SLQ_FA_MODE", 0);  // (timing experiments only: 1 skips the counter poll, 2 the alpha items)
  a.fail = p->ring_fail_d;
  a.dbg = nullptr;
  a.cnt = p->fa_cnt;
  a.cnt_rounds = p->fa_rounds;
  a.gen = gen;
  a.xcc_tab = p->fa_cnt + (size_t)p->NP * 8 * p->fa_rounds;
  const int rc_l = p->dtype == SLQ_F64 ? slq_ring_fa_launch_f64_l64(a) : slq_ring_fa_launch_f32_l64(a);
  if (rc_l != 0) return fail(SLQ_EINVAL, "no fused update + alpha kernel for pass %d with %d ring columns", pass, rc);
  return SLQ_OK;
}

// enqueue the deg-step launch sequence on the context stream (also run under stream capture)
static int enqueue_run(slq_plan *p, double rtol, int fused_mode, bool nt) {
  if (p->ring32_on && p->nstale > 0) return fail(SLQ_EINVAL, "SLQ_RING32 does not combine with preloaded stale ring columns");
  const bool fused = fused_mode != 0 && !p->ring32_on;  // the fused passes do not feed the fp32 archive
  hipStream_t st = p->ctx->stream;
  const int bp = p->bpad, deg = p->deg, S = p->S;
  const double eps = p->dtype == SLQ_F64 ? std::numeric_limits<double>::epsilon()
                                         : (double)std::numeric_limits<float>::epsilon();
  const double residual_tol = std::sqrt((double)p->n) * rtol;   // lanczos.h:110
  const double orth_tol = 2.0 * eps * std::sqrt((double)p->n);  // lanczos.h:53
  // alpha and nu[1..] start from zero (the reference's fresh np.zeros buffers, lanczos.py:101-102)
  HIP_TRY(hipMemsetAsync(p->st.alpha, 0, (size_t)(deg + 1) * bp * 8, st));
  HIP_TRY(hipMemsetAsync(p->st.nu + bp, 0, (size_t)deg * bp * 8, st));
  HIP_TRY(hipMemsetAsync(p->st.gram, 0, (size_t)2 * (kFusedMaxR + 1) * bp * 8, st));  // (the Gram rows of a previous run are never read - index guards - but need not be trusted to be)
  if (p->fa_on) HIP_TRY(hipMemsetAsync(p->fa_cnt, 0, ((size_t)p->NP * 8 * p->fa_rounds + 8) * sizeof(int), st));
  int fa_gen = 0;         // fused update + alpha passes launched so far in this run (their counters' generation)
  int fa_alpha_slab = -1; // >= 0: step j's alpha dot was taken by step j - 1's update pass and lies in this slab of `part` (raw: not yet / nu^2)
  const dim3 gA(p->nblkA, p->NP), gS(p->nblkS, p->NP), gU(p->nblkU, p->NP), gF((bp + 63) / 64);
  const dim3 gAf(p->nblkF, p->NP);
  const dim3 gT(p->nblkT, p->NP);
  const slq_operator *op = p->op;
  bool prev_xt = false;  // the previous step's update pass produced the cross term W_{j}.W_{j-1}
  for (int j = 0; j < deg; ++j) {
    const int sc_ = j % S, sp_ = (j + S - 1) % S, sn_ = (j + 1) % S;
    const int first = (j == 0);
    // reorth columns: the last `orth` ring vectors; before step orth-1 only j+1 exist, unless the
    // drop-in entry preloaded the caller's stale ring columns as vectors t < 0 (lanczos_single)
    const int r = p->orth > 0 ? std::min(j + 1 + p->nstale, p->orth) : 0;
    int nblk_last = p->nblkS;
    // exact modified Gram-Schmidt order (one ring column at a time, each dot taken on the updated w):
    // used when stale ring columns take part, whose projections are NOT small, so block-CGS and the
    // reference's MGS would differ at second order (1e-6..1e-5 measured with 18 stale vectors)
    const bool mgs = p->nstale > 0 || p->sw.mgs;
    // Recomputing the SpMM in every pass pays while the gathers are served from cache. A row whose
    // neighbours are scattered over the whole vector (random graph, 16 per row) pays an HBM row fetch per
    // gather and per pass: there the sweeps that gather once and store are 1.3-1.55x faster (configs[2]:
    // 0.59 -> 0.38 s at orth 3), while grids keep the passes even in natural 3-D order (2 far gathers per row:
    // 135 vs 178 ms). SLQ_FUSED=2 forces the passes.
    const bool gathers_cached = fused_mode == 2 || op->far_per_row <= 4.0;
    // ... with r >= 1 the merged pass can store u for the update pass to read back (SLQ_STORED_U, default on):
    // one gather pass per step, 7 reads + 2 writes instead of the sweeps' 9 reads + 3 writes
    const bool stored_u = op->kind == OP_CSR && fused && !gathers_cached && r >= 1 && r <= kFusedMaxR && !mgs &&
                          p->sw.stored_u && p->sw.merged && !plan_tiled(p);
    if (op->kind == OP_CSR && fused && (gathers_cached || stored_u) && r <= kFusedMaxR && !mgs) {
      // ---- fused passes: recompute the SpMM, write once (slq_kernels.hpp: k_csr_pass) ----
      const int V = p->dtype == SLQ_F64 ? 2 : 4;
      const size_t lds0 = sizeof(double) * kWaves * 64 * V;
      // wide panels (one row per wave) of an operator with workgroup tiles (SLQ_TILES): the tile's distinct panel rows are
      // staged once in LDS (k_csr_tile_pass); everything else about the sequence is the same
      // (ring-sized tiles serve up to kRingMaxR ring columns; steps with more take the generic passes, on the same row order)
      const bool tiled = plan_tiled(p) && !stored_u && (!op->tiles_ringed || r <= kRingMaxR || p->ring_deep);
      const bool gen = tiled && op->tiles_ringed && (p->ring_gen || r > kRingMaxR);  // k_ring_pass rather than k_csr_ring_pass
      const size_t lds_tile = tiled ? (size_t)(SLQ_TILE_DB ? 2 : 1) * op->tiles.max_cols * p->PW * p->esz : 0;  // the tile image(s)
      // the alpha-only pass of a symmetric operator stays on the upper triangle (half the gathers) rather than the ring
      // the alpha-only pass of a symmetric operator: ring-fed over the upper-triangle stream where the operator has one
      // (SLQ_RING_ALPHA=2, default; else the generic upper-triangle pass), ring-fed over the full rows (1), generic (0)
      const bool alpha_tiled = tiled && !(op->tiles_ringed && op->rowptr_u != nullptr &&
                                          (p->sw.ring_alpha == 0 || (p->sw.ring_alpha == 2 && p->rs_desc_u == nullptr)));
#define CSR_PASS_RC(PASS, LP, RCT, LDS, XT)                                                          \
  do {                                                                                               \
    if (tl && gen)                                                                                   \
      SLQ_TRY(launch_ring_gen(p, PASS, RCT, gT, st, j, XT));                                         \
    else if (tl)                                                                                     \
      DISPATCH(p->dtype, p->LPR, (launch_tile_pass<F, L, PASS, LP, RCT>(p, gT, lds0 + lds_tile, st, j, XT))); \
    else                                                                                             \
      DISPATCH(p->dtype, p->LPR,                                                                     \
               (launch_csr_pass<F, L, PASS, LP, RCT>(pipe_on, (PASS == PASS_ALPHA ? gAf : gU), LDS, st, p->n, \
                   half ? op->rowptr_u : op->rowptr, half ? op->colind_u : op->colind,               \
                   (const F *)(half ? op->vals_u : op->vals), (F *)p->ring, p->slot_stride, S, j,    \
                   p->st.coefA, p->st.coefB, p->st.gamma, p->part, bp, XT)));                        \
  } while (0)
#define CSR_PASS(PASS, LP, SP, I0, RC, LDS, XT)                                                        \
  do {                                                                                               \
    const bool tl = PASS == PASS_ALPHA ? alpha_tiled : tiled;                                        \
    const bool half = !tl && PASS == PASS_ALPHA && op->rowptr_u != nullptr;                          \
    switch (PASS == PASS_ALPHA ? 0 : (RC)) {                                                         \
      case 0: CSR_PASS_RC(PASS, LP, ((PASS == PASS_DOTS || PASS == PASS_ADOTS) ? 1 : 0), LDS, XT); break; \
      case 1: CSR_PASS_RC(PASS, LP, (PASS == PASS_ALPHA ? 0 : 1), LDS, XT); break;                   \
      case 2: CSR_PASS_RC(PASS, LP, (PASS == PASS_ALPHA ? 0 : 2), LDS, XT); break;                   \
      case 3: CSR_PASS_RC(PASS, LP, (PASS == PASS_ALPHA ? 0 : 3), LDS, XT); break;                   \
      case 4: CSR_PASS_RC(PASS, LP, (PASS == PASS_ALPHA ? 0 : 4), LDS, XT); break;                   \
      case 5: CSR_PASS_RC(PASS, LP, (PASS == PASS_ALPHA ? 0 : 5), LDS, XT); break;                   \
      case 6: CSR_PASS_RC(PASS, LP, (PASS == PASS_ALPHA ? 0 : 6), LDS, XT); break;                   \
      case 7: CSR_PASS_RC(PASS, LP, (PASS == PASS_ALPHA ? 0 : 7), LDS, XT); break;                   \
      default: CSR_PASS_RC(PASS, LP, (PASS == PASS_ALPHA ? 0 : 8), LDS, XT); break;                  \
    }                                                                                                \
  } while (0)
      // alpha pass: grid and residency cap (nblkF, alpha_pad) are chosen in slq_plan_create
      const size_t ldsA = lds0 + (alpha_tiled ? 0 : p->alpha_pad);
      // dots/update passes: 64 KiB of LDS padding pins residency at 2 workgroups per CU whatever the variant's
      // register count (62-96 VGPRs would admit 3 for some). Their grid is 2 per CU *per panel*: blocks are
      // dispatched panel-major, so panel 0 fills the chip, panel 1 follows as its workgroups retire, and an
      // XCD's L2 holds one panel's gather halo at a time (both panels side by side fetch 9.2/10.8 GB per
      // dots/update launch instead of 6.5/8.6 GB).
      // (the pipelined row loop runs ONE resident workgroup per CU: 96 KiB of padding)
      const size_t fused_pad = tiled ? 0 : (size_t)(p->sw.fused_pad >= 0 ? p->sw.fused_pad : (p->pipelined ? 98304 : 65536));
      const bool pipe_on = p->pipelined && !tiled;
      // r >= 1: alpha comes out of the dots pass (PASS_ADOTS: two gather passes per step instead of three)
      const bool merged = r > 0 && (tiled || p->sw.merged);  // (the tiled kernels have the merged form only)
      // cross term: the update pass of the previous step left W_c.W_p behind, so the alpha pass skips W_p
      const int xt_a = (prev_xt && j > 0) ? 1 : 0;
      const int su = stored_u ? 2 : 0;
      const int xt_u = ((!merged && p->sw.cross) ? 1 : 0) | su;
      // Gram sequence (ring-fed plans, r >= 1): alpha-only pass, projections from the Gram rows of the last two update passes
      // (k_fin_gram), update pass that also takes the new vector against every ring column it reads (slq_kernels.hpp)
      const bool gram = gen && p->gram && r >= 1 && p->nstale == 0;
      if (gram) {
        const int xa = j > 0 ? 1 : 0;  // (alpha_j's -beta q_j.q_{j-1} part is a Gram entry: the pass leaves W_p unread)
        const size_t slab = (size_t)p->part_maxblk * bp;  // (the edge kernel's partials live in slab 12 of `part`, away from every pass's own)
        double *part_e = p->part + 12 * slab;
        const int nblk_e = std::max(1, std::min(p->part_maxblk, (op->fa.nedges + kWaves - 1) / kWaves));
        if (fa_alpha_slab < 0) {
          PROFILED(p, SLQ_K_SPMM, { if (nt) CSR_PASS(PASS_ALPHA, 1, 1, 0, 0, ldsA, xa); else CSR_PASS(PASS_ALPHA, 0, 0, 0, 0, ldsA, xa); });
          PROFILED(p, SLQ_K_FINALIZE,
                   hipLaunchKernelGGL(k_fin_gram, dim3((bp + 63) / 64, r), dim3(kFinThreads), 0, st, p->st, p->part, alpha_tiled ? p->nblkT : p->nblkF, j, r, orth_tol,
                                      (const double *)nullptr, 0, 0));
        } else {
          // alpha_j's dot came out of step j - 1's update pass (slq_ring_fa.hpp) and the edge kernel: raw sums, normalised here
          PROFILED(p, SLQ_K_FINALIZE,
                   hipLaunchKernelGGL(k_fin_gram, dim3((bp + 63) / 64, r), dim3(kFinThreads), 0, st, p->st, p->part + (size_t)fa_alpha_slab * p->nblkT * bp, p->nblkT, j, r,
                                      orth_tol, (const double *)part_e, op->fa.nedges > 0 ? nblk_e : 0, 1));
        }
        // the next step's alpha dot rides on this update pass when that step is a Gram step of at most kRingMaxR columns too
        const int r_next = std::min(j + 2 + p->nstale, p->orth);
        const bool fa = p->fa_on && j + 1 < deg && r <= kRingMaxR && r_next <= kRingMaxR;
        if (fa) {
          PROFILED(p, SLQ_K_REORTH_UPD, SLQ_TRY(launch_ring_fa(p, PASS_UPDATEG, r, gT, st, j, ++fa_gen)));
          if (op->fa.nedges > 0)
            PROFILED(p, SLQ_K_SPMM,
                     DISPATCH(p->dtype, 64, (launch_alpha_edges<F, L>(p, nblk_e, st, sn_, part_e))));
          fa_alpha_slab = 1 + r;
        } else {
          PROFILED(p, SLQ_K_REORTH_UPD, SLQ_TRY(launch_ring_gen(p, PASS_UPDATEG, r, gT, st, j, 0)));
          fa_alpha_slab = -1;
        }
        PROFILED(p, SLQ_K_FINALIZE,
                 hipLaunchKernelGGL(k_fin_beta_gram, dim3((bp + 63) / 64, r + 1), dim3(kFinThreads), 0, st, p->st, p->part, p->nblkT, j, r, residual_tol));
        prev_xt = false;
        continue;
      }
      fa_alpha_slab = -1;
      // the same sequence on the generic passes (k_csr_pass<PASS_UPDATEG>; r04): alpha-only pass over the upper triangle, projections from Gram rows
      if (p->gram_csr && !tiled && gathers_cached && !stored_u && r >= 1 && p->nstale == 0) {
        const int xa = j > 0 ? 1 : 0;
        PROFILED(p, SLQ_K_SPMM, { if (nt) CSR_PASS(PASS_ALPHA, 1, 1, 0, 0, ldsA, xa); else CSR_PASS(PASS_ALPHA, 0, 0, 0, 0, ldsA, xa); });
        PROFILED(p, SLQ_K_FINALIZE,
                 hipLaunchKernelGGL(k_fin_gram, dim3((bp + 63) / 64, r), dim3(kFinThreads), 0, st, p->st, p->part, alpha_tiled ? p->nblkT : p->nblkF, j, r, orth_tol,
                                    (const double *)nullptr, 0, 0));
        PROFILED(p, SLQ_K_REORTH_UPD, DISPATCH(p->dtype, p->LPR, (launch_csr_updateg<F, L>(p, r, pipe_on, gU, lds0 + fused_pad, st, j))));
        PROFILED(p, SLQ_K_FINALIZE,
                 hipLaunchKernelGGL(k_fin_beta_gram, dim3((bp + 63) / 64, r + 1), dim3(kFinThreads), 0, st, p->st, p->part, p->nblkU, j, r, residual_tol));
        prev_xt = false;
        continue;
      }
      if (merged) {
        PROFILED(p, SLQ_K_REORTH_DOT, { if (nt) CSR_PASS(PASS_ADOTS, 1, 1, 0, r, lds0 + fused_pad, su); else CSR_PASS(PASS_ADOTS, 0, 0, 0, r, lds0 + fused_pad, su); });
        PROFILED(p, SLQ_K_FINALIZE,
                 hipLaunchKernelGGL(k_fin_adots, dim3((bp + 63) / 64, r), dim3(kFinThreads), 0, st, p->st, p->part, tiled ? p->nblkT : p->nblkU, j, r, orth_tol));
      } else {
      PROFILED(p, SLQ_K_SPMM, { if (nt) CSR_PASS(PASS_ALPHA, 1, 1, 0, 0, ldsA, xt_a); else CSR_PASS(PASS_ALPHA, 0, 0, 0, 0, ldsA, xt_a); });
      PROFILED(p, SLQ_K_FINALIZE,
               hipLaunchKernelGGL(k_fin_alpha, gF, dim3(kFinThreads), 0, st, p->st, p->part, alpha_tiled ? p->nblkT : p->nblkF, j, xt_a));
      if (r > 0) {
        PROFILED(p, SLQ_K_REORTH_DOT, { if (nt) CSR_PASS(PASS_DOTS, 1, 1, 0, r, lds0 + fused_pad, 0); else CSR_PASS(PASS_DOTS, 0, 0, 0, r, lds0 + fused_pad, 0); });
        PROFILED(p, SLQ_K_FINALIZE,
                 hipLaunchKernelGGL(k_fin_gamma, dim3((bp + 63) / 64, r), dim3(kFinThreads), 0, st, p->st,
                                    p->part, tiled ? p->nblkT : p->nblkU, j, 0, orth_tol));
      }
      }
      const size_t ldsU = lds0 + fused_pad;
      // (generic passes: the run's last step does not store W_deg; the ring-fed ones add the bit themselves, launch_ring_gen; the older tiled kernels store)
      const int xt_uu = xt_u | ((!tiled && j == deg - 1 && !p->keep_basis && p->last_nostore && !stored_u) ? 16 : 0);
      PROFILED(p, (r == 0 ? SLQ_K_AXPY_NORM : SLQ_K_REORTH_UPD),
               { if (nt) CSR_PASS(PASS_UPDATE, 1, 1, 0, r, ldsU, xt_uu); else CSR_PASS(PASS_UPDATE, 0, 0, 0, r, ldsU, xt_uu); });
#undef CSR_PASS
#undef CSR_PASS_RC
      nblk_last = tiled ? p->nblkT : p->nblkU;
      prev_xt = (xt_u & 1) != 0;
    } else {
    prev_xt = false;
    if (op->kind == OP_CSR && plan_tiled(p) && op->tiles_ringed) {
      // the sweeps' SpMM + three-term step on the ring-fed tiles (k_csr_ring_pass<PASS_SPMM>): same result slot, same alpha partials
      PROFILED(p, SLQ_K_SPMM, {
        if (p->ring_gen) SLQ_TRY(launch_ring_gen(p, PASS_SPMM, 0, gT, st, j, 0));
        else if (nt) DISPATCH(p->dtype, p->LPR, (launch_tile_pass<F, L, PASS_SPMM, 1, 0>(p, gT, 0, st, j, 0)));
        else DISPATCH(p->dtype, p->LPR, (launch_tile_pass<F, L, PASS_SPMM, 0, 0>(p, gT, 0, st, j, 0)));
      });
      PROFILED(p, SLQ_K_FINALIZE,
               hipLaunchKernelGGL(k_fin_alpha, gF, dim3(kFinThreads), 0, st, p->st, p->part, p->nblkT, j, 0));
    } else if (op->kind == OP_CSR) {
      const int pol = nt ? 11 : 0;  // tens digit: load policy, units: store policy
      const size_t spmm_pad = (size_t)p->sw.spmm_pad;  // dynamic LDS only to cap residency at 2 per CU (panel after panel: 38.8 -> 34.7 ms per 26 launches at orth 30)
#define SPMM_LAUNCH(LP, SP)                                                                          \
  DISPATCH(p->dtype, p->LPR,                                                                         \
           (k_spmm_3term<F, L, LP, SP><<<gA, dim3(kBlock), spmm_pad, st>>>(                          \
               p->n, op->rowptr, op->colind, (const F *)op->vals, (const F *)slot_ptr(p, sc_),       \
               (const F *)slot_ptr(p, sp_), (F *)slot_ptr(p, sn_), p->st.coefA, p->part, bp, first)))
      PROFILED(p, SLQ_K_SPMM, {
        switch (pol) {
          case 1: SPMM_LAUNCH(0, 1); break;
          case 2: SPMM_LAUNCH(0, 2); break;
          case 10: SPMM_LAUNCH(1, 0); break;
          case 11: SPMM_LAUNCH(1, 1); break;
          case 12: SPMM_LAUNCH(1, 2); break;
          default: SPMM_LAUNCH(0, 0); break;
        }
      });
#undef SPMM_LAUNCH
      PROFILED(p, SLQ_K_FINALIZE,
               hipLaunchKernelGGL(k_fin_alpha, gF, dim3(kFinThreads), 0, st, p->st, p->part, p->nblkA, j, 0));
    } else if (op->kind == OP_DENSE && p->sw.dense_mfma && (p->dtype == SLQ_F64 || p->dense_ks > 0)) {
      int nb = 0;
      PROFILED(p, SLQ_K_SPMM, SLQ_TRY(launch_dense_mfma(p, slot_ptr(p, sc_), slot_ptr(p, sp_), slot_ptr(p, sn_), first, 0, &nb)));
      PROFILED(p, SLQ_K_FINALIZE,
               hipLaunchKernelGGL(k_fin_alpha, gF, dim3(kFinThreads), 0, st, p->st, p->part, nb, j, 0));
    } else {
      SLQ_TRY(apply_operator_unfused(p, sc_));
      PROFILED(p, SLQ_K_AXPY_NORM,
               DISPATCH(p->dtype, p->LPR,
                        (k_3term<F, L><<<gS, dim3(kBlock), 0, st>>>(p->n,
                                            (const F *)p->T, (const F *)slot_ptr(p, sc_),
                                            (const F *)slot_ptr(p, sp_), (F *)slot_ptr(p, sn_),
                                            p->st.coefA, p->part, bp, first))));
      PROFILED(p, SLQ_K_FINALIZE,
               hipLaunchKernelGGL(k_fin_alpha, gF, dim3(kFinThreads), 0, st, p->st, p->part, p->nblkS, j, 0));
    }
    if (r == 0) {
      PROFILED(p, SLQ_K_AXPY_NORM,
               DISPATCH(p->dtype, p->LPR,
                        (k_axpy_norm<F, L, 0><<<gS, dim3(kBlock), 0, st>>>(p->n,
                                            (F *)slot_ptr(p, sn_), (const F *)slot_ptr(p, sc_),
                                            p->st.coefB, p->part, bp))));
    } else if (mgs) {
      for (int i = 0; i < r; ++i) {
        PROFILED(p, SLQ_K_REORTH_DOT,
                 DISPATCH(p->dtype, p->LPR,
                          (k_reorth_dot<F, L><<<gS, dim3(kBlock), 0, st>>>(p->n,
                                              (F *)p->ring, p->slot_stride, S, j, i, 1, (int)(i == 0),
                                              p->st.coefB, p->part, bp))));
        PROFILED(p, SLQ_K_FINALIZE,
                 hipLaunchKernelGGL(k_fin_gamma, dim3((bp + 63) / 64, 1), dim3(kFinThreads), 0, st, p->st,
                                    p->part, p->nblkS, j, i, orth_tol));
        SLQ_TRY(launch_reorth_update_range(p, j, i, i + 1));
      }
    } else {
      const int dot_chunk = p->ring32_on ? kReorthChunk32 : kReorthChunk;  // reorth columns per dots launch
      for (int i0 = 0; i0 < r; i0 += dot_chunk) {
        const int rc = std::min(dot_chunk, r - i0);
        PROFILED(p, SLQ_K_REORTH_DOT,
                 DISPATCH(p->dtype, p->LPR,
                          (launch_reorth_dot<F, L>(p, gS, st, j, i0, rc))));
        PROFILED(p, SLQ_K_FINALIZE,
                 hipLaunchKernelGGL(k_fin_gamma, dim3((bp + 63) / 64, rc), dim3(kFinThreads), 0, st, p->st,
                                    p->part, p->nblkS, j, i0, orth_tol));
      }
      SLQ_TRY(launch_reorth_update(p, j, r, 0, p->sw.defer_axpy && !p->ring32_on));
    }
    }  // !fused
    PROFILED(p, SLQ_K_FINALIZE,
             hipLaunchKernelGGL(k_fin_beta, gF, dim3(kFinThreads), 0, st, p->st, p->part, nblk_last, j, residual_tol, prev_xt ? 1 : 0));
  }
  HIP_TRY(hipGetLastError());
  if (p->launch_error) {
    p->launch_error = false;
    return fail(SLQ_EINVAL, "internal: a pass of the launch sequence had no kernel for this plan's tiles");
  }
  return SLQ_OK;
}

extern "C" int slq_plan_run(slq_plan *p, double rtol) {
  if (!p) return fail(SLQ_EINVAL, "plan is NULL");
  if (!p->probes_ready) return fail(SLQ_EINVAL, "slq_plan_run: set or generate probes first");
  HIP_TRY(hipSetDevice(p->ctx->device));
  hipStream_t st = p->ctx->stream;
  const int fused = p->sw.fused;
  const bool nt = p->sw.nt != 0;
  // The launch sequence of a run (7 launches per Lanczos step, ~210 for k = 30) depends only on
  // the plan, so it is captured into a hipGraph once and replayed: launch-bound for small n,
  // a few per cent for n = 1e6. Not used while per-kernel events are recorded, nor for host-callback
  // operators (they synchronise with the host every step).
  const bool graph_ok = p->sw.graph && !p->prof && p->op->kind != OP_CALLBACK && p->op->kind != OP_DEVICE_CALLBACK;
  if (!graph_ok) {
    SLQ_TRY(enqueue_run(p, rtol, fused, nt));
  } else {
    const unsigned variant = p->sw.key() * 31u + (unsigned)p->nstale;  // every switch of the sequence + the stale-column count
    if (!p->graph_exec || p->graph_rtol != rtol || p->graph_variant != variant) {
      if (p->graph_exec) {
        HIP_TRY(hipGraphExecDestroy(p->graph_exec));
        p->graph_exec = nullptr;
      }
      hipGraph_t graph = nullptr;
      HIP_TRY(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
      const int rc = enqueue_run(p, rtol, fused, nt);
      hipError_t ce = hipStreamEndCapture(st, &graph);
      if (rc != SLQ_OK) {
        if (graph) hipGraphDestroy(graph);
        return rc;
      }
      if (ce != hipSuccess) return fail(SLQ_EHIP, "hipStreamEndCapture: %s", hipGetErrorString(ce));
      ce = hipGraphInstantiate(&p->graph_exec, graph, nullptr, nullptr, 0);
      hipGraphDestroy(graph);
      if (ce != hipSuccess) {
        p->graph_exec = nullptr;
        return fail(SLQ_EHIP, "hipGraphInstantiate: %s", hipGetErrorString(ce));
      }
      p->graph_rtol = rtol;
      p->graph_variant = variant;
    }
    HIP_TRY(hipGraphLaunch(p->graph_exec, st));
  }
  p->probes_ready = false;
  p->ran = true;
  return SLQ_OK;
}

extern "C" int slq_plan_get_tridiag(slq_plan *p, void *alpha, void *beta, int32_t *steps) {
  if (!p) return fail(SLQ_EINVAL, "plan is NULL");
  if (!p->ran) return fail(SLQ_EINVAL, "slq_plan_get_tridiag: no completed run");
  HIP_TRY(hipSetDevice(p->ctx->device));
  hipStream_t st = p->ctx->stream;
  const int bp = p->bpad, deg = p->deg, P = p->nprobes;
  std::vector<double> ha((size_t)(deg + 1) * bp), hn((size_t)(deg + 1) * bp);
  std::vector<int> hs(bp);
  HIP_TRY(hipMemcpyAsync(ha.data(), p->st.alpha, ha.size() * 8, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(hn.data(), p->st.nu, hn.size() * 8, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(hs.data(), p->st.steps, (size_t)bp * 4, hipMemcpyDeviceToHost, st));
  int ring_bad = 0;
  HIP_TRY(hipMemcpyAsync(&ring_bad, p->ring_fail_d, sizeof(int), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  SLQ_TRY(ring_flag_status(ring_bad));
  for (int i = 0; i < P; ++i) {
    for (int t = 0; t <= deg; ++t) {
      const double a = ha[(size_t)t * bp + i];
      const double b = (t == 0) ? 0.0 : hn[(size_t)t * bp + i];  // beta[0] = 0 (lanczos.h:121)
      if (p->dtype == SLQ_F64) {
        if (alpha) ((double *)alpha)[(size_t)i * (deg + 1) + t] = a;
        if (beta) ((double *)beta)[(size_t)i * (deg + 1) + t] = b;
      } else {
        if (alpha) ((float *)alpha)[(size_t)i * (deg + 1) + t] = (float)a;
        if (beta) ((float *)beta)[(size_t)i * (deg + 1) + t] = (float)b;
      }
    }
    if (steps) steps[i] = hs[i];
  }
  return SLQ_OK;
}

extern "C" int slq_plan_quadrature(slq_plan *p, int fun_id, const double *fun_params, double *quad,
                                   double *nodes, double *weights) {
  if (!p) return fail(SLQ_EINVAL, "plan is NULL");
  if (!p->ran) return fail(SLQ_EINVAL, "slq_plan_quadrature: no completed run");
  if (fun_id < SLQ_FUN_NONE || fun_id > SLQ_FUN_SOFTSIGN) return fail(SLQ_EINVAL, "Unknown function id %d.", fun_id);
  HIP_TRY(hipSetDevice(p->ctx->device));
  hipStream_t st = p->ctx->stream;
  const int deg = p->deg, P = p->nprobes;
  const double p0 = fun_params ? fun_params[0] : 0.0, p1 = fun_params ? fun_params[1] : 0.0;
  const int lanes = quadrature_lanes(deg);
  const size_t lds = (size_t)3 * deg * lanes * 8;
  if (lds > 48 * 1024)
    HIP_TRY(hipFuncSetAttribute((const void *)k_quadrature, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIP_TRY(hipMemsetAsync(p->fail_d, 0, sizeof(int), st));
  PROFILED(p, SLQ_K_QUADRATURE,
           hipLaunchKernelGGL(k_quadrature, dim3((P + lanes - 1) / lanes), dim3(64), lds, st, p->st, lanes, fun_id, p0, p1,
                              p->quad_d, (nodes ? p->nodes_d : nullptr), (weights ? p->weights_d : nullptr), p->fail_d));
  HIP_TRY(hipGetLastError());
  int bad2[2] = {0, 0};  // fail_d, ring_fail_d
  HIP_TRY(hipMemcpyAsync(bad2, p->fail_d, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
  if (quad) HIP_TRY(hipMemcpyAsync(quad, p->quad_d, (size_t)P * 8, hipMemcpyDeviceToHost, st));
  if (nodes) HIP_TRY(hipMemcpyAsync(nodes, p->nodes_d, (size_t)P * deg * 8, hipMemcpyDeviceToHost, st));
  if (weights) HIP_TRY(hipMemcpyAsync(weights, p->weights_d, (size_t)P * deg * 8, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  const int bad = bad2[0];
  SLQ_TRY(ring_flag_status(bad2[1]));
  if (bad) return fail(SLQ_ENOTCONV, "tridiagonal QL did not converge for at least one probe");
  return SLQ_OK;
}

extern "C" int slq_plan_get_basis(slq_plan *p, int probe, void *Q, int64_t ldq) {
  if (!p || !Q) return fail(SLQ_EINVAL, "plan/Q is NULL");
  if (!p->keep_basis) return fail(SLQ_EINVAL, "plan was created without keep_basis");
  if (!p->ran) return fail(SLQ_EINVAL, "no completed run");
  if (probe < 0 || probe >= p->nprobes || ldq < p->n) return fail(SLQ_EINVAL, "bad probe index or ldq");
  HIP_TRY(hipSetDevice(p->ctx->device));
  hipStream_t st = p->ctx->stream;
  const int bp = p->bpad, deg = p->deg;
  std::vector<double> hn((size_t)(deg + 1) * bp), hscale((size_t)deg * bp, 0.0);
  HIP_TRY(hipMemcpyAsync(hn.data(), p->st.nu, hn.size() * 8, hipMemcpyDeviceToHost, st));
  SLQ_TRY(check_ring_flag(p));
  for (int t = 0; t < deg; ++t) {
    const double nu = hn[(size_t)t * bp + probe];
    hscale[(size_t)t * bp + probe] = nu > 0.0 ? 1.0 / nu : 0.0;
  }
  double *dscale = nullptr;
  HIP_TRY(hipMalloc((void **)&dscale, hscale.size() * 8));
  hipError_t e = hipMemcpyAsync(dscale, hscale.data(), hscale.size() * 8, hipMemcpyHostToDevice, st);
  int rc = e == hipSuccess ? SLQ_OK : fail(SLQ_EHIP, "scale upload: %s", hipGetErrorString(e));
  for (int t = 0; t < deg && rc == SLQ_OK; ++t)
    rc = panel_to_host(p, t, probe, 1, (char *)Q + (size_t)t * (size_t)ldq * p->esz, ldq, dscale + (size_t)t * bp);
  hipFree(dscale);
  return rc;
}

static int update_chunk_cols(const slq_plan *p) {
  const int V = p->dtype == SLQ_F64 ? 2 : 4;
  return std::min(192, (int)((150 * 1024 - sizeof(double) * kWaves * 64 * V) / ((size_t)p->PW * p->esz + sizeof(int))));  // (192: the update sweep's column masks)
}

// w(slot (j+1)%S) -= sum_{i<r} gamma[i] * W_{j-i}, gamma staged through LDS in chunks
// axpy: the first chunk also applies the three-term step's `w -= cB W_c` (the dots sweeps ran in mode 2 and stored nothing)
static int launch_reorth_update_range(slq_plan *p, int j, int istart, int r, bool axpy, int klass) {
  if (axpy && (istart != 0 || r < 1)) return fail(SLQ_EINVAL, "internal: the deferred axpy rides on column 0 of the first update chunk");
  hipStream_t st = p->ctx->stream;
  const int V = p->dtype == SLQ_F64 ? 2 : 4;
  const int kUpdChunk = update_chunk_cols(p);
  const dim3 gS(p->nblkS, p->NP);
  for (int i0 = istart; i0 < r; i0 += kUpdChunk) {
    const int rc = std::min(kUpdChunk, r - i0);
    const size_t lds = sizeof(double) * kWaves * 64 * V + (size_t)rc * p->PW * p->esz + (size_t)rc * sizeof(int);  // reduction scratch, gamma, per-column flags
    PROFILED(p, klass,
             DISPATCH(p->dtype, p->LPR,
                      (launch_reorth_update_kernel<F, L>(p, gS, lds, st, j, i0, rc, (int)(i0 + rc >= r), axpy && i0 == 0))));  // last chunk: w is final; (axpy: column 0 of the chunk must be W_c)
  }
  return SLQ_OK;
}

static int launch_reorth_update(slq_plan *p, int j, int r, int istart, bool axpy, int klass) { return launch_reorth_update_range(p, j, istart, r, axpy, klass); }

// Y = f(A) X on the device: result left in ring slot `deg` (panel layout)
static int fun_action_device(slq_plan *p, int fun_id, const double *fun_params) {
  if (!p->keep_basis) return fail(SLQ_EINVAL, "plan was created without keep_basis");
  if (!p->ran) return fail(SLQ_EINVAL, "no completed run");
  if (fun_id < SLQ_FUN_IDENTITY || fun_id > SLQ_FUN_SOFTSIGN) return fail(SLQ_EINVAL, "Unknown function id %d.", fun_id);
  HIP_TRY(hipSetDevice(p->ctx->device));
  hipStream_t st = p->ctx->stream;
  const int deg = p->deg;
  const double p0 = fun_params ? fun_params[0] : 0.0, p1 = fun_params ? fun_params[1] : 0.0;
  // eigenvectors of every probe's T: in LDS up to deg = 141, beyond that in a global scratch (stays in L2)
  const size_t lds_onchip = ((size_t)2 * deg + (size_t)deg * (deg + 1)) * 8;
  const bool zg = lds_onchip > 160 * 1024;
  const size_t lds = zg ? (size_t)2 * deg * 8 : lds_onchip;
  double *zscr = nullptr;
  if (zg) HIP_TRY(hipMalloc((void **)&zscr, (size_t)p->nprobes * deg * (deg + 1) * 8));
  struct ScratchGuard { double *q; ~ScratchGuard() { if (q) hipFree(q); } } guard{zscr};
  if (lds > 48 * 1024)
    HIP_TRY(hipFuncSetAttribute((const void *)k_fun_coeffs<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIP_TRY(hipMemsetAsync(p->fail_d, 0, sizeof(int), st));
  // gamma row deg-1-t = -g_t: the update kernel's "w -= gamma W" then accumulates +g_t W_t into a
  // zeroed slot while walking t = deg-1 .. 0
  HIP_TRY(hipMemsetAsync(p->st.gamma, 0, (size_t)deg * p->bpad * 8, st));
  if (zg) {
    PROFILED(p, SLQ_K_QUADRATURE,
             (k_fun_coeffs<true><<<dim3(p->nprobes), dim3(64), lds, st>>>(p->st, fun_id, p0, p1, -1.0, 1, p->st.gamma, zscr, p->fail_d)));
  } else {
    PROFILED(p, SLQ_K_QUADRATURE,
             (k_fun_coeffs<false><<<dim3(p->nprobes), dim3(64), lds, st>>>(p->st, fun_id, p0, p1, -1.0, 1, p->st.gamma, nullptr, p->fail_d)));
  }
  // output accumulates in slot `deg` (the spare slot behind the basis; it held the last residual)
  HIP_TRY(hipMemsetAsync(slot_ptr(p, deg), 0, (size_t)p->slot_stride * p->esz, st));
  SLQ_TRY(launch_reorth_update(p, deg - 1, deg, 0, false, SLQ_K_COMBINE));  // (its own profile class: these bytes are not the recurrence's update sweep)
  HIP_TRY(hipGetLastError());
  int bad = 0;
  HIP_TRY(hipMemcpyAsync(&bad, p->fail_d, sizeof(int), hipMemcpyDeviceToHost, st));
  SLQ_TRY(check_ring_flag(p));  // (synchronises)
  if (bad) return fail(SLQ_ENOTCONV, "tridiagonal QL did not converge for at least one probe");
  return SLQ_OK;
}

extern "C" int slq_plan_fun_action(slq_plan *p, int fun_id, const double *fun_params, void *Y, int64_t ldy) {
  if (!p || !Y) return fail(SLQ_EINVAL, "plan/Y is NULL");
  if (ldy < p->n) return fail(SLQ_EINVAL, "ldy < n");
  SLQ_TRY(fun_action_device(p, fun_id, fun_params));
  return panel_to_host(p, p->deg, 0, p->nprobes, Y, ldy, nullptr);
}

// ---- diagonal estimator state (device-resident) ----------------------------------------------------
struct slq_diag {
  slq_context *ctx;
  int64_t n, count;
  double *buf;  // numer | denom | msum, n doubles each (in the operator's stored row order)
  const slq_operator *op;
};

extern "C" int slq_diag_create(slq_context *ctx, int64_t n, slq_diag **out) {
  if (!ctx || !out || n <= 0) return fail(SLQ_EINVAL, "bad arguments");
  *out = nullptr;
  HIP_TRY(hipSetDevice(ctx->device));
  slq_diag *d = new (std::nothrow) slq_diag();
  if (!d) return fail(SLQ_ENOMEM, "host allocation failed");
  d->ctx = ctx; d->n = n; d->count = 0; d->buf = nullptr; d->op = nullptr;
  hipError_t e = hipMalloc((void **)&d->buf, (size_t)3 * n * 8);
  if (e == hipSuccess) e = hipMemsetAsync(d->buf, 0, (size_t)3 * n * 8, ctx->stream);
  if (e != hipSuccess) { delete d; /* not yet retained */ return fail(SLQ_ENOMEM, "diag accumulators: %s", hipGetErrorString(e)); }
  ctx_retain(ctx);
  *out = d;
  return SLQ_OK;
}

extern "C" int slq_diag_destroy(slq_diag *d) {
  if (!d) return SLQ_OK;
  hipSetDevice(d->ctx->device);
  if (d->buf) hipFree(d->buf);
  ctx_release(d->ctx);
  delete d;
  return SLQ_OK;
}

extern "C" int slq_diag_update(slq_diag *d, slq_plan *p, int fun_id, const double *fun_params) {
  if (!d || !p) return fail(SLQ_EINVAL, "diag/plan is NULL");
  if (d->n != p->n || d->ctx != p->ctx) return fail(SLQ_EINVAL, "diag accumulator does not match the plan");
  if (d->op && d->op != p->op) return fail(SLQ_EINVAL, "diag accumulator was started with another operator");
  d->op = p->op;
  SLQ_TRY(fun_action_device(p, fun_id, fun_params));
  hipStream_t st = p->ctx->stream;
  // coefB is free after a run: reuse it for the per-probe scale of the stored probes
  k_probe_scale<<<dim3((p->bpad + 255) / 256), dim3(256), 0, st>>>(p->st, p->st.coefB);
  const dim3 g((p->n + 63) / 64);
  if (p->dtype == SLQ_F64)
    k_diag_accumulate<double><<<g, dim3(64), 0, st>>>(p->n, (const double *)slot_ptr(p, 0), (const double *)slot_ptr(p, p->deg),
                                                      p->PW, p->nprobes, p->st.coefB, d->buf, d->buf + d->n, d->buf + 2 * d->n);
  else
    k_diag_accumulate<float><<<g, dim3(64), 0, st>>>(p->n, (const float *)slot_ptr(p, 0), (const float *)slot_ptr(p, p->deg),
                                                     p->PW, p->nprobes, p->st.coefB, d->buf, d->buf + d->n, d->buf + 2 * d->n);
  HIP_TRY(hipGetLastError());
  d->count += p->nprobes;
  return SLQ_OK;
}

extern "C" int slq_diag_get(slq_diag *d, double *numer, double *denom, double *running_mean, int64_t *count) {
  if (!d) return fail(SLQ_EINVAL, "diag is NULL");
  HIP_TRY(hipSetDevice(d->ctx->device));
  hipStream_t st = d->ctx->stream;
  if (numer) HIP_TRY(hipMemcpyAsync(numer, d->buf, (size_t)d->n * 8, hipMemcpyDeviceToHost, st));
  if (denom) HIP_TRY(hipMemcpyAsync(denom, d->buf + d->n, (size_t)d->n * 8, hipMemcpyDeviceToHost, st));
  if (running_mean) HIP_TRY(hipMemcpyAsync(running_mean, d->buf + 2 * d->n, (size_t)d->n * 8, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  if (running_mean && d->count > 0)
    for (int64_t i = 0; i < d->n; ++i) running_mean[i] /= (double)d->count;
  if (d->op && d->op->perm_h) {  // stored row i is caller row perm[i]
    const std::vector<int32_t> &perm = *d->op->perm_h;
    std::vector<double> tmp((size_t)d->n);
    for (double *arr : {numer, denom, running_mean}) {
      if (!arr) continue;
      for (int64_t i = 0; i < d->n; ++i) tmp[(size_t)perm[(size_t)i]] = arr[i];
      memcpy(arr, tmp.data(), (size_t)d->n * 8);
    }
  }
  if (count) *count = d->count;
  return SLQ_OK;
}


// ---------------------------------------------------------------------------------------------------
// tall-skinny device matrices (xtrace / hutch++ dense algebra on the matrix cores)
// ---------------------------------------------------------------------------------------------------
struct slq_dmat {
  slq_context *ctx;
  int64_t n;
  int cols;
  double *d;  // column-major, ld = n
};

extern "C" int slq_dmat_create(slq_context *ctx, int64_t n, int cols, slq_dmat **out) {
  if (!ctx || !out || n <= 0 || cols <= 0) return fail(SLQ_EINVAL, "bad arguments");
  *out = nullptr;
  HIP_TRY(hipSetDevice(ctx->device));
  slq_dmat *m = new (std::nothrow) slq_dmat();
  if (!m) return fail(SLQ_ENOMEM, "host allocation failed");
  m->ctx = ctx; m->n = n; m->cols = cols; m->d = nullptr;
  hipError_t e = hipMalloc((void **)&m->d, (size_t)n * cols * 8);
  if (e == hipSuccess) e = hipMemsetAsync(m->d, 0, (size_t)n * cols * 8, ctx->stream);
  if (e != hipSuccess) { delete m; return fail(e == hipErrorOutOfMemory ? SLQ_ENOMEM : SLQ_EHIP, "dmat: %s", hipGetErrorString(e)); }
  ctx_retain(ctx);
  *out = m;
  return SLQ_OK;
}

extern "C" int slq_dmat_destroy(slq_dmat *m) {
  if (!m) return SLQ_OK;
  hipSetDevice(m->ctx->device);
  hipStreamSynchronize(m->ctx->stream);
  if (m->d) hipFree(m->d);
  ctx_release(m->ctx);
  delete m;
  return SLQ_OK;
}

static int dmat_range(const slq_dmat *m, int c0, int nc, const char *what) {
  if (!m) return fail(SLQ_EINVAL, "%s: matrix is NULL", what);
  if (c0 < 0 || nc <= 0 || c0 + nc > m->cols) return fail(SLQ_EINVAL, "%s: columns [%d, %d) outside [0, %d)", what, c0, c0 + nc, m->cols);
  return SLQ_OK;
}

extern "C" int slq_dmat_set(slq_dmat *m, int c0, int nc, const double *host, int64_t ld) {
  SLQ_TRY(dmat_range(m, c0, nc, "slq_dmat_set"));
  if (!host || ld < m->n) return fail(SLQ_EINVAL, "bad host array");
  HIP_TRY(hipSetDevice(m->ctx->device));
  HIP_TRY(hipMemcpy2DAsync(m->d + (size_t)c0 * m->n, (size_t)m->n * 8, host, (size_t)ld * 8, (size_t)m->n * 8, (size_t)nc,
                           hipMemcpyHostToDevice, m->ctx->stream));
  HIP_TRY(hipStreamSynchronize(m->ctx->stream));
  return SLQ_OK;
}

extern "C" int slq_dmat_get(slq_dmat *m, int c0, int nc, double *host, int64_t ld) {
  SLQ_TRY(dmat_range(m, c0, nc, "slq_dmat_get"));
  if (!host || ld < m->n) return fail(SLQ_EINVAL, "bad host array");
  HIP_TRY(hipSetDevice(m->ctx->device));
  HIP_TRY(hipMemcpy2DAsync(host, (size_t)ld * 8, m->d + (size_t)c0 * m->n, (size_t)m->n * 8, (size_t)m->n * 8, (size_t)nc,
                           hipMemcpyDeviceToHost, m->ctx->stream));
  HIP_TRY(hipStreamSynchronize(m->ctx->stream));
  return SLQ_OK;
}

extern "C" int slq_dmat_ptr(slq_dmat *m, int c0, void **dptr) {
  SLQ_TRY(dmat_range(m, c0, 1, "slq_dmat_ptr"));
  if (!dptr) return fail(SLQ_EINVAL, "dptr is NULL");
  *dptr = m->d + (size_t)c0 * m->n;
  return SLQ_OK;
}

extern "C" int slq_dmat_generate(slq_dmat *m, int c0, int nc, int pdf, uint64_t seed, uint64_t probe_offset) {
  SLQ_TRY(dmat_range(m, c0, nc, "slq_dmat_generate"));
  if (pdf < 0 || pdf > 2) return fail(SLQ_EINVAL, "Invalid distribution id %d supplied.", pdf);
  HIP_TRY(hipSetDevice(m->ctx->device));
  hipStream_t st = m->ctx->stream;
  double *x = m->d + (size_t)c0 * m->n;
  k_gen_cols<<<dim3((unsigned)((m->n + 255) / 256)), dim3(256), 0, st>>>(m->n, x, nc, pdf == SLQ_PDF_RADEMACHER ? 0 : 1, seed, probe_offset);
  if (pdf == SLQ_PDF_SPHERE) k_scale_cols_sphere<<<dim3(nc), dim3(256), 0, st>>>(m->n, x);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(st));
  return SLQ_OK;
}

extern "C" int slq_dmat_copy(slq_dmat *dst, int d0, slq_dmat *src, int s0, int nc) {
  SLQ_TRY(dmat_range(dst, d0, nc, "slq_dmat_copy(dst)"));
  SLQ_TRY(dmat_range(src, s0, nc, "slq_dmat_copy(src)"));
  if (dst->n != src->n || dst->ctx != src->ctx) return fail(SLQ_EINVAL, "mismatched operands");
  HIP_TRY(hipSetDevice(dst->ctx->device));
  HIP_TRY(hipMemcpyAsync(dst->d + (size_t)d0 * dst->n, src->d + (size_t)s0 * src->n, (size_t)nc * dst->n * 8,
                         hipMemcpyDeviceToDevice, dst->ctx->stream));
  HIP_TRY(hipStreamSynchronize(dst->ctx->stream));
  return SLQ_OK;
}

// dst[dr0 : dr0 + nrows, d0 : d0 + nc] = src[sr0 : sr0 + nrows, s0 : s0 + nc]: a block of rows of some columns, between matrices
// of different heights - what a row-sharded sketch (primate_amd.distributed) takes out of / puts into full columns.
extern "C" int slq_dmat_copy_rows(slq_dmat *dst, int d0, int64_t dr0, slq_dmat *src, int s0, int64_t sr0, int64_t nrows, int nc) {
  SLQ_TRY(dmat_range(dst, d0, nc, "slq_dmat_copy_rows(dst)"));
  SLQ_TRY(dmat_range(src, s0, nc, "slq_dmat_copy_rows(src)"));
  if (dst->ctx != src->ctx) return fail(SLQ_EINVAL, "mismatched operands");
  if (nrows == 0) return SLQ_OK;  // (an empty shard - more ranks than rows - copies nothing, wherever it nominally starts)
  if (nrows < 0 || dr0 < 0 || sr0 < 0 || dr0 + nrows > dst->n || sr0 + nrows > src->n) return fail(SLQ_EINVAL, "row range outside the matrix");
  HIP_TRY(hipSetDevice(dst->ctx->device));
  HIP_TRY(hipMemcpy2DAsync(dst->d + (size_t)d0 * dst->n + dr0, (size_t)dst->n * 8, src->d + (size_t)s0 * src->n + sr0, (size_t)src->n * 8,
                           (size_t)nrows * 8, (size_t)nc, hipMemcpyDeviceToDevice, dst->ctx->stream));
  HIP_TRY(hipStreamSynchronize(dst->ctx->stream));
  return SLQ_OK;
}

extern "C" int slq_dmat_gemm_tn(slq_dmat *A, int a0, int ma, slq_dmat *B, int b0, int mb, double *C_host) {
  SLQ_TRY(dmat_range(A, a0, ma, "slq_dmat_gemm_tn(A)"));
  SLQ_TRY(dmat_range(B, b0, mb, "slq_dmat_gemm_tn(B)"));
  if (!C_host || A->n != B->n || A->ctx != B->ctx) return fail(SLQ_EINVAL, "mismatched operands");
  slq_context *ctx = A->ctx;
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const int n = (int)A->n;
  const int tiles = ((ma + 16 * kTnA - 1) / (16 * kTnA)) * ((mb + 16 * kTnB - 1) / (16 * kTnB));
  // enough single-wave workgroups to fill the chip: ~16 per CU
  int nslab = std::max(1, std::min((n + 1023) / 1024, (ctx->num_cus * 16 + tiles - 1) / tiles));
  int slab_rows = ((n + nslab - 1) / nslab + 15) / 16 * 16;
  nslab = (n + slab_rows - 1) / slab_rows;
  const size_t cnt = (size_t)ma * mb;
  double *buf = nullptr;
  HIP_TRY(hipMalloc((void **)&buf, ((size_t)nslab + 1) * cnt * 8));
  k_gemm_tn<<<dim3(tiles, nslab), dim3(64), 0, st>>>(n, A->d + (size_t)a0 * n, (int64_t)n, B->d + (size_t)b0 * n, (int64_t)n, ma, mb,
                                                    slab_rows, buf);
  k_sum_slabs<<<dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st>>>(buf, nslab, (int64_t)cnt, buf + (size_t)nslab * cnt);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(C_host, buf + (size_t)nslab * cnt, cnt * 8, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  hipFree(buf);
  if (e != hipSuccess) return fail(SLQ_EHIP, "slq_dmat_gemm_tn: %s", hipGetErrorString(e));
  return SLQ_OK;
}

extern "C" int slq_dmat_gemm_nn(slq_dmat *OUT, int o0, slq_dmat *A, int a0, int ma, const double *C_host, int mb,
                                double alpha, double beta) {
  SLQ_TRY(dmat_range(OUT, o0, mb, "slq_dmat_gemm_nn(OUT)"));
  SLQ_TRY(dmat_range(A, a0, ma, "slq_dmat_gemm_nn(A)"));
  if (!C_host || A->n != OUT->n || A->ctx != OUT->ctx) return fail(SLQ_EINVAL, "mismatched operands");
  if (A == OUT && !(o0 + mb <= a0 || a0 + ma <= o0)) return fail(SLQ_EINVAL, "output columns overlap the input columns");
  slq_context *ctx = A->ctx;
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const int n = (int)A->n;
  double *dC = nullptr;
  HIP_TRY(hipMalloc((void **)&dC, (size_t)ma * mb * 8));
  hipError_t e = hipMemcpyAsync(dC, C_host, (size_t)ma * mb * 8, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) {
    k_gemm_nn<<<dim3((n + 15) / 16, (mb + 16 * kNnB - 1) / (16 * kNnB)), dim3(64), 0, st>>>(
        n, OUT->d + (size_t)o0 * n, (int64_t)n, A->d + (size_t)a0 * n, (int64_t)n, ma, dC, mb, alpha, beta);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  hipFree(dC);
  if (e != hipSuccess) return fail(SLQ_EHIP, "slq_dmat_gemm_nn: %s", hipGetErrorString(e));
  return SLQ_OK;
}

// f(A) X for the probes of a completed keep_basis run, written to OUT[:, o0 : o0 + nprobes] (fp64 plans)
extern "C" int slq_plan_fun_action_dmat(slq_plan *p, int fun_id, const double *fun_params, slq_dmat *OUT, int o0) {
  if (!p) return fail(SLQ_EINVAL, "plan is NULL");
  SLQ_TRY(dmat_range(OUT, o0, p->nprobes, "slq_plan_fun_action_dmat"));
  if (p->dtype != SLQ_F64 || OUT->n != p->n || OUT->ctx != p->ctx) return fail(SLQ_EINVAL, "plan and matrix do not match (fp64, same n, same context)");
  SLQ_TRY(fun_action_device(p, fun_id, fun_params));
  hipStream_t st = p->ctx->stream;
  dim3 g((p->n + 63) / 64, (p->nprobes + 63) / 64);
  hipLaunchKernelGGL(k_panel_to_cols<double>, g, dim3(256), 0, st, p->n, (const double *)slot_ptr(p, p->deg), 0, p->nprobes,
                     OUT->d + (size_t)o0 * p->n, p->PW, (const double *)nullptr, p->op->perm_d);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(st));
  return SLQ_OK;
}

extern "C" int slq_plan_get_probes_dmat(slq_plan *p, slq_dmat *OUT, int o0) {
  if (!p) return fail(SLQ_EINVAL, "plan is NULL");
  if (!p->probes_ready) return fail(SLQ_EINVAL, "no probes set, or they were consumed by a run");
  SLQ_TRY(dmat_range(OUT, o0, p->nprobes, "slq_plan_get_probes_dmat"));
  if (p->dtype != SLQ_F64 || OUT->n != p->n || OUT->ctx != p->ctx) return fail(SLQ_EINVAL, "plan and matrix do not match (fp64, same n, same context)");
  HIP_TRY(hipSetDevice(p->ctx->device));
  hipStream_t st = p->ctx->stream;
  // the probes as the estimators see them: sphere probes are stored as the normal draw g and used as
  // sqrt(n) g / ||g|| (k_fin_init), so the copy carries that scale (coefB is free until the run starts)
  k_probe_scale<<<dim3((p->bpad + 255) / 256), dim3(256), 0, st>>>(p->st, p->st.coefB);
  dim3 g((p->n + 63) / 64, (p->nprobes + 63) / 64);
  hipLaunchKernelGGL(k_panel_to_cols<double>, g, dim3(256), 0, st, p->n, (const double *)slot_ptr(p, 0), 0, p->nprobes,
                     OUT->d + (size_t)o0 * p->n, p->PW, (const double *)p->st.coefB, p->op->perm_d);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(st));
  return SLQ_OK;
}

extern "C" int slq_measure_stream(slq_context *ctx, int mode, size_t bytes_per_stream, int reps, double *gbps) {
  if (!ctx || !gbps || mode < 0 || mode > 2 || reps < 1 || bytes_per_stream < (1u << 20))
    return fail(SLQ_EINVAL, "bad arguments");
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const int64_t nvec = (int64_t)(bytes_per_stream / 16);
  double *buf = nullptr;
  HIP_TRY(hipMalloc((void **)&buf, (size_t)nvec * 32 + 64));
  double *w = buf, *q = buf + nvec * 2, *sink = q + nvec * 2;
  hipEvent_t a = nullptr, b = nullptr;
  hipError_t e = hipMemsetAsync(buf, 0, (size_t)nvec * 32 + 64, st);
  if (e == hipSuccess) e = hipEventCreate(&a);
  if (e == hipSuccess) e = hipEventCreate(&b);
  const int blocks = ctx->num_cus * 2;  // the launch shape of the streaming sweeps
  float ms = 0.f;
  if (e == hipSuccess) {
    for (int i = 0; i < 2; ++i) k_stream_probe<<<dim3(blocks), dim3(kBlock), 0, st>>>(w, q, nvec, mode, 1e-9, sink);
    e = hipEventRecord(a, st);
    for (int i = 0; i < reps; ++i) k_stream_probe<<<dim3(blocks), dim3(kBlock), 0, st>>>(w, q, nvec, mode, 1e-9, sink);
    if (e == hipSuccess) e = hipEventRecord(b, st);
    if (e == hipSuccess) e = hipEventSynchronize(b);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, a, b);
    if (e == hipSuccess) e = hipGetLastError();
  }
  if (a) hipEventDestroy(a);
  if (b) hipEventDestroy(b);
  hipFree(buf);
  if (e != hipSuccess) return fail(SLQ_EHIP, "slq_measure_stream: %s", hipGetErrorString(e));
  const double streams = mode == 1 ? 3.0 : 2.0;
  *gbps = streams * (double)nvec * 16.0 * reps / (ms * 1e-3) / 1e9;
  return SLQ_OK;
}

extern "C" int slq_plan_set_probes_device(slq_plan *p, const void *d_X, int64_t ldx) {
  if (!p || !d_X) return fail(SLQ_EINVAL, "plan/X is NULL");
  if (ldx < p->n) return fail(SLQ_EINVAL, "ldx (%lld) < n (%d)", (long long)ldx, p->n);
  HIP_TRY(hipSetDevice(p->ctx->device));
  hipStream_t st = p->ctx->stream;
  if (ldx != p->n) return fail(SLQ_EINVAL, "device probes must be contiguous columns (ldx == n)");
  if (p->nprobes < p->bpad) HIP_TRY(hipMemsetAsync(slot_ptr(p, 0), 0, (size_t)p->slot_stride * p->esz, st));
  dim3 g((p->n + 63) / 64, (p->nprobes + 63) / 64);
  PROFILED(p, SLQ_K_PROBES, {
    if (p->dtype == SLQ_F64)
      hipLaunchKernelGGL(k_cols_to_panel<double>, g, dim3(256), 0, st, p->n, (const double *)d_X, 0, p->nprobes, (double *)slot_ptr(p, 0), p->PW, p->op->perm_d);
    else
      hipLaunchKernelGGL(k_cols_to_panel<float>, g, dim3(256), 0, st, p->n, (const float *)d_X, 0, p->nprobes, (float *)slot_ptr(p, 0), p->PW, p->op->perm_d);
  });
  p->pdf_sphere = 0;
  return init_from_probes(p, 0);
}

extern "C" int slq_fttr_batch(slq_context *ctx, int nb, int n, int k, const double *theta, const double *alpha,
                              const double *beta, double *weights) {
  if (!ctx || !theta || !alpha || !beta || !weights) return fail(SLQ_EINVAL, "NULL argument");
  if (nb <= 0 || n <= 0 || k <= 0) return fail(SLQ_EINVAL, "bad sizes");
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  double *buf = nullptr;
  const size_t nth = (size_t)nb * k, nab = (size_t)nb * n;
  HIP_TRY(hipMalloc((void **)&buf, (2 * nth + 2 * nab) * 8));
  double *dth = buf, *dw = buf + nth, *da = dw + nth, *db = da + nab;
  hipError_t e = hipMemcpyAsync(dth, theta, nth * 8, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipMemcpyAsync(da, alpha, nab * 8, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipMemcpyAsync(db, beta, nab * 8, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) {
    k_fttr<<<dim3((unsigned)((nth + 127) / 128)), dim3(128), 0, st>>>(nb, n, k, dth, da, db, dw);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(weights, dw, nth * 8, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  hipFree(buf);
  if (e != hipSuccess) return fail(SLQ_EHIP, "slq_fttr_batch: %s", hipGetErrorString(e));
  return SLQ_OK;
}

extern "C" int slq_eigh_tridiag_batch(slq_context *ctx, int nb, int deg, const double *d, const double *e, double *w,
                                      double *Z) {
  if (!ctx || !d || !e || !w) return fail(SLQ_EINVAL, "ctx/d/e/w is NULL");
  if (nb <= 0 || deg <= 0 || deg > kMaxDeg) return fail(SLQ_EINVAL, "bad batch size or degree (deg <= %d)", kMaxDeg);
  // eigenvectors on chip when they fit in LDS (deg <= 141), otherwise in a global scratch that stays in L2
  const size_t lds_onchip = ((size_t)2 * deg + (size_t)deg * (deg + 1)) * 8 + (size_t)deg * sizeof(int);
  const bool zg = lds_onchip > 160 * 1024;
  const size_t lds = zg ? (size_t)2 * deg * 8 + (size_t)deg * sizeof(int) : lds_onchip;
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const size_t in = (size_t)nb * deg, zz = Z ? in * deg : 0, zs = zg ? in * (deg + 1) : 0;
  double *buf = nullptr;
  HIP_TRY(hipMalloc((void **)&buf, (3 * in + zz + zs + 1) * 8));
  double *dd = buf, *de = dd + in, *dw = de + in, *dz = Z ? dw + in : nullptr, *dscr = zg ? dw + in + zz : nullptr;
  int *dfail = (int *)(dw + in + zz + zs);
  hipError_t err = hipMemcpyAsync(dd, d, in * 8, hipMemcpyHostToDevice, st);
  if (err == hipSuccess) err = hipMemcpyAsync(de, e, in * 8, hipMemcpyHostToDevice, st);
  if (err == hipSuccess) err = hipMemsetAsync(dfail, 0, sizeof(int), st);
  if (err == hipSuccess && lds > 48 * 1024)
    err = hipFuncSetAttribute((const void *)k_eigh_tridiag<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (err == hipSuccess) {
    if (zg)
      k_eigh_tridiag<true><<<dim3(nb), dim3(64), lds, st>>>(deg, dd, de, dw, dz, dscr, dfail);
    else
      k_eigh_tridiag<false><<<dim3(nb), dim3(64), lds, st>>>(deg, dd, de, dw, dz, nullptr, dfail);
    err = hipGetLastError();
  }
  int bad = 0;
  if (err == hipSuccess) err = hipMemcpyAsync(&bad, dfail, sizeof(int), hipMemcpyDeviceToHost, st);
  if (err == hipSuccess) err = hipMemcpyAsync(w, dw, in * 8, hipMemcpyDeviceToHost, st);
  if (err == hipSuccess && Z) err = hipMemcpyAsync(Z, dz, zz * 8, hipMemcpyDeviceToHost, st);
  if (err == hipSuccess) err = hipStreamSynchronize(st);
  hipFree(buf);
  if (err != hipSuccess) return fail(SLQ_EHIP, "slq_eigh_tridiag_batch: %s", hipGetErrorString(err));
  if (bad) return fail(SLQ_ENOTCONV, "tridiagonal QL did not converge for at least one matrix");
  return SLQ_OK;
}

extern "C" int slq_quadrature_batch(slq_context *ctx, int nb, int deg, const double *d, const double *e,
                                    int fun_id, const double *fun_params, double *quad, double *nodes,
                                    double *weights) {
  if (!ctx || !d || !e) return fail(SLQ_EINVAL, "ctx/d/e is NULL");
  if (nb <= 0 || deg <= 0 || deg > kMaxDeg) return fail(SLQ_EINVAL, "bad batch size or degree");
  if (fun_id < SLQ_FUN_NONE || fun_id > SLQ_FUN_SOFTSIGN) return fail(SLQ_EINVAL, "Unknown function id %d.", fun_id);
  const int lanes = quadrature_lanes(deg);
  const size_t lds = (size_t)3 * deg * lanes * 8;
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const int bp = (nb + 63) / 64 * 64;
  StepState s;
  memset(&s, 0, sizeof(s));
  s.bpad = bp;
  s.nprobes = nb;
  s.deg = deg;
  double *buf = nullptr;
  const size_t in = (size_t)nb * deg;
  const size_t total = (size_t)(2 * (deg + 1) + 1) * bp + 2 * in + bp + 2 * (size_t)bp * deg + 8;
  HIP_TRY(hipMalloc((void **)&buf, total * 8));
  double *q = buf;
  s.alpha = q; q += (size_t)(deg + 1) * bp;
  s.nu = q; q += (size_t)(deg + 1) * bp;
  s.vnorm2 = q; q += bp;
  double *dd = q; q += in;
  double *de = q; q += in;
  double *dq = q; q += bp;
  double *dn = q; q += (size_t)bp * deg;
  double *dw = q; q += (size_t)bp * deg;
  int *dfail = (int *)q;
  int rc = SLQ_OK;
  hipError_t err = hipMemcpyAsync(dd, d, in * 8, hipMemcpyHostToDevice, st);
  if (err == hipSuccess) err = hipMemcpyAsync(de, e, in * 8, hipMemcpyHostToDevice, st);
  if (err == hipSuccess) err = hipMemsetAsync(dfail, 0, sizeof(int), st);
  if (err == hipSuccess) {
    k_load_tridiag<<<dim3((bp + 255) / 256), dim3(256), 0, st>>>(s, dd, de, nb);
    if (lds > 48 * 1024)
      err = hipFuncSetAttribute((const void *)k_quadrature, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  if (err == hipSuccess) {
    const double p0 = fun_params ? fun_params[0] : 0.0, p1 = fun_params ? fun_params[1] : 0.0;
    k_quadrature<<<dim3((nb + lanes - 1) / lanes), dim3(64), lds, st>>>(s, lanes, fun_id, p0, p1, dq, dn, dw, dfail);
    err = hipGetLastError();
  }
  int bad = 0;
  if (err == hipSuccess) err = hipMemcpyAsync(&bad, dfail, sizeof(int), hipMemcpyDeviceToHost, st);
  if (err == hipSuccess && quad) err = hipMemcpyAsync(quad, dq, (size_t)nb * 8, hipMemcpyDeviceToHost, st);
  if (err == hipSuccess && nodes) err = hipMemcpyAsync(nodes, dn, in * 8, hipMemcpyDeviceToHost, st);
  if (err == hipSuccess && weights) err = hipMemcpyAsync(weights, dw, in * 8, hipMemcpyDeviceToHost, st);
  if (err == hipSuccess) err = hipStreamSynchronize(st);
  hipFree(buf);
  if (err != hipSuccess) rc = fail(SLQ_EHIP, "slq_quadrature_batch: %s", hipGetErrorString(err));
  else if (bad) rc = fail(SLQ_ENOTCONV, "tridiagonal QL did not converge for at least one rule");
  return rc;
}

// ---------------------------------------------------------------------------------------------------
// stand-alone operator product
// ---------------------------------------------------------------------------------------------------
extern "C" int slq_operator_matmat(slq_operator *op, const void *X, int64_t ldx, void *Y, int64_t ldy, int b) {
  if (!op || !X || !Y || b <= 0) return fail(SLQ_EINVAL, "bad arguments");
  if (ldx < op->n || ldy < op->n) return fail(SLQ_EINVAL, "leading dimension < n");
  slq_plan *p = nullptr;
  // a keep_basis plan with deg = 1 gives two slots: 0 = X, 1 = Y
  SLQ_TRY(slq_plan_create(op->ctx, op, b, 1, 0, 1, &p));
  hipStream_t st = op->ctx->stream;
  int rc = slq_plan_set_probes(p, X, ldx);
  if (rc == SLQ_OK) {
    if (op->kind == OP_CSR) {
      dim3 g(p->nblkS, p->NP);
      DISPATCH(p->dtype, p->LPR,
               (k_spmm_plain<F, L><<<g, dim3(kBlock), 0, st>>>(p->n, op->rowptr, op->colind,
                                   (const F *)op->vals, (const F *)slot_ptr(p, 0), (F *)slot_ptr(p, 1), p->n)));
    } else {
      rc = apply_operator_unfused(p, 0);
      if (rc == SLQ_OK) {
        hipError_t e = hipMemcpyAsync(slot_ptr(p, 1), p->T, (size_t)p->slot_stride * p->esz, hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) rc = fail(SLQ_EHIP, "copy: %s", hipGetErrorString(e));
      }
    }
  }
  if (rc == SLQ_OK) rc = panel_to_host(p, 1, 0, b, Y, ldy, nullptr);
  slq_plan_destroy(p);
  return rc;
}

// ---------------------------------------------------------------------------------------------------
// one-shot entries
// ---------------------------------------------------------------------------------------------------
extern "C" int slq_quad_batch(slq_context *ctx, slq_operator *op, const void *X, int64_t ldx, int pdf,
                              uint64_t seed, uint64_t probe_offset, int nprobes, int deg, double rtol,
                              int orth, int fun_id, const double *fun_params, double *quad_out,
                              double *nodes_out, double *weights_out) {
  if (!ctx || !op) return fail(SLQ_EINVAL, "ctx/op is NULL");
  if (nprobes <= 0) return fail(SLQ_EINVAL, "nprobes must be positive");
  int d = deg, o = orth;
  SLQ_TRY(normalise_params(op->n, &d, &o));
  // size the probe chunk to the free device memory (all probes at once when they fit)
  size_t free_b = 0, total_b = 0;
  SLQ_TRY(slq_context_meminfo(ctx, &free_b, &total_b));
  int chunk = nprobes;
  for (;;) {
    size_t need = 0;
    SLQ_TRY(plan_bytes_on(op, chunk, d, o, 0, &need));
    if (need + ((size_t)1 << 30) <= free_b || chunk <= 8) break;
    chunk = (chunk + 1) / 2;
  }
  int rc = SLQ_OK;
  for (int c0 = 0; c0 < nprobes && rc == SLQ_OK; c0 += chunk) {
    const int nc = std::min(chunk, nprobes - c0);
    slq_plan *p = nullptr;
    rc = slq_plan_create(ctx, op, nc, d, o, 0, &p);
    if (rc != SLQ_OK) break;
    if (X)
      rc = slq_plan_set_probes(p, (const char *)X + (size_t)c0 * (size_t)ldx * esize(op->dtype), ldx);
    else
      rc = slq_plan_generate_probes(p, pdf, seed, probe_offset + (uint64_t)c0);
    if (rc == SLQ_OK) rc = slq_plan_run(p, rtol);
    if (rc == SLQ_OK)
      rc = slq_plan_quadrature(p, fun_id, fun_params, quad_out ? quad_out + c0 : nullptr,
                               nodes_out ? nodes_out + (size_t)c0 * d : nullptr,
                               weights_out ? weights_out + (size_t)c0 * d : nullptr);
    slq_plan_destroy(p);
  }
  return rc;
}

extern "C" int slq_fAv_batch(slq_context *ctx, slq_operator *op, const void *X, int64_t ldx, int nvec, int deg,
                             double rtol, int orth, int fun_id, const double *fun_params, void *Y, int64_t ldy) {
  if (!ctx || !op || !X || !Y) return fail(SLQ_EINVAL, "ctx/op/X/Y is NULL");
  if (nvec <= 0) return fail(SLQ_EINVAL, "nvec must be positive");
  if (ldx < op->n || ldy < op->n) return fail(SLQ_EINVAL, "ldx/ldy < n");
  int d = deg, o = orth;
  SLQ_TRY(normalise_params(op->n, &d, &o));
  // the whole basis of every column is kept (deg + 1 panels): chunk the columns to the free device memory
  size_t free_b = 0, total_b = 0;
  SLQ_TRY(slq_context_meminfo(ctx, &free_b, &total_b));
  int chunk = nvec;
  for (;;) {
    size_t need = 0;
    SLQ_TRY(plan_bytes_on(op, chunk, d, o, 1, &need));
    if (need + ((size_t)1 << 30) <= free_b || chunk <= 8) break;
    chunk = (chunk + 1) / 2;
  }
  const size_t es = esize(op->dtype);
  int rc = SLQ_OK;
  slq_plan *p = nullptr;
  int plan_cols = 0;
  for (int c0 = 0; c0 < nvec && rc == SLQ_OK; c0 += chunk) {
    const int nc = std::min(chunk, nvec - c0);
    if (nc != plan_cols) {
      if (p) slq_plan_destroy(p);
      p = nullptr;
      rc = slq_plan_create(ctx, op, nc, d, o, 1, &p);
      plan_cols = nc;
      if (rc != SLQ_OK) break;
    }
    rc = slq_plan_set_probes(p, (const char *)X + (size_t)c0 * (size_t)ldx * es, ldx);
    if (rc == SLQ_OK) rc = slq_plan_run(p, rtol);
    if (rc == SLQ_OK) rc = slq_plan_fun_action(p, fun_id, fun_params, (char *)Y + (size_t)c0 * (size_t)ldy * es, ldy);
  }
  if (p) slq_plan_destroy(p);
  return rc;
}

template <typename F>
static int lanczos_single(slq_context *ctx, slq_operator *op, F *v, int deg, F rtol, int orth, F *alpha,
                          F *beta, F *Q, size_t ncv) {
  if (!ctx || !op || !v || !alpha || !beta || !Q) return fail(SLQ_EINVAL, "NULL argument");
  if (op->dtype != (sizeof(F) == 8 ? SLQ_F64 : SLQ_F32)) return fail(SLQ_EINVAL, "operator dtype does not match the entry point");
  if (deg < 1) return fail(SLQ_EINVAL, "Number of steps must be positive!");
  if (deg > op->n) return fail(SLQ_EINVAL, "deg exceeds the operator dimension");
  // precondition of the reference kernel (lanczos.h:91): orth < ncv <= deg is NOT enforced there;
  // what it needs to be well defined is 2 <= ncv and orth <= ncv
  if (ncv < 2 || (size_t)std::max(orth, 0) > ncv) return fail(SLQ_EINVAL, "need ncv >= 2 and orth <= ncv (ncv=%zu, orth=%d)", ncv, orth);
  if (orth < 0) return fail(SLQ_EINVAL, "orth must be non-negative at the native boundary");
  const int n = (int)op->n;
  const bool keep = true;
  slq_plan *p = nullptr;
  SLQ_TRY(slq_plan_create(ctx, op, 1, deg, orth, keep, &p));
  int rc = slq_plan_set_probes(p, v, n);
  // Stale ring columns. The reference clears only column ncv-1 and writes column 0 on entry
  // (lanczos.h:118-120); during the first orth-1 steps its MGS sweep also walks columns ncv-1,
  // ncv-2, ... ncv-orth+1 with whatever the caller left there (lanczos.h:58 with reverse indices),
  // e.g. the previous probe's Lanczos vectors when MatrixFunction.quad reuses its Q
  // (operators.py:138-148). Reproduce that: those columns become vectors t = -1, -2, ... of the
  // ring (t = -1 is the cleared column), with nu_t = ||column||.
  const int nst = (rc == SLQ_OK && p->orth >= 2) ? p->orth - 1 : 0;
  if (nst > 0) {
    hipStream_t st = ctx->stream;
    std::vector<double> norms((size_t)nst + 1, 0.0);
    bool any = false;
    for (int k = 2; k <= nst && rc == SLQ_OK; ++k) {  // t = -k  <->  caller column ncv - k
      const F *col = Q + (size_t)(ncv - k) * n;
      double s2 = 0.0;
      for (int i = 0; i < n; ++i) s2 += (double)col[i] * (double)col[i];
      norms[(size_t)k] = std::sqrt(s2);
      if (s2 > 0.0) any = true;
    }
    if (any) {
      rc = ensure_stage(p, 1);
      for (int k = 1; k <= nst && rc == SLQ_OK; ++k) {
        const int slot = p->S - k;
        hipError_t e = hipMemsetAsync(slot_ptr(p, slot), 0, (size_t)p->slot_stride * p->esz, st);
        if (e == hipSuccess && k >= 2 && norms[(size_t)k] > 0.0) {
          e = hipMemcpyAsync(p->stage, Q + (size_t)(ncv - k) * n, (size_t)n * sizeof(F), hipMemcpyHostToDevice, st);
          if (e == hipSuccess) {
            dim3 g((n + 63) / 64, 1);
            k_cols_to_panel<F><<<g, dim3(256), 0, st>>>(n, (const F *)p->stage, 0, 1, (F *)slot_ptr(p, slot), p->PW, op->perm_d);
            e = hipStreamSynchronize(st);
          }
        }
        if (e == hipSuccess) e = hipMemcpyAsync(p->st.nu - (size_t)k * p->bpad, &norms[(size_t)k], sizeof(double), hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) rc = fail(SLQ_EHIP, "stale ring upload: %s", hipGetErrorString(e));
      }
      if (rc == SLQ_OK) p->nstale = nst;
    }
  }
  if (rc == SLQ_OK) rc = slq_plan_run(p, (double)rtol);
  std::vector<F> a(deg + 1), b(deg + 1);
  int32_t steps = 0;
  if (rc == SLQ_OK) rc = slq_plan_get_tridiag(p, a.data(), b.data(), &steps);
  if (rc == SLQ_OK) {
    // the reference writes alpha[0..steps) and beta[0..steps]; later entries keep the caller's values
    for (int t = 0; t < steps; ++t) alpha[t] = a[t];
    for (int t = 0; t <= steps; ++t) beta[t] = b[t];
    // ring columns: Lanczos vector t sits in column t % ncv; the last write wins (lanczos.h:143-147).
    // Vectors 0..steps-1 are written (vector `steps` is never normalised into the ring, :140-142).
    std::vector<F> Qfull((size_t)n * deg);
    rc = slq_plan_get_basis(p, 0, Qfull.data(), n);
    if (rc == SLQ_OK) {
      // column ncv-1 is zeroed on entry (lanczos.h:119) unless a later vector lands there
      memset(Q + (size_t)(ncv - 1) * n, 0, (size_t)n * sizeof(F));
      for (int t = 0; t < steps; ++t) memcpy(Q + (size_t)(t % ncv) * n, Qfull.data() + (size_t)t * n, (size_t)n * sizeof(F));
    }
    // v is scratch in the reference and ends as the last unnormalised residual; give it back
    if (rc == SLQ_OK) {
      const int slot = steps % p->S;
      rc = panel_to_host(p, slot, 0, 1, v, n, nullptr);
    }
  }
  slq_plan_destroy(p);
  return rc == SLQ_OK ? steps : rc;
}

extern "C" int slq_lanczos_f64(slq_context *ctx, slq_operator *op, double *v, int deg, double rtol, int orth,
                               double *alpha, double *beta, double *Q, size_t ncv) {
  return lanczos_single<double>(ctx, op, v, deg, rtol, orth, alpha, beta, Q, ncv);
}
extern "C" int slq_lanczos_f32(slq_context *ctx, slq_operator *op, float *v, int deg, float rtol, int orth,
                               float *alpha, float *beta, float *Q, size_t ncv) {
  return lanczos_single<float>(ctx, op, v, deg, rtol, orth, alpha, beta, Q, ncv);
}

#ifdef SLQ_DEBUG_TIMES
// Diagnostic build only (-DSLQ_DEBUG_TIMES, scripts/wave_drift.py): a device buffer of per-wave progress stamps.
static size_t g_dbg_bytes = 0;
extern "C" int slq_debug_times_begin(size_t bytes) {
  if (g_dbg_host_handle) hipFree(g_dbg_host_handle);
  HIP_TRY(hipMalloc((void **)&g_dbg_host_handle, bytes));
  HIP_TRY(hipMemset(g_dbg_host_handle, 0, bytes));
  g_dbg_bytes = bytes;
  HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(slq::g_dbg_times), &g_dbg_host_handle, sizeof(void *)));
  return SLQ_OK;
}
extern "C" int slq_debug_times_read(void *host, size_t bytes) {
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(host, g_dbg_host_handle, std::min(bytes, g_dbg_bytes), hipMemcpyDeviceToHost));
  return SLQ_OK;
}
#endif

// lrr_icp.hip — hand-written HIP kernels (gfx950 / CDNA4) + the C-ABI of include/lrr_icp.h.
//
// The per-iteration hot path of LRR's robust ICP:
//   K1  map moving features + exact nearest fixed feature      (k_match_ball[2] on the later iterations
//                                                               of a level, k_match_tile on the first,
//                                                               k_match_direct on coarse levels; then
//                                                               k_match_hard[2] (wave per query, with the
//                                                               full scan for what no ring settles);
//                                                               k_match_brute: literal all-vs-all)
//   K2  projected residuals, robust scale, Beaton-Tukey weights (k_f_residual, k_f_weights, k_selw_*;
//                                                               forward classes: k_fw_residual, k_fw_weights)
//   K5  normalisation statistics                                (k_to_sum, k_f_radii, k_fw_stats)
//   K3  weighted normal equations J^T P J, J^T P b              (k_f_accumulate, k_fw_accumulate)
//   K4  error metrics                                           (k_errors)
// plus the step-by-step twins behind the granular C-ABI calls (k_residual, k_strength, ...).
// Everything is fp64: the reference is double throughout (cdcl/cdcl_feature.h:21) and "match
// sets exactly" is a parity requirement.  This is distance search + small reductions: no MFMA.
// The file is compiled with -ffp-contract=off so that per-pair quantities are bit-identical to
// the CPU restatement used as checker; the only fused operations are the explicit fma() calls
// of dist2().  Reductions are two-stage with a fixed block->partial->final order, so results
// are run-to-run deterministic (no floating-point atomics anywhere).
//
// Device-internal order.  A cloud is stored in CANONICAL order: features sorted by 8x8x8-cell
// tile of the search grid, then by cell.  Per-match arrays of a direction are indexed by the
// canonical position s of the match's from-feature (so every pass streams them coalesced and
// gathers target records of neighbouring s from neighbouring memory); the target of a match is
// kept as a canonical target index.  The reference's match-list order exists only as slot[s]
// (list position of source s, -1 = not in the active subset) and is applied when matches are
// handed back to the host.  Sums are order-independent up to rounding, so this changes nothing
// observable beyond the last bits.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "lrr_icp.h"

// ---------------------------------------------------------------------------------------------
// host-side objects
// ---------------------------------------------------------------------------------------------
namespace {

constexpr int kBlock = 256;          // 4 wavefronts of 64
constexpr int kMaxBlocks = 2048;     // reduction grids: 8 blocks / CU on 256 CUs (one dependent gather chain per thread)
constexpr int kWide = 1024;          // block size of the fused estimate's streaming passes: at most kWideBlocks of them, so a
constexpr int kWideBlocks = 256;     // consumer's prologue sums <= 256 rows of producer partials (one block per CU, 16 waves each)
constexpr int kStreamBatch = 2;      // sources a thread of those passes has in flight at once
constexpr int kMaxRed = 96;          // max doubles reduced per pass (78 + 12 for dof 12)
constexpr int kRecStride = 10;       // feature record: x y z pxx pxy pxz pyy pyz pzz strength
constexpr int kTile = 8;             // canonical order: tiles of kTile^3 grid cells
constexpr int kScanUnroll = 8;       // candidate loads in flight per lane in the global-memory grid search
constexpr int kRingMax = 8;          // rings walked by the wave-per-query phase before the brute-force fallback
constexpr int kTileBudget = 48;      // candidates a lane may scan from LDS before handing its query over
constexpr int kTileCap = 2048;       // target points a block may stage in LDS (64 KB) ...
constexpr int kTileRows = 256;       // ... spread over at most this many (z,y) rows of the target grid
constexpr int kTileNx = 20;          // ... of at most this many cells each
constexpr int kSelThreads = 1024;
constexpr int kMaxSelJobs = 32;
constexpr int kProfEvents = 4096;
constexpr int kCountSlots = 64;      // K1 list counters: cleared once per kCountSlots submissions
constexpr int kSelCap = 8192;        // keys of one window bin the final block can select from in LDS
constexpr int kSelStates = 16;      // remembered results (window centres)
constexpr int kSelHistBlocks = 128;  // blocks per job of the chip-wide select kernels (<= kSelPre keys per thread up to 262k keys)
// fused scratch: P1 (2*kMaxBlocks x 5), P2 (kMaxBlocks x 4), P3 (kMaxBlocks x 2), P4 (2*kMaxBlocks x 1),
// P6 (2*256 x 96), output block
constexpr int kAccBlocks = 128;      // accumulate grid per direction, generic form (90 fp64 accumulators per thread: 1 wave/SIMD)
constexpr int kKronBlocks = 256;     // ... of the 12-dof Kronecker form (72 accumulators: two blocks per CU, 2 waves/SIMD)
constexpr int kKron = 72;            // its sums: 10 (unique products of (l,1)(l,1)^T) x 6 (unique projector entries) + 12 (rhs)
constexpr int kFusedP1 = 0, kFusedP2 = kFusedP1 + 2 * kMaxBlocks * 5, kFusedP3 = kFusedP2 + kMaxBlocks * 4,
              kFusedP4 = kFusedP3 + kMaxBlocks * 2, kFusedP6 = kFusedP4 + 2 * kMaxBlocks,
              kFusedOut = kFusedP6 + 2 * kKronBlocks * 96, kFusedP5 = kFusedOut + 256,  // P5: per-block counts of keys below the two selection windows
              kFusedDoubles = kFusedP5 + 2 * kMaxBlocks * 2;

// one candidate of the search grid: 32 B = two 16-B loads
struct GridPoint { double x, y, z; int canon, orig; };
// one source feature in canonical order
struct SrcPoint { double x, y, z; long long orig; };
// shape_ and the directions the angle diagnostics compare: sheets directions_[0] (the normal), tubes
// directions_[0] x directions_[1] (the tangent)
struct ShapeRec { double sd[3], td[3]; int shape, ndirs; };
struct Chunk { int start, count; };  // <= kBlock consecutive canonical features of ONE tile
struct SelState { int bin, count; long long k, k_in_bin, m; };  // window bin holding rank k of one job

struct SelJob {
  const double *k0; int n0;
  const double *k1; int n1;
  long long k;     // 0-based rank; < 0: MAD rank (m-1)/2+1 of the m counted keys (cnt rows below)
  double *out;     // out[0] = k-th smallest, out[1] = sum of keys < it, out[2] = their count, out[3] = m
  const double *cnt;               // per-block partial counts (1 double per row) or nullptr
  int cnt_a0, cnt_a1, cnt_b0, cnt_b1;  // two row ranges whose sum is m
  long long m_fixed;               // m when cnt == nullptr
  int state;                       // slot of the previous result that centres the window (< 0: none)
  // folded selection (k_f_weights<true> + k_selw_fold_final): the rows of P5 that belong to this job's list(s) and
  // its column (0: strength keys, 1: spatial keys)
  int row_a0, row_a1, row_b0, row_b1, col;
};
constexpr int kFoldStage = 192;     // in-window keys a block of the weight pass can stage per selection
constexpr int kFoldAllMax = 4096;   // matches per list up to which the folded selection gathers every key (lrr_icp_estimate_symmetric)
constexpr int kSelPack = 16;     // (the 30 order statistics of a pair of MUSE lists: two packs)
struct SelPack { SelJob j[kSelPack]; };  // the jobs of one windowed selection, passed as a kernel argument


}  // namespace

struct K1Lane {
  hipStream_t stream = nullptr;
  double *d_bd = nullptr; int *d_bi = nullptr; size_t brute_cap = 0;   // brute-force partial argmin
  int *d_hard = nullptr, *d_active = nullptr; size_t list_cap = 0;
  int *d_counts = nullptr;       // kCountSlots x {hard, overflow, active, -}: one zeroed slot per K1 submission
  int count_slot = 0;
  int *d_order = nullptr; size_t order_cap = 0;
};

struct lrr_icp_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  double *d_partials = nullptr;  // kMaxBlocks * kMaxRed
  double *d_final = nullptr;     // kMaxRed + select outputs
  double *h_pinned = nullptr;    // results staging (mapped: kernels write it through h_pinned_dev)
  double *h_pinned_dev = nullptr;
  SelJob *d_jobs = nullptr;
  SelJob *h_jobs = nullptr;
  SelState *d_selstate = nullptr;
  int *d_selhist = nullptr;      // kMaxSelJobs x 2048, zero between selections
  double *d_selprev = nullptr;   // kSelStates previous results
  double *d_selwin = nullptr;    // kSelStates relative half-widths of the folded selection's windows (self-tuning)
  // folded selection (host side): the last two results of the 4 selections, and what they belonged to
  double sel_last[4] = {0, 0, 0, 0}, sel_prev[4] = {0, 0, 0, 0};
  const void *sel_dirs[2] = {nullptr, nullptr};
  int64_t sel_n[2] = {-1, -1};
  int sel_runs = 0;
  unsigned long long *d_selcoll = nullptr;  // kMaxSelJobs x kSelCap gathered keys
  int *d_selcursor = nullptr;    // kMaxSelJobs, zero between selections
  double *d_fused = nullptr;     // P1..P6 partial tables + the fused output block
  double *h_fused = nullptr, *h_fused_dev = nullptr;
  std::vector<double> fused_vals;  // the values of the last tagged result set (wait_chunks)
  double *d_muse = nullptr;        // gathered keys of a coarse list (lrr_icp_muse_scale)
  size_t muse_cap = 0;
  // completion word of the reduce kernels (mapped host memory) and their last-block counter
  unsigned long long *h_seq = nullptr, *h_seq_dev = nullptr, seq = 0;
  // block rows of k_errors, written by the blocks themselves: kWideBlocks rows of 8 doubles {s0, tag, s1, tag, s2, tag, -, -}
  double *h_rows = nullptr, *h_rows_dev = nullptr;
  struct lrr_icp_dir *err_pending = nullptr;  // the direction whose k_errors rows are on their way (lrr_icp_errors_launch)
  double err_tag = 0.0;
  int err_nb = 0;
  int *d_done = nullptr;
  unsigned long long *d_cert_stats = nullptr;  // {queries, searched} of the certificate kernel
  int *d_fault = nullptr;  // set by the residual passes when a match index is outside the to-cloud (a K1 phase left a source
                           // unserved): the estimate then fails with an error instead of gathering from a wild address
  // K1 scratch: one set per submission lane.  Lane 0 runs on `stream`; lane 1 has its own stream so
  // that the two directions of a symmetric estimator search concurrently (lrr_icp_match_pair).
  K1Lane lane[2];
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  // match-kernel profile (event pairs resolved lazily)
  bool prof = false;
  unsigned prof_every = 1, prof_tick = 0;
  int64_t cert_host_seen = 0, cert_host_listed = 0;
  int64_t k1_paths[LRR_ICP_K1_PATHS] = {0};  // searches (directions) served by each K1 state (lrr_icp_k1_path_counts)
  std::vector<hipEvent_t> pe;
  int pe_used = 0;
  double match_ms = 0.0;
  int64_t match_launches = 0;
  std::vector<float> match_samples;  // elapsed ms of every timed K1 submission since the last reset (lrr_icp_match_profile_samples)
};

struct lrr_icp_cloud {
  lrr_icp_ctx *ctx = nullptr;
  int64_t n = 0;
  int float_strength = 0;
  void *slab = nullptr;         // one allocation holding every array below
  double *rec = nullptr;        // n x 10, canonical order
  SrcPoint *src = nullptr;      // n, canonical order (K1 source reads)
  int *orig_of = nullptr;       // canonical -> original index
  int *canon_of = nullptr;      // original -> canonical index
  GridPoint *gp = nullptr;      // n, sorted by grid cell (raster z,y,x): the search structure
  ShapeRec *shp = nullptr;      // n, canonical order: shapes and directions (post-convergence diagnostics); space reserved in the slab,
                                // filled by lrr_icp_cloud_set_shapes
  bool has_shapes = false;
  std::vector<int> h_orig_of;   // host copy of orig_of for later per-feature uploads
  double *str = nullptr;        // n, canonical order: strength_ alone (the passes that need nothing else of the from-record)
  float4 *gpf = nullptr;        // n, raster order like gp: (x, y, z) in cell units relative to the grid origin as floats + canonical index bits:
                                // the single-precision pre-filter of the ball search
  GridPoint *cpt = nullptr;     // n, canonical order: location + indices of target `canon` in one 32-B read (previous-match bound)
  double *cov = nullptr;        // n x 9, canonical order: covariance_ (own allocation; lrr_icp_cloud_set_covariances, the CDC-EM E-step)
  double *loc_orig = nullptr;   // n x 3, ORIGINAL order: the box test of the device-side ROI selection streams it (FeaturesInROIs walks the features in this order)
  int *cell_start = nullptr;    // ncell + 1
  int *fine_start = nullptr;    // ncell * fine + 1: every cell cut into `fine` slices along x (ball search)
  int fine = 1;
  Chunk *chunks = nullptr; int nchunks = 0;
  double lo[3] = {0, 0, 0}, h = 1.0, inv_h = 1.0, tol = 0.0;
  int dim[3] = {1, 1, 1};
  // Voronoi-lookup semantics of the production rrl matcher (lrr_icp_cloud_build_voronoi): the occupied voxels of
  // the reference's 1 mm index volume as a cloud of their own, and the feature each stands for
  lrr_icp_cloud *vor = nullptr;
  int *vor_site_to_feature = nullptr;  // site canonical index -> canonical index of the feature in THIS cloud
  std::vector<int> h_vor_site_to_orig;  // ... -> its original index (stand-alone queries)
  double vor_origin[3] = {0, 0, 0};
  int vor_size[3] = {1, 1, 1};
};

struct lrr_icp_dir {
  lrr_icp_ctx *ctx = nullptr;
  lrr_icp_cloud *from = nullptr, *to = nullptr;
  int64_t cap = 0;   // allocated length of the per-source arrays (>= from->n)
  int64_t n = 0;     // active matches
  int64_t prev_n = -1;  // subset size of the last grid match on this source cloud (-1: none): m_to is a valid bound for those
  int64_t searches = 0; // grid searches on this source cloud so far (m_to holds the last match of every source ever active)
  int *slot = nullptr;   // list position of source s, -1 when s is not in the active subset
  int *m_to = nullptr;   // canonical target index
  int *m_site = nullptr; // Voronoi-lookup semantics: canonical index of the matched SITE in to->vor (-1: never searched); m_to then holds the feature behind it
  bool last_voronoi = false;  // the last search ran with the lookup semantics (its bounds refer to the site cloud)
  double *w = nullptr, *wc = nullptr, *r2 = nullptr, *key_g = nullptr, *key_s = nullptr;
  double *q_loc = nullptr;     // 3 per source: location of the matched target, cached by the fused residual pass
  bool qloc_in_step = false;   // ... and in step with m_to for EVERY source: written by a fused estimate over all sources, no search
                               // since other than the inline certificate kernel (which marks the entries it changes): that kernel may stream it
  bool qps_in_step = false;    // the same for the rest of the matched target's record (projector, strength: 7 doubles per source behind the
                               // locations, at q_loc + 3 n): the symmetric estimate's residual and normal-equation passes stream it then
  bool last_search_cert = false;  // the last search went through a certificate kernel: the run is near its fixed point, the next
                               // estimate caches the whole record
  double *from_loc = nullptr;  // 3 per source: pre-mapped copies
  int from_is_copy = 0;
  unsigned *flag = nullptr; int64_t flag_cap = 0;  // per target: stamp of the last call that matched it
  unsigned flag_gen = 0;
  // K1 certificate state (k_match_cert): per-source lower bound of the second-nearest distance under the
  // transform of the last search, valid when have_bounds
  float *d2lb = nullptr;
  bool have_bounds = false;
  lrr_icp_xform prev_xf;
  // list positions not materialised by the last search (finest level): slot[s] = perm^-1(orig) under these keys
  bool slot_lazy = false;
  unsigned slot_keys[4] = {0, 0, 0, 0};
  int slot_half = 1;
  // the counter of the sources the last certificate kernel listed (device), and its value as the estimate that
  // followed read it back (-1: unknown): a short list goes straight to the wave-per-query kernel
  const int *cert_list_count = nullptr;
  int last_list_count = -1;
  bool cert_inline = false;  // the last search was k_match_cert_inline2: its statistics are added on the host
  double err_stash[3] = {0.0, 0.0, 0.0};  // error sums collected on this direction's behalf (lrr_icp_errors_collect)
  bool err_stashed = false;
};

namespace {

inline float __int_as_float_host(int v) { float f; std::memcpy(&f, &v, sizeof f); return f; }

int fail(lrr_icp_ctx *ctx, int code, const std::string &msg) {
  if (ctx) ctx->err = msg;
  return code;
}

#define HIPCHK(ctx, expr)                                                                    \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess)                                                                    \
      return fail((ctx), LRR_ICP_ERR_HIP,                                                    \
                  std::string(#expr) + ": " + hipGetErrorString(e_));                        \
  } while (0)

// ---------------------------------------------------------------------------------------------
// device helpers — canonical operation order shared with the CPU checker
// ---------------------------------------------------------------------------------------------
struct XfD { double A[9], t[3], c[3]; };
struct NormD { double cm[3], rm, cf[3], rf; };
struct Keys { unsigned k[4]; };
// per-location parameters of a batch of query locations, refreshed by the host before every phase
struct BLoc {
  XfD xf[2];        // [0] forward transform (moving -> fixed), [1] backward
  Keys keys[2];     // subset permutations of the two lists
  int n_points[2];  // active subset sizes
  int active;       // 0: the location takes no part in this submission
  int mode[2];      // LRR_ICP_SCALE_RMS / LRR_ICP_SCALE_INJECTED per role (F-list, B-list)
  int pad_;
  double inj[2];
  double jac[3];    // center_moving_ of the forward transform as the accumulation sees it
};

struct GridView {
  const GridPoint *gp;   // points sorted by grid cell
  const int *cell_start;
  const int *fine_start;  // cells cut into `fine` x-slices; points of a cell are sorted by slice
  const float4 *gpf;      // gp as floats in cell units + canonical index (pre-filter)
  const GridPoint *cpt;   // points in canonical order (exact confirmation)
  double lox, loy, loz, h, inv_h, tol;
  int dx, dy, dz, n, fine;
  // Voronoi-lookup semantics (LRR_ICP_NN_VORONOI): the mapped query is first rounded to the centre of its voxel of
  // the 1 mm volume (origin qo*, size qs*), clamped to the volume (rrl .txx:530-543)
  int quant, qsx, qsy, qsz;
  double qox, qoy, qoz;
};
__device__ __forceinline__ void vor_quantize(const GridView &g, double &x, double &y, double &z) {
  if (!g.quant) return;
  x = g.qox + fmin(fmax(floor((x - g.qox) + 0.5), 0.0), (double)(g.qsx - 1));
  y = g.qoy + fmin(fmax(floor((y - g.qoy) + 0.5), 0.0), (double)(g.qsy - 1));
  z = g.qoz + fmin(fmax(floor((z - g.qoz) + 0.5), 0.0), (double)(g.qsz - 1));
}

__device__ __forceinline__ void map_loc(const XfD &T, double x0, double x1, double x2, double &y0,
                                        double &y1, double &y2) {
  // R (x - c) + t   (cdcl_trans_rigid3d.cxx:171-174, cdcl_trans_affine.txx:112-115)
  double d0 = x0 - T.c[0], d1 = x1 - T.c[1], d2 = x2 - T.c[2];
  y0 = ((T.A[0] * d0 + T.A[1] * d1) + T.A[2] * d2) + T.t[0];
  y1 = ((T.A[3] * d0 + T.A[4] * d1) + T.A[5] * d2) + T.t[1];
  y2 = ((T.A[6] * d0 + T.A[7] * d1) + T.A[8] * d2) + T.t[2];
}

__device__ __forceinline__ double dist2(double qx, double qy, double qz, double px, double py,
                                        double pz) {
  double dx = qx - px, dy = qy - py, dz = qz - pz;
  return fma(dz, dz, fma(dy, dy, dx * dx));
}

// e^T P e clamped at 0 (cdcl .txx:214-220): v = e^T P, then v . e; P symmetric, 6 unique
__device__ __forceinline__ double proj_residual2(double y0, double y1, double y2, double q0,
                                                 double q1, double q2, double pxx, double pxy,
                                                 double pxz, double pyy, double pyz, double pzz) {
  double e0 = y0 - q0, e1 = y1 - q1, e2 = y2 - q2;
  double v0 = (e0 * pxx + e1 * pxy) + e2 * pxz;
  double v1 = (e0 * pxy + e1 * pyy) + e2 * pyz;
  double v2 = (e0 * pxz + e1 * pyz) + e2 * pzz;
  double r2 = (v0 * e0 + v1 * e1) + v2 * e2;
  return r2 < 0.0 ? 0.0 : r2;
}

__device__ __forceinline__ unsigned fmix32(unsigned h) {
  h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
  return h;
}

// keyed permutation of [0, n): 4-round Feistel over 2^(2*half) >= n, cycle-walked
__device__ __forceinline__ int subset_perm(int i, int n, int half, const Keys &keys) {
  unsigned mask = (1u << half) - 1u;
  unsigned long long x = (unsigned)i;
  do {
    unsigned L = (unsigned)(x >> half), R = (unsigned)(x & mask);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      unsigned F = fmix32(R ^ keys.k[r]) & mask;
      unsigned nL = R;
      R = L ^ F;
      L = nL;
    }
    x = ((unsigned long long)L << half) | R;
  } while (x >= (unsigned long long)n);
  return (int)x;
}

__device__ __forceinline__ int subset_perm_inv(int p, int n, int half, const Keys &keys) {
  unsigned mask = (1u << half) - 1u;
  unsigned long long x = (unsigned)p;
  do {
    unsigned L = (unsigned)(x >> half), R = (unsigned)(x & mask);
#pragma unroll
    for (int r = 3; r >= 0; --r) {
      unsigned pR = L;
      unsigned pL = R ^ (fmix32(L ^ keys.k[r]) & mask);
      L = pL;
      R = pR;
    }
    x = ((unsigned long long)L << half) | R;
  } while (x >= (unsigned long long)n);
  return (int)x;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// Block-wide sum of K per-thread values; result for value k lands in partials[blockIdx.x*K + k].
// Fixed order.  Few values: lanes by shuffle tree, then the 4 waves in index order.  Many values
// (the normal equations: 90): a shuffle tree per value costs 12 cross-lane operations each, so the
// wave transposes 16 values at a time through LDS instead — lane (part, k) adds value k of 16
// lanes in lane order, two shuffles join the four parts — then the 4 waves in index order.
// ld > 0: column-major destination (value k of block b at partials[k * ld + b]) — what k_reduce_partials reads
// coalesced when there are hundreds of rows of many values
template <int K, int NW = kBlock / 64>
__device__ __forceinline__ void block_reduce_store(double (&v)[K], double *partials, int ld = 0) {
  __shared__ double red[NW][K];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if constexpr (K <= 8) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      double s = wave_sum(v[k]);
      if (lane == 0) red[wv][k] = s;
    }
  } else {
    constexpr int C = 16;
    __shared__ double tr[NW][C][65];  // [value][lane], rows padded against bank conflicts
    const int kk = lane & 15, part = lane >> 4;
#pragma unroll
    for (int c0 = 0; c0 < K; c0 += C) {
#pragma unroll
      for (int k = 0; k < C; ++k)
        if (c0 + k < K) tr[wv][k][lane] = v[c0 + k];
      __syncthreads();
      double s = 0.0;
#pragma unroll
      for (int j = 0; j < 16; ++j) s += tr[wv][kk][part * 16 + j];
      s += __shfl_xor(s, 16, 64);
      s += __shfl_xor(s, 32, 64);
      if (lane < C && c0 + lane < K) red[wv][c0 + lane] = s;
      __syncthreads();
    }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < K; k += NW * 64) {
    double s = red[0][k];
#pragma unroll
    for (int w2 = 1; w2 < NW; ++w2) s += red[w2][k];
    if (ld > 0) partials[(size_t)k * ld + blockIdx.x] = s;
    else partials[(size_t)blockIdx.x * K + k] = s;
  }
}

// Workgroups are dealt round robin to the 8 XCDs, each with its own L2.  xcd_block turns block lb of a
// grid of nb (a multiple of 8) into a logical block so that every XCD works on one contiguous eighth
// of a canonical (spatially sorted) range and its L2 holds one region of the gathered records.
__device__ __forceinline__ int xcd_block(int lb, int nb) { return (lb & 7) * (nb >> 3) + (lb >> 3); }

// K block-wide totals at once (per-thread partial sums in, totals out, same value in every thread
// and in every block: lanes by shuffle tree, then the 4 waves in index order).  The consumer kernels
// of the fused estimate derive their scalars from the producers' partial tables with it: one memory
// round trip and one barrier pair for all scalars of a kernel.
template <int K, int NW = kBlock / 64>
__device__ __forceinline__ void block_sum_all(double (&v)[K]) {
  __shared__ double sh[NW][K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    double s = wave_sum(v[k]);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6][k] = s;
  }
  __syncthreads();
  if constexpr (NW > 4) {
    // wide blocks: K threads add the NW wave sums (every thread doing so would hold K x NW values in registers at once)
    __shared__ double tot[K];
    if ((int)threadIdx.x < K) {
      double t = sh[0][threadIdx.x];
#pragma unroll
      for (int w2 = 1; w2 < NW; ++w2) t += sh[w2][threadIdx.x];
      tot[threadIdx.x] = t;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = tot[k];
  } else {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      double t = sh[0][k];
#pragma unroll
      for (int w2 = 1; w2 < NW; ++w2) t += sh[w2][k];
      v[k] = t;
    }
  }
  __syncthreads();
}

// ---- tagged results in mapped host memory ------------------------------------------------------------------
// A result crosses to the host as a 16-byte chunk {value, check}: check = chunk_check(tag, bits of value), tag = the
// submission's number.  The host takes a value only when the check word it reads belongs to the tag it waits for AND
// to the value bits it reads (chunk_take): the two 8-byte halves of a chunk may arrive in any order or at different
// times (no assumption that the 16-byte store lands indivisibly, no fence on the device) — a half that is still the
// previous submission's, or a slot that was never written, fails the test and the host polls on.  Naturally aligned
// 8-byte words are the unit of both the device's store and the host's load.
__host__ __device__ inline unsigned long long chunk_check(unsigned long long tag, unsigned long long vbits) {
  unsigned long long x = (vbits ^ (tag * 0x9E3779B97F4A7C15ull)) * 0xBF58476D1CE4E5B9ull;
  x ^= x >> 31;
  x = (x + tag) * 0x94D049BB133111EBull;
  x ^= x >> 29;
  return x | 1ull;  // never 0: a cleared slot holds no result
}
__device__ __forceinline__ double2 tagged_chunk(double value, double tag) {
  const unsigned long long c = chunk_check((unsigned long long)tag, (unsigned long long)__double_as_longlong(value));
  return make_double2(value, __longlong_as_double((long long)c));
}

// one wavefront per output value: lane l sums blocks l, l+64, ... in order, then the fixed
// shuffle tree — deterministic for a given (nblocks, K)
// The results also go straight into mapped host memory (host[head + k]; block 0 first forwards the
// `head` scalars earlier kernels left in `fwd`): the host reads them after the stream
// synchronisation, no copy in between.
// The block that finishes last then stores the submission's sequence number in a mapped host word:
// the host polls that word (half the latency of a stream synchronisation, measured 6.9 vs 12.6 us).
// cnt_a / cnt_b (may be null): two device counters forwarded behind the K results (how many sources the last K1
// certificate kernels listed: the host sizes the next search with them)
__global__ __launch_bounds__(64) void k_reduce_partials(const double *partials, int nblocks, int K, double *out,
                                                        const double *fwd, int head, double *host, int *done,
                                                        volatile unsigned long long *host_seq, unsigned long long seq,
                                                        const int *cnt_a = nullptr, const int *cnt_b = nullptr, int ld = 0,
                                                        double tag = 0.0, const int *fault = nullptr) {
  int k = blockIdx.x;
  double s = 0.0;
  if (ld > 0) for (int b = threadIdx.x; b < nblocks; b += 64) s += partials[(size_t)k * ld + b];  // column-major rows (block_reduce_store)
  else for (int b = threadIdx.x; b < nblocks; b += 64) s += partials[(size_t)b * K + k];
  s = wave_sum(s);
  if (tag != 0.0) {
    // tagged results: value i travels as the 16-byte chunk {value, check word} at host[2 i]; the host takes a value once
    // the check word of THIS tag and THIS value is there (wait_chunks / chunk_take) — no fence, no counter, no last block
    double2 *hc = reinterpret_cast<double2 *>(host);
    if (threadIdx.x == 0) { out[k] = s; hc[head + k] = tagged_chunk(s, tag); }
    if (k == 0) {
      for (int t = threadIdx.x; t < head; t += 64) hc[t] = tagged_chunk(fwd[t], tag);
      if (threadIdx.x == 0) hc[head + K] = tagged_chunk(cnt_a ? (double)*cnt_a : -1.0, tag);
      if (threadIdx.x == 1) hc[head + K + 1] = tagged_chunk(cnt_b ? (double)*cnt_b : -1.0, tag);
      if (threadIdx.x == 2) hc[head + K + 2] = tagged_chunk(fault ? (double)*fault : 0.0, tag);  // (checked_match)
    }
    return;
  }
  if (threadIdx.x == 0) { out[k] = s; host[head + k] = s; }
  if (k == 0) for (int t = threadIdx.x; t < head; t += 64) host[t] = fwd[t];
  if (k == 0 && threadIdx.x == 0) {
    host[head + K] = cnt_a ? (double)*cnt_a : -1.0;
    host[head + K + 1] = cnt_b ? (double)*cnt_b : -1.0;
  }
  __threadfence_system();  // this block's host stores are visible before it is counted
  if (threadIdx.x == 0 && atomicAdd(done, 1) == K - 1) {
    *done = 0;  // ready for the next submission (same stream: nothing runs in between)
    __threadfence_system();
    *host_seq = seq;
  }
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// distance from coordinate v to the slab [lo, hi] widened by tol (cells own their points up to
// rounding: tol covers the difference between floor((x-lo)/h) and the exact cell walls)
__device__ __forceinline__ double slab_gap(double v, double lo, double hi, double tol) {
  double g = v < lo ? lo - v : (v > hi ? v - hi : 0.0);
  g -= tol;
  return g > 0.0 ? g : 0.0;
}

// best candidate so far: squared distance, original index (tie-break: a linear scan in index
// order returns the lowest index among equally distant points), canonical index (the result)
struct Best { double d; int orig, canon; };
__device__ __forceinline__ void consider(Best &b, double d, int orig, int canon) {
  if (d < b.d || (d == b.d && orig < b.orig)) { b.d = d; b.orig = orig; b.canon = canon; }
}

// A match index as the passes after K1 gather with it: inside the to-cloud, or 0 with the context's fault word set (and stored as 0; the
// estimate that follows reports "match index out of range" instead of reading from a wild address).  K1 writes a valid
// index for every active source — this is the guard behind that invariant (round 2 lost a GPU box to a work-in-progress
// search that left -1 in m_to: to_rec[-1 * 10] lies below the cloud's slab).
__device__ __forceinline__ int checked_match(int mt, int nt, int *fault, const int *stored) {
  if ((unsigned)mt < (unsigned)nt) return mt;
  if (fault) atomicOr(fault, 1);
  *const_cast<int *>(stored) = 0;  // the passes behind this one gather with the stored index too
  return 0;
}

// stopping bound of the ring search: distance from the query to the nearest INTERIOR face of the
// (2r+1)^3 block around its cell (faces on the grid boundary have nothing behind them)
__device__ __forceinline__ double ring_bound(const GridView &g, int cx, int cy, int cz, int r, double yx,
                                             double yy, double yz) {
  double lb = INFINITY;
  if (cx - r > 0) lb = fmin(lb, yx - (g.lox + (cx - r) * g.h));
  if (cx + r < g.dx - 1) lb = fmin(lb, (g.lox + (cx + r + 1) * g.h) - yx);
  if (cy - r > 0) lb = fmin(lb, yy - (g.loy + (cy - r) * g.h));
  if (cy + r < g.dy - 1) lb = fmin(lb, (g.loy + (cy + r + 1) * g.h) - yy);
  if (cz - r > 0) lb = fmin(lb, yz - (g.loz + (cz - r) * g.h));
  if (cz + r < g.dz - 1) lb = fmin(lb, (g.loz + (cz + r + 1) * g.h) - yz);
  return lb;
}
// true when no unvisited point can beat (or tie with a lower index) the best so far
__device__ __forceinline__ bool ring_settled(double lb, double tol, double bd) {
  if (lb == INFINITY) return true;  // the block covers the whole grid
  lb -= tol;
  return lb > 0.0 && bd < lb * lb;
}

#include "lrr_icp_k1.inc"
#include "lrr_icp_solve.inc"
#include "lrr_icp_passes.inc"
#include "lrr_icp_fused.inc"
#include "lrr_icp_grid.inc"

// ---------------------------------------------------------------------------------------------
// host helpers
// ---------------------------------------------------------------------------------------------
inline int grid_for(int64_t n) {  // a multiple of 8 blocks: see xcd_block
  int64_t b = ((n + kBlock - 1) / kBlock + 7) / 8 * 8;
  if (b < 8) b = 8;
  return (int)std::min<int64_t>(b, kMaxBlocks);
}
// exclusive scan of n ints in place on stream st; sums: scratch of >= n / kScanItems + 1 ints
int device_scan(lrr_icp_ctx *ctx, hipStream_t st, int *data, long long n, int *sums) {
  const int nb = (int)((n + kScanItems - 1) / kScanItems);
  if (nb <= 0) return LRR_ICP_OK;
  hipLaunchKernelGGL(k_scan_blocks, dim3(nb), dim3(kScanThreads), 0, st, data, n, sums);
  HIPCHK(ctx, hipGetLastError());
  if (nb > 1) {
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(kScanThreads), 0, st, sums, nb);
    HIPCHK(ctx, hipGetLastError());
    hipLaunchKernelGGL(k_scan_add, dim3(nb), dim3(kScanThreads), 0, st, data, n, sums);
    HIPCHK(ctx, hipGetLastError());
  }
  return LRR_ICP_OK;
}
inline int grid_wide(int64_t n) {  // blocks of kWide threads, a multiple of 8, at most kWideBlocks
  int64_t b = ((n + kWide - 1) / kWide + 7) / 8 * 8;
  if (b < 8) b = 8;
  return (int)std::min<int64_t>(b, kWideBlocks);
}
// the same for a launch that serves both lists of a symmetric estimate: half of the chip's one round each, so that no
// block waits for a compute unit to come free (1024-thread blocks: one per CU)
inline int grid_wide_pair(int64_t n) { return std::min(grid_wide(n), kWideBlocks / 2); }
inline int blocks_for(int64_t n) { return (int)std::max<int64_t>(1, (n + kBlock - 1) / kBlock); }

XfD to_dev(const lrr_icp_xform *x) {
  XfD d;
  std::memcpy(d.A, x->A, sizeof d.A);
  std::memcpy(d.t, x->t, sizeof d.t);
  std::memcpy(d.c, x->c, sizeof d.c);
  return d;
}
NormD to_dev(const lrr_icp_norm *m) {
  NormD d;
  if (m) {
    std::memcpy(d.cm, m->cm, sizeof d.cm); d.rm = m->rm;
    std::memcpy(d.cf, m->cf, sizeof d.cf); d.rf = m->rf;
  } else {
    d.cm[0] = d.cm[1] = d.cm[2] = 0.0; d.rm = 1.0;
    d.cf[0] = d.cf[1] = d.cf[2] = 0.0; d.rf = 1.0;
  }
  return d;
}
XfD identity_xf() {
  XfD d;
  std::memset(&d, 0, sizeof d);
  d.A[0] = d.A[4] = d.A[8] = 1.0;
  return d;
}

GridView view_of(const lrr_icp_cloud *c) {
  GridView g;
  g.gp = c->gp; g.cell_start = c->cell_start; g.fine_start = c->fine_start; g.fine = c->fine;
  g.gpf = c->gpf; g.cpt = c->cpt;
  g.lox = c->lo[0]; g.loy = c->lo[1]; g.loz = c->lo[2];
  g.h = c->h; g.inv_h = c->inv_h; g.tol = c->tol;
  g.dx = c->dim[0]; g.dy = c->dim[1]; g.dz = c->dim[2];
  g.n = (int)c->n;
  g.quant = 0; g.qsx = g.qsy = g.qsz = 1; g.qox = g.qoy = g.qoz = 0.0;
  return g;
}
// the search structure of a cloud's Voronoi sites, with the voxel rounding of the queries switched on
GridView view_of_voronoi(const lrr_icp_cloud *c) {
  GridView g = view_of(c->vor);
  g.quant = 1;
  g.qsx = c->vor_size[0]; g.qsy = c->vor_size[1]; g.qsz = c->vor_size[2];
  g.qox = c->vor_origin[0]; g.qoy = c->vor_origin[1]; g.qoz = c->vor_origin[2];
  return g;
}

PairSrc pairs_of(const lrr_icp_dir *d) {
  PairSrc ps;
  ps.from_src = d->from->src; ps.from_str = d->from->str; ps.from_rec = d->from->rec; ps.to_rec = d->to->rec;
  ps.slot = d->slot; ps.m_to = d->m_to;
  ps.from_loc = d->from_is_copy ? d->from_loc : nullptr;
  ps.nsrc = (int)d->from->n;
  return ps;
}

// finish a block-partials reduction: final sum on device, copy K doubles to host, synchronise
// waits until the reduce kernel of submission `seq` has stored its results in mapped host memory:
// polls the sequence word it writes last; falls back to a stream synchronisation (which also reports
// a failed kernel) when the word does not arrive within 50 ms
int wait_results(lrr_icp_ctx *ctx, unsigned long long seq) {
  volatile unsigned long long *w = ctx->h_seq;
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spin = 0; *w != seq; ++spin) {
    if ((spin & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(50)) {
      HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
      if (*w != seq) return fail(ctx, LRR_ICP_ERR_HIP, "results did not arrive");
      break;
    }
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  return LRR_ICP_OK;
}

// one tagged chunk (tagged_chunk): true and *value once the check word in the chunk belongs to `tag` and to the value
// bits beside it; a torn or stale chunk (either half not yet this submission's) reads as "not there yet"
inline bool chunk_take(const volatile double *chunk, double tag, double *value) {
  const volatile unsigned long long *w = reinterpret_cast<const volatile unsigned long long *>(chunk);
  const unsigned long long check = w[1], vbits = w[0];
  if (check != chunk_check((unsigned long long)tag, vbits)) return false;
  std::memcpy(value, &vbits, sizeof vbits);
  return true;
}
// waits until `count` tagged chunks (k_reduce_partials) have arrived in mapped host memory and copies their values to
// ctx->fused_vals; 50 ms without them: stream synchronisation (reports a failed kernel), then failure
int wait_chunks(lrr_icp_ctx *ctx, const double *chunks, int count, double tag) {
  const volatile double *c = chunks;
  const auto t0 = std::chrono::steady_clock::now();
  bool synced = false;
  ctx->fused_vals.resize((size_t)count);
  for (int i = 0; i < count; ++i) {
    unsigned spin = 0;
    while (!chunk_take(c + 2 * i, tag, &ctx->fused_vals[(size_t)i])) {
      if ((++spin & 1023u) != 0 || std::chrono::steady_clock::now() - t0 < std::chrono::milliseconds(50)) continue;
      if (synced) return fail(ctx, LRR_ICP_ERR_HIP, "results did not arrive");
      HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
      synced = true;
    }
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  return LRR_ICP_OK;
}
inline double seq_tag(unsigned long long seq) { return (double)(seq & 0xfffffffffffffull); }  // exact in a double, never 0 (seq >= 1)

int finish_reduce(lrr_icp_ctx *ctx, int nblocks, int K, double *host_out) {
  const unsigned long long seq = ++ctx->seq;
  hipLaunchKernelGGL(k_reduce_partials, dim3(K), dim3(64), 0, ctx->stream, ctx->d_partials, nblocks, K, ctx->d_final, (const double *)nullptr, 0, ctx->h_pinned_dev, ctx->d_done, ctx->h_seq_dev, seq);
  HIPCHK(ctx, hipGetLastError());
  { int rc = wait_results(ctx, seq); if (rc) return rc; }
  std::memcpy(host_out, ctx->h_pinned, sizeof(double) * K);
  return LRR_ICP_OK;
}

// k-th values of the nj jobs in ctx->h_jobs: three asynchronous launches per kSelPack jobs, the job
// descriptors travel as kernel arguments
int launch_select_kth(lrr_icp_ctx *ctx, int nj) {
  for (int j0 = 0; j0 < nj; j0 += kSelPack) {
    int nc = std::min(kSelPack, nj - j0);
    SelPack pk;
    for (int j = 0; j < kSelPack; ++j) pk.j[j] = ctx->h_jobs[j0 + std::min(j, nc - 1)];
    hipLaunchKernelGGL(k_selw_hist, dim3(kSelHistBlocks, nc), dim3(kBlock), 0, ctx->stream, pk, ctx->d_selprev, ctx->d_selhist);
    HIPCHK(ctx, hipGetLastError());
    hipLaunchKernelGGL(k_selw_collect, dim3(kSelHistBlocks, nc), dim3(kBlock), 0, ctx->stream, pk, ctx->d_selprev, ctx->d_selhist, ctx->d_selstate, ctx->d_selcoll, ctx->d_selcursor);
    HIPCHK(ctx, hipGetLastError());
    hipLaunchKernelGGL(k_selw_final, dim3(nc), dim3(kSelThreads), 0, ctx->stream, pk, ctx->d_selprev, ctx->d_selhist, ctx->d_selstate, ctx->d_selcoll, ctx->d_selcursor);
    HIPCHK(ctx, hipGetLastError());
  }
  return LRR_ICP_OK;
}

// run up to kMaxSelJobs selection jobs; results (4 doubles per job) land in host_out
int run_select(lrr_icp_ctx *ctx, std::vector<SelJob> &jobs, double *host_out, bool need_sums = false) {
  int nj = (int)jobs.size();
  if (nj == 0) return LRR_ICP_OK;
  if (nj > kMaxSelJobs) return fail(ctx, LRR_ICP_ERR_ARG, "too many selection jobs");
  double *d_out = ctx->d_final + kMaxRed;
  for (int j = 0; j < nj; ++j) { jobs[j].out = d_out + 4 * j; ctx->h_jobs[j] = jobs[j]; }
  if (need_sums) {  // trimmed MUSE sums: single-block kernel that also accumulates the keys below the k-th
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_jobs, ctx->h_jobs, sizeof(SelJob) * nj, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_select, dim3(nj), dim3(kSelThreads), 0, ctx->stream, ctx->d_jobs);
    HIPCHK(ctx, hipGetLastError());
  } else {
    int rc = launch_select_kth(ctx, nj);
    if (rc) return rc;
  }
  HIPCHK(ctx, hipMemcpyAsync(ctx->h_pinned, d_out, sizeof(double) * 4 * nj, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  std::memcpy(host_out, ctx->h_pinned, sizeof(double) * 4 * nj);
  return LRR_ICP_OK;
}

SelJob make_job(const double *k0, int n0, const double *k1, int n1, long long k, long long m) {
  SelJob j{};
  j.k0 = k0; j.n0 = n0; j.k1 = k1; j.n1 = n1; j.k = k; j.out = nullptr;
  j.cnt = nullptr; j.cnt_a0 = j.cnt_a1 = j.cnt_b0 = j.cnt_b1 = 0; j.m_fixed = m; j.state = -1;
  j.row_a0 = j.row_a1 = j.row_b0 = j.row_b1 = 0; j.col = 0;
  return j;
}

// rrel_util_median_abs_dev_scale, dof = 1 (cdcl .txx:520-536): k-th of m values, 0 if m <= 1
inline long long mad_rank(long long m) { return (m - 1) / 2 + 1; }
inline double mad_from_kth(long long m, double kth) { return 1.4826 * (1 + 5.0 / (double)(m - 1)) * kth; }

// a new stamp for the distinct-target flags; the array is cleared when the stamp wraps around
int next_flag_gen(lrr_icp_dir *d, unsigned *gen) {
  if (++d->flag_gen == 0u) {
    HIPCHK(d->ctx, hipMemsetAsync(d->flag, 0, sizeof(unsigned) * d->flag_cap, d->ctx->stream));
    d->flag_gen = 1u;
  }
  *gen = d->flag_gen;
  return LRR_ICP_OK;
}

int ensure_dir_capacity(lrr_icp_dir *d) {
  lrr_icp_ctx *ctx = d->ctx;
  int64_t need = std::max<int64_t>(d->from->n, 1);
  if (need <= d->cap) return LRR_ICP_OK;
  void *ptrs[] = {d->slot, d->m_to, d->m_site, d->w, d->wc, d->r2, d->key_g, d->key_s, d->from_loc, d->q_loc, d->d2lb};
  for (void *p : ptrs) if (p) (void)hipFree(p);
  d->slot = d->m_to = d->m_site = nullptr; d->w = d->wc = d->r2 = d->key_g = d->key_s = d->from_loc = d->q_loc = nullptr;
  d->d2lb = nullptr;
  d->have_bounds = false;
  d->cap = 0;
  HIPCHK(ctx, hipMalloc(&d->d2lb, sizeof(float) * need));
  HIPCHK(ctx, hipMalloc(&d->slot, sizeof(int) * need));
  HIPCHK(ctx, hipMalloc(&d->m_to, sizeof(int) * need));
  HIPCHK(ctx, hipMalloc(&d->m_site, sizeof(int) * need));
  HIPCHK(ctx, hipMalloc(&d->w, sizeof(double) * need));
  HIPCHK(ctx, hipMalloc(&d->wc, sizeof(double) * need));
  HIPCHK(ctx, hipMalloc(&d->r2, sizeof(double) * need));
  HIPCHK(ctx, hipMalloc(&d->key_g, sizeof(double) * need));
  HIPCHK(ctx, hipMalloc(&d->key_s, sizeof(double) * need));
  HIPCHK(ctx, hipMalloc(&d->from_loc, sizeof(double) * 3 * need));
  HIPCHK(ctx, hipMalloc(&d->q_loc, sizeof(double) * 10 * need));  // locations (3 per source), then projector + strength (7 per source)
  d->qloc_in_step = d->qps_in_step = false;
  d->cap = need;
  return LRR_ICP_OK;
}

int ensure_lists(lrr_icp_ctx *ctx, K1Lane &L, size_t need) {
  if (need <= L.list_cap) return LRR_ICP_OK;
  if (L.d_hard) (void)hipFree(L.d_hard);
  if (L.d_active) (void)hipFree(L.d_active);
  L.d_hard = L.d_active = nullptr;
  L.list_cap = 0;
  HIPCHK(ctx, hipMalloc(&L.d_hard, sizeof(int) * need));
  HIPCHK(ctx, hipMalloc(&L.d_active, sizeof(int) * need));
  L.list_cap = need;
  return LRR_ICP_OK;
}

int perm_half(int64_t n) {
  int half = 1;
  while ((1LL << (2 * half)) < n) ++half;
  return half;
}

unsigned host_fmix32(unsigned h) {
  h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
  return h;
}

// the list counters of one K1 submission: slots are handed out round robin and all cleared together
// when the round starts over (every earlier submission has drained by then: same stream)
int fresh_counts(lrr_icp_ctx *ctx, K1Lane &L, hipStream_t st, int **counts) {
  if (L.count_slot == 0) HIPCHK(ctx, hipMemsetAsync(L.d_counts, 0, kCountSlots * 4 * sizeof(int), st));
  *counts = L.d_counts + 4 * L.count_slot;
  L.count_slot = (L.count_slot + 1) % kCountSlots;
  return LRR_ICP_OK;
}

// phase 2 of the grid search; reads its list length on the device and falls through when it is 0
HardArgs hard_args(K1Lane &L, const lrr_icp_cloud *to, const GridView &g, const QSrc &q, const XfD &xf, int *m_to, int *counts,
                   float *d2lb = nullptr) {
  HardArgs a;
  a.d2lb = d2lb;
  a.stats = nullptr; a.seen = 0; a.listed = counts + 2;
  a.g = g; a.q = q; a.xf = xf;
  a.hard = L.d_hard; a.hard_count = counts;
  a.use_prev = q.src ? 1 : 0;
  a.to_pt = to->cpt;
  a.m_to = m_to; a.overflow_count = counts + 1;
  a.m_feat = nullptr; a.s2f = nullptr;
  return a;
}
// blocks of the ball kernels: one wavefront each, rounded up to a multiple of 8 for the XCD mapping
inline unsigned ball_blocks_for(int64_t n) { return (unsigned)((((n + kBallBlock - 1) / kBallBlock) + 7) / 8 * 8); }
inline int hard_blocks_for(int64_t n) { return (int)std::min<int64_t>(1024, std::max<int64_t>(1, (n + 3) / 4)); }

// phase 2 of the grid search: the wave-per-query kernel (a query its ring walk cannot settle is scanned against every
// target by the same wavefront, wave_scan_all: no third phase)
int launch_hard_phases(lrr_icp_ctx *ctx, K1Lane &L, hipStream_t st, const lrr_icp_cloud *to, const GridView &g, const QSrc &q, const XfD &xf, int n, int *m_to, int *counts, bool /*all_have_prev*/,
                       float *d2lb = nullptr, bool cert = false, bool short_list = false) {
  HardArgs h = hard_args(L, to, g, q, xf, m_to, counts, d2lb);
  if (cert && ctx->prof) { h.stats = ctx->d_cert_stats; h.seen = (unsigned long long)n; }
  if (short_list) { h.hard = L.d_active; h.hard_count = counts + 2; }  // the certificate kernel's list itself
  hipLaunchKernelGGL(k_match_hard, dim3(hard_blocks_for(n)), dim3(kBlock), 0, st, h);
  HIPCHK(ctx, hipGetLastError());
  return LRR_ICP_OK;
}

int launch_brute(lrr_icp_ctx *ctx, K1Lane &L, hipStream_t st, const lrr_icp_cloud *to, const GridView &g, const QSrc &q, const int *active,
                 const XfD &xf, int n, int *m_to) {
  int bx = (n + kBlock * kBruteQ - 1) / (kBlock * kBruteQ);
  int S = std::max(1, std::min(64, 1024 / bx));
  int max_chunks = (int)((to->n + kBruteTile - 1) / kBruteTile);
  S = std::min(S, std::max(1, max_chunks));
  int chunk = (int)((to->n + S - 1) / S);
  chunk = ((chunk + kBruteTile - 1) / kBruteTile) * kBruteTile;
  S = (int)((to->n + chunk - 1) / chunk);
  size_t need = (size_t)S * (size_t)n;
  if (need > L.brute_cap) {
    if (L.d_bd) (void)hipFree(L.d_bd);
    if (L.d_bi) (void)hipFree(L.d_bi);
    L.d_bd = nullptr; L.d_bi = nullptr;
    L.brute_cap = 0;
    HIPCHK(ctx, hipMalloc(&L.d_bd, sizeof(double) * need));
    HIPCHK(ctx, hipMalloc(&L.d_bi, sizeof(int) * 2 * need));
    L.brute_cap = need;
  }
  int *po = L.d_bi, *pc = L.d_bi + need;
  hipLaunchKernelGGL(k_match_brute, dim3(bx, S), dim3(kBlock), 0, st, g, q, active, xf, n, chunk, L.d_bd, po, pc);
  HIPCHK(ctx, hipGetLastError());
  hipLaunchKernelGGL(k_brute_merge, dim3(blocks_for(n)), dim3(kBlock), 0, st, L.d_bd, po, pc, n, S, q, active, m_to);
  HIPCHK(ctx, hipGetLastError());
  return LRR_ICP_OK;
}

struct ProfScope {  // HIP events around K1 submissions (all phases), recorded on the stream that runs them
  lrr_icp_ctx *ctx;
  hipStream_t st;
  int launches;
  bool on;
  bool attached = false;  // the events were handed to the launch itself (attach): no marker packets
  ProfScope(lrr_icp_ctx *c, hipStream_t stream, int n_launches)
      : ctx(c), st(stream), launches(n_launches),
        on(n_launches > 0 && c->prof && c->pe_used + 2 <= (int)c->pe.size() && (c->prof_tick++ % c->prof_every) == 0) {}
  // A submission that is ONE kernel gives its two events to the launch (hipExtLaunchKernelGGL): they then carry the
  // kernel's own start and end timestamps — what rocprofv3 reports — instead of the completion times of two marker
  // packets around it (which add ~2 us to a 9-us kernel).  Call before the launch; returns the event pair or nulls.
  void attach(hipEvent_t *start, hipEvent_t *stop) {
    *start = *stop = nullptr;
    if (!on) return;
    attached = true;
    *start = ctx->pe[ctx->pe_used];
    *stop = ctx->pe[ctx->pe_used + 1];
  }
  void begin() { if (on && !attached) (void)hipEventRecord(ctx->pe[ctx->pe_used], st); }
  ~ProfScope() {
    if (!on) return;
    if (!attached) (void)hipEventRecord(ctx->pe[ctx->pe_used + 1], st);
    ctx->pe_used += 2;
    ctx->match_launches += launches;  // timed submissions only: the mean stays a mean when the event pool runs out
  }
};

int resolve_profile(lrr_icp_ctx *ctx) {
  if (ctx->pe_used == 0) return LRR_ICP_OK;
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  for (int i = 0; i + 1 < ctx->pe_used; i += 2) {
    float ms = 0.f;
    HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->pe[i], ctx->pe[i + 1]));
    ctx->match_ms += ms;
    if (ctx->match_samples.size() < (size_t)1 << 20) ctx->match_samples.push_back(ms);
  }
  ctx->pe_used = 0;
  return LRR_ICP_OK;
}

// a residual pass met a match index outside the to-cloud (checked_match): clears the word, fails the call
int match_fault(lrr_icp_ctx *ctx, const char *who) {
  (void)hipMemsetAsync(ctx->d_fault, 0, sizeof(int), ctx->stream);
  return fail(ctx, LRR_ICP_ERR_STATE, std::string(who) + ": a match index is outside the to-cloud (a K1 phase left an active source without a match)");
}

// certificate statistics of a search whose kernel had no wave-per-query kernel behind it to add them on the device
void note_inline_cert(lrr_icp_ctx *ctx, lrr_icp_dir *d) {
  if (!d->cert_inline) return;
  d->cert_inline = false;
  if (!ctx->prof || d->last_list_count < 0) return;
  ctx->cert_host_seen += d->from->n;
  ctx->cert_host_listed += d->last_list_count;
}

DirArgs dir_args(lrr_icp_dir *d) {
  DirArgs a;
  a.ps = pairs_of(d);
  a.n = (int)d->n;
  a.w = d->w; a.wc = d->wc; a.r2 = d->r2; a.key_s = d->key_s; a.key_g = d->key_g; a.q_loc = d->q_loc;
  a.flag = d->flag; a.gen = d->flag_gen;
  a.float_product = d->from->float_strength;
  a.nt = (int)d->to->n; a.fault = d->ctx->d_fault;
  return a;
}

}  // namespace

// =============================================================================================
// C-ABI
// =============================================================================================
extern "C" {

int lrr_icp_create(int device, lrr_icp_ctx **out) {
  if (!out) return LRR_ICP_ERR_ARG;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) return LRR_ICP_ERR_HIP;
  lrr_icp_ctx *ctx = new (std::nothrow) lrr_icp_ctx;
  if (!ctx) return LRR_ICP_ERR_HIP;
  ctx->device = device;
  auto bail = [&](hipError_t e) { fprintf(stderr, "lrr_icp_create: %s\n", hipGetErrorString(e)); lrr_icp_destroy(ctx); return LRR_ICP_ERR_HIP; };
  hipError_t e;
  if ((e = hipSetDevice(device)) != hipSuccess) return bail(e);
  if ((e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess) return bail(e);
  if ((e = hipEventCreate(&ctx->ev0)) != hipSuccess) return bail(e);
  if ((e = hipEventCreate(&ctx->ev1)) != hipSuccess) return bail(e);
  if ((e = hipMalloc(&ctx->d_partials, sizeof(double) * kMaxBlocks * kMaxRed)) != hipSuccess) return bail(e);
  if ((e = hipMalloc(&ctx->d_final, sizeof(double) * (kMaxRed + 4 * kMaxSelJobs))) != hipSuccess) return bail(e);
  if ((e = hipHostMalloc(&ctx->h_pinned, sizeof(double) * (kMaxRed + 4 * kMaxSelJobs), hipHostMallocMapped | hipHostMallocCoherent)) != hipSuccess) return bail(e);
  if ((e = hipHostGetDevicePointer((void **)&ctx->h_pinned_dev, ctx->h_pinned, 0)) != hipSuccess) return bail(e);
  if ((e = hipMalloc(&ctx->d_jobs, sizeof(SelJob) * kMaxSelJobs)) != hipSuccess) return bail(e);
  if ((e = hipHostMalloc(&ctx->h_jobs, sizeof(SelJob) * kMaxSelJobs)) != hipSuccess) return bail(e);
  ctx->lane[0].stream = ctx->stream;
  if ((e = hipStreamCreateWithFlags(&ctx->lane[1].stream, hipStreamNonBlocking)) != hipSuccess) return bail(e);
  if ((e = hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming)) != hipSuccess) return bail(e);
  if ((e = hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming)) != hipSuccess) return bail(e);
  for (K1Lane &L : ctx->lane)
    if ((e = hipMalloc(&L.d_counts, kCountSlots * 4 * sizeof(int))) != hipSuccess) return bail(e);
  if ((e = hipMalloc(&ctx->d_fused, sizeof(double) * kFusedDoubles)) != hipSuccess) return bail(e);
  if ((e = hipHostMalloc(&ctx->h_fused, sizeof(double) * 512, hipHostMallocMapped | hipHostMallocCoherent)) != hipSuccess) return bail(e);
  std::memset(ctx->h_fused, 0, sizeof(double) * 512);  // no chunk carries a valid tag yet
  if ((e = hipHostGetDevicePointer((void **)&ctx->h_fused_dev, ctx->h_fused, 0)) != hipSuccess) return bail(e);
  if ((e = hipHostMalloc(&ctx->h_rows, sizeof(double) * 8 * kWideBlocks, hipHostMallocMapped | hipHostMallocCoherent)) != hipSuccess) return bail(e);
  std::memset(ctx->h_rows, 0, sizeof(double) * 8 * kWideBlocks);
  if ((e = hipHostGetDevicePointer((void **)&ctx->h_rows_dev, ctx->h_rows, 0)) != hipSuccess) return bail(e);
  if ((e = hipHostMalloc(&ctx->h_seq, 64, hipHostMallocMapped | hipHostMallocCoherent)) != hipSuccess) return bail(e);
  *ctx->h_seq = 0;
  if ((e = hipHostGetDevicePointer((void **)&ctx->h_seq_dev, ctx->h_seq, 0)) != hipSuccess) return bail(e);
  if ((e = hipMalloc(&ctx->d_done, sizeof(int))) != hipSuccess) return bail(e);
  if ((e = hipMemset(ctx->d_done, 0, sizeof(int))) != hipSuccess) return bail(e);
  if ((e = hipMalloc(&ctx->d_fault, sizeof(int))) != hipSuccess) return bail(e);
  if ((e = hipMemset(ctx->d_fault, 0, sizeof(int))) != hipSuccess) return bail(e);
  if ((e = hipMalloc(&ctx->d_cert_stats, 2 * sizeof(unsigned long long))) != hipSuccess) return bail(e);
  if ((e = hipMemset(ctx->d_cert_stats, 0, 2 * sizeof(unsigned long long))) != hipSuccess) return bail(e);
  if ((e = hipMalloc(&ctx->d_selstate, sizeof(SelState) * kMaxSelJobs)) != hipSuccess) return bail(e);
  if ((e = hipMalloc(&ctx->d_selhist, sizeof(int) * kMaxSelJobs * 2048)) != hipSuccess) return bail(e);
  if ((e = hipMemset(ctx->d_selhist, 0, sizeof(int) * kMaxSelJobs * 2048)) != hipSuccess) return bail(e);
  if ((e = hipMalloc(&ctx->d_selprev, sizeof(double) * kSelStates)) != hipSuccess) return bail(e);
  if ((e = hipMemset(ctx->d_selprev, 0, sizeof(double) * kSelStates)) != hipSuccess) return bail(e);
  if ((e = hipMalloc(&ctx->d_selwin, sizeof(double) * kSelStates)) != hipSuccess) return bail(e);
  {
    double w0[kSelStates];
    for (int k = 0; k < kSelStates; ++k) w0[k] = 1.0 / 256.0;
    if ((e = hipMemcpy(ctx->d_selwin, w0, sizeof w0, hipMemcpyHostToDevice)) != hipSuccess) return bail(e);
  }
  if ((e = hipMalloc(&ctx->d_selcoll, sizeof(unsigned long long) * kMaxSelJobs * kSelCap)) != hipSuccess) return bail(e);
  if ((e = hipMalloc(&ctx->d_selcursor, sizeof(int) * kMaxSelJobs)) != hipSuccess) return bail(e);
  if ((e = hipMemset(ctx->d_selcursor, 0, sizeof(int) * kMaxSelJobs)) != hipSuccess) return bail(e);
  *out = ctx;
  return LRR_ICP_OK;
}

void lrr_icp_destroy(lrr_icp_ctx *ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  if (ctx->lane[1].stream) (void)hipStreamSynchronize(ctx->lane[1].stream);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  for (hipEvent_t ev : ctx->pe) (void)hipEventDestroy(ev);
  if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
  if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
  for (K1Lane &L : ctx->lane) {
    void *ls[] = {L.d_bd, L.d_bi, L.d_hard, L.d_active, L.d_counts, L.d_order};
    for (void *p : ls) if (p) (void)hipFree(p);
  }
  if (ctx->lane[1].stream) (void)hipStreamDestroy(ctx->lane[1].stream);
  if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
  if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
  void *dev[] = {ctx->d_partials, ctx->d_final, ctx->d_jobs, ctx->d_fused, ctx->d_selstate, ctx->d_selhist,
                 ctx->d_selprev, ctx->d_selcoll, ctx->d_selcursor, ctx->d_selwin};
  for (void *p : dev) if (p) (void)hipFree(p);
  if (ctx->h_pinned) (void)hipHostFree(ctx->h_pinned);
  if (ctx->h_jobs) (void)hipHostFree(ctx->h_jobs);
  if (ctx->h_fused) (void)hipHostFree(ctx->h_fused);
  if (ctx->h_seq) (void)hipHostFree(ctx->h_seq);
  if (ctx->h_rows) (void)hipHostFree(ctx->h_rows);
  if (ctx->d_muse) (void)hipFree(ctx->d_muse);
  if (ctx->d_done) (void)hipFree(ctx->d_done);
  if (ctx->d_cert_stats) (void)hipFree(ctx->d_cert_stats);
  if (ctx->d_fault) (void)hipFree(ctx->d_fault);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

const char *lrr_icp_last_error(const lrr_icp_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int lrr_icp_synchronize(lrr_icp_ctx *ctx) {
  if (!ctx) return LRR_ICP_ERR_ARG;
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return LRR_ICP_OK;
}

int lrr_icp_timer_start(lrr_icp_ctx *ctx) {
  if (!ctx) return LRR_ICP_ERR_ARG;
  HIPCHK(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  return LRR_ICP_OK;
}

int lrr_icp_timer_stop(lrr_icp_ctx *ctx, double *elapsed_ms) {
  if (!ctx || !elapsed_ms) return LRR_ICP_ERR_ARG;
  HIPCHK(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  HIPCHK(ctx, hipEventSynchronize(ctx->ev1));
  float ms = 0.f;
  HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
  *elapsed_ms = ms;
  return LRR_ICP_OK;
}

int lrr_icp_match_profile(lrr_icp_ctx *ctx, int enable, double *total_ms, int64_t *launches, int reset) {
  if (!ctx) return LRR_ICP_ERR_ARG;
  int rc = resolve_profile(ctx);
  if (rc) return rc;
  if (total_ms) *total_ms = ctx->match_ms;
  if (launches) *launches = ctx->match_launches;
  if (reset) { ctx->match_ms = 0.0; ctx->match_launches = 0; ctx->match_samples.clear(); }
  if (enable && ctx->pe.empty()) {
    ctx->pe.resize(kProfEvents);
    for (auto &ev : ctx->pe) HIPCHK(ctx, hipEventCreate(&ev));
  }
  ctx->prof = enable != 0;
  ctx->prof_every = enable > 1 ? (unsigned)enable : 1u;  // enable = N > 1: every N-th submission is timed (two event records cost a few us)
  ctx->prof_tick = 0;
  return LRR_ICP_OK;
}

// the individual timed K1 submissions behind lrr_icp_match_profile's total (each = the launch(es) serving its directions)
int lrr_icp_match_profile_samples(lrr_icp_ctx *ctx, double *ms, int64_t cap, int64_t *count) {
  if (!ctx || !count || cap < 0 || (cap > 0 && !ms)) return LRR_ICP_ERR_ARG;
  int rc = resolve_profile(ctx);
  if (rc) return rc;
  *count = (int64_t)ctx->match_samples.size();
  for (int64_t i = 0; i < cap && i < *count; ++i) ms[i] = (double)ctx->match_samples[(size_t)i];
  return LRR_ICP_OK;
}

// what a pair of event records with NOTHING between them measures on the context's stream (the two marker packets
// themselves): the part of a timed K1 submission that is not kernel time
int lrr_icp_event_overhead(lrr_icp_ctx *ctx, int pairs, double *ms_per_pair) {
  if (!ctx || pairs <= 0 || !ms_per_pair) return LRR_ICP_ERR_ARG;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  std::vector<hipEvent_t> ev(2 * (size_t)pairs);
  for (auto &e : ev) HIPCHK(ctx, hipEventCreate(&e));
  for (int i = 0; i < pairs; ++i) {
    // a trivial launch in front, as in the timed loop where other kernels precede the submission
    hipLaunchKernelGGL(k_copy_w, dim3(1), dim3(64), 0, ctx->stream, (const double *)ctx->d_final, ctx->d_final, 0);
    (void)hipEventRecord(ev[2 * i], ctx->stream);
    (void)hipEventRecord(ev[2 * i + 1], ctx->stream);
  }
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  double tot = 0.0;
  for (int i = 0; i < pairs; ++i) { float ms = 0.f; HIPCHK(ctx, hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1])); tot += ms; }
  for (auto &e : ev) (void)hipEventDestroy(e);
  *ms_per_pair = tot / pairs;
  return LRR_ICP_OK;
}

int lrr_icp_k1_path_counts(lrr_icp_ctx *ctx, int64_t counts[LRR_ICP_K1_PATHS], int reset) {
  if (!ctx || !counts) return LRR_ICP_ERR_ARG;
  for (int k = 0; k < LRR_ICP_K1_PATHS; ++k) counts[k] = ctx->k1_paths[k];
  if (reset) for (int k = 0; k < LRR_ICP_K1_PATHS; ++k) ctx->k1_paths[k] = 0;
  return LRR_ICP_OK;
}

int lrr_icp_cert_stats(lrr_icp_ctx *ctx, int64_t *queries, int64_t *searched, int reset) {
  if (!ctx) return LRR_ICP_ERR_ARG;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  unsigned long long v[2] = {0, 0};
  HIPCHK(ctx, hipMemcpy(v, ctx->d_cert_stats, sizeof v, hipMemcpyDeviceToHost));
  if (queries) *queries = (int64_t)v[0] + ctx->cert_host_seen;
  if (searched) *searched = (int64_t)v[1] + ctx->cert_host_listed;
  if (reset) { HIPCHK(ctx, hipMemset(ctx->d_cert_stats, 0, sizeof v)); ctx->cert_host_seen = ctx->cert_host_listed = 0; }
  return LRR_ICP_OK;
}

// ---- clouds ----------------------------------------------------------------------------------
int lrr_icp_cloud_create(lrr_icp_ctx *ctx, int64_t n, const double *loc, const double *proj,
                         const double *strength, int float_strength, lrr_icp_cloud **out) {
  if (!ctx || !out || n < 0 || n > 0x3fffffff || (n > 0 && (!loc || !proj))) return fail(ctx, LRR_ICP_ERR_ARG, "cloud_create: bad argument");
  *out = nullptr;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  // the error projector must be symmetric (nn^T, I - tt^T, I: features/itkFeatureImageFilter.txx:1053-1072)
  for (int64_t i = 0; i < n; ++i) {
    const double *P = proj + 9 * i;
    if (P[1] != P[3] || P[2] != P[6] || P[5] != P[7]) return fail(ctx, LRR_ICP_ERR_ARG, "cloud_create: error_projector_ must be symmetric");
    for (int d = 0; d < 3; ++d)
      if (!std::isfinite(loc[3 * i + d])) return fail(ctx, LRR_ICP_ERR_ARG, "cloud_create: non-finite location");
  }
  lrr_icp_cloud *c = new (std::nothrow) lrr_icp_cloud;
  if (!c) return fail(ctx, LRR_ICP_ERR_HIP, "out of host memory");
  c->ctx = ctx; c->n = n; c->float_strength = float_strength;
  // uniform grid, ~0.5 points per cell on average (features sit on surfaces: occupied cells hold
  // far more); replaces the kd-tree / Voronoi volume of the reference
  double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
  for (int64_t i = 0; i < n; ++i)
    for (int d = 0; d < 3; ++d) {
      double v = loc[3 * i + d];
      if (i == 0 || v < lo[d]) lo[d] = v;
      if (i == 0 || v > hi[d]) hi[d] = v;
    }
  double ext[3], maxext = 0.0, maxabs = 0.0;
  for (int d = 0; d < 3; ++d) {
    ext[d] = hi[d] - lo[d];
    maxext = std::max(maxext, ext[d]);
    maxabs = std::max(maxabs, std::max(std::fabs(lo[d]), std::fabs(hi[d])));
  }
  double h = 1.0;
  if (n > 0 && maxext > 0.0) {
    double vol = 1.0;
    for (int d = 0; d < 3; ++d) vol *= std::max(ext[d], maxext * 1e-3);
    double ppc = 0.5;  // target points per cell
    if (const char *e = std::getenv("LRR_ICP_GRID_PPC")) { double v = std::atof(e); if (v > 0.0) ppc = v; }
    h = std::cbrt(vol / std::max<double>(1.0, (double)n / ppc));
    h = std::max(h, maxext / 512.0);
  }
  int dim[3];
  for (int d = 0; d < 3; ++d) dim[d] = (int)std::min<double>(512.0, std::floor(ext[d] / h) + 1.0);
  int64_t ncell = (int64_t)dim[0] * dim[1] * dim[2];
  c->h = h; c->inv_h = 1.0 / h;
  for (int d = 0; d < 3; ++d) { c->lo[d] = lo[d]; c->dim[d] = dim[d]; }
  c->tol = 1e-9 * (maxabs + maxext + h);
  const size_t m = (size_t)std::max<int64_t>(n, 1);
  // x-slices per cell: rows stay h x h wide in (y,z) (few rows per ball) while the x-run of a ball
  // is cut at h/fine resolution (few candidates per row)
  int fine = 8;
  if (const char *e = std::getenv("LRR_ICP_GRID_FINE")) { int v = std::atoi(e); if (v >= 1 && v <= 64) { fine = 1; while (fine * 2 <= v) fine *= 2; } }
  while (fine > 1 && ncell * fine > (int64_t)1 << 25) fine >>= 1;
  c->fine = fine;
  // The grid is built on the device: the raw arrays go up once, a counting sort by x-slice gives the raster
  // order (the search structure), a second count by (tile, cell) the canonical order; ranks inside a slice /
  // cell are by original index, so the result does not depend on the order the atomics land in.
  const int tdx = (dim[0] + kTile - 1) / kTile, tdy = (dim[1] + kTile - 1) / kTile, tdz = (dim[2] + kTile - 1) / kTile;
  const long long ntile = (long long)tdx * tdy * tdz, nccell = ntile * (kTile * kTile * kTile);
  const long long nfine = (long long)ncell * fine;
  auto cleanup = [&](int code, const std::string &msg) { lrr_icp_cloud_destroy(c); return fail(ctx, code, msg); };
  struct Part { void **dst; size_t bytes, off; };
  Chunk *d_chunks_dummy = nullptr;
  (void)d_chunks_dummy;
  const size_t max_chunks = (size_t)(n / kBlock + ntile + 1);
  Part parts[] = {
      {(void **)&c->rec, sizeof(double) * m * kRecStride, 0},
      {(void **)&c->src, sizeof(SrcPoint) * m, 0},
      {(void **)&c->gp, sizeof(GridPoint) * m, 0},
      {(void **)&c->cpt, sizeof(GridPoint) * m, 0},
      {(void **)&c->gpf, sizeof(float4) * m, 0},
      {(void **)&c->str, sizeof(double) * m, 0},
      {(void **)&c->loc_orig, sizeof(double) * 3 * m, 0},
      {(void **)&c->shp, sizeof(ShapeRec) * m, 0},
      {(void **)&c->orig_of, sizeof(int) * m, 0},
      {(void **)&c->canon_of, sizeof(int) * m, 0},
      {(void **)&c->cell_start, sizeof(int) * ((size_t)ncell + 1), 0},
      {(void **)&c->fine_start, sizeof(int) * ((size_t)nfine + 1), 0},
      {(void **)&c->chunks, sizeof(Chunk) * max_chunks, 0},
  };
  size_t total = 0;
  for (Part &pt : parts) { pt.off = total; total += (pt.bytes + 255) & ~(size_t)255; }
  if (hipMalloc(&c->slab, total) != hipSuccess) return cleanup(LRR_ICP_ERR_HIP, "cloud_create: hipMalloc failed");
  for (const Part &pt : parts) *pt.dst = (unsigned char *)c->slab + pt.off;
  hipStream_t st = ctx->stream;
  // build scratch: projectors, strengths, slice of every feature, raster permutation, cursors, canonical cell counts, scan sums
  const size_t nsums = (size_t)(std::max(nfine, nccell) / kScanItems + 2);
  const size_t scratch_bytes = sizeof(double) * (9 * m + m) + sizeof(int) * (2 * m + (size_t)nfine + 1 + (size_t)nccell + 1 + nsums);
  unsigned char *scratch = nullptr;
  if (hipMalloc(&scratch, scratch_bytes) != hipSuccess) return cleanup(LRR_ICP_ERR_HIP, "cloud_create: hipMalloc failed");
  double *d_proj = (double *)scratch, *d_str = d_proj + 9 * m;
  int *d_slice = (int *)(d_str + m), *d_raster = d_slice + m, *d_cursor = d_raster + m, *d_cstart = d_cursor + (size_t)nfine + 1, *d_sums = d_cstart + (size_t)nccell + 1;
  auto bail = [&](const char *what) { (void)hipStreamSynchronize(st); (void)hipFree(scratch); return cleanup(LRR_ICP_ERR_HIP, std::string("cloud_create: ") + what); };
  bool ok = true;
  if (n > 0) {
    ok = ok && hipMemcpyAsync(c->loc_orig, loc, sizeof(double) * 3 * (size_t)n, hipMemcpyHostToDevice, st) == hipSuccess;
    ok = ok && hipMemcpyAsync(d_proj, proj, sizeof(double) * 9 * (size_t)n, hipMemcpyHostToDevice, st) == hipSuccess;
    if (strength) ok = ok && hipMemcpyAsync(d_str, strength, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, st) == hipSuccess;
  }
  ok = ok && hipMemsetAsync(c->fine_start, 0, sizeof(int) * ((size_t)nfine + 1), st) == hipSuccess;
  ok = ok && hipMemsetAsync(d_cursor, 0, sizeof(int) * ((size_t)nfine + 1), st) == hipSuccess;
  ok = ok && hipMemsetAsync(d_cstart, 0, sizeof(int) * ((size_t)nccell + 1), st) == hipSuccess;
  if (!ok) return bail("upload failed");
  GridGeom G;
  G.lox = lo[0]; G.loy = lo[1]; G.loz = lo[2]; G.inv_h = c->inv_h;
  G.dx = dim[0]; G.dy = dim[1]; G.dz = dim[2]; G.fine = fine; G.tdx = tdx; G.tdy = tdy;
  std::vector<int> tile_start((size_t)ntile + 1, 0);
  if (n > 0) {
    const int nbn = blocks_for(n);
    hipLaunchKernelGGL(k_g_count, dim3(nbn), dim3(kBlock), 0, st, c->loc_orig, (int)n, G, d_slice, c->fine_start, d_cstart);
    if (device_scan(ctx, st, c->fine_start, nfine + 1, d_sums) || device_scan(ctx, st, d_cstart, nccell + 1, d_sums)) return bail("scan failed");
    hipLaunchKernelGGL(k_g_cell_start, dim3((unsigned)((ncell + 1 + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, c->fine_start, fine, (long long)ncell, c->cell_start);
    hipLaunchKernelGGL(k_g_scatter, dim3(nbn), dim3(kBlock), 0, st, d_slice, (int)n, c->fine_start, d_cursor, d_raster);
    hipLaunchKernelGGL(k_g_place, dim3(nbn), dim3(kBlock), 0, st, d_slice, d_raster, (int)n, G, c->fine_start, d_cstart, c->loc_orig, c->canon_of, c->orig_of, c->gp, c->gpf);
    hipLaunchKernelGGL(k_g_fill, dim3(nbn), dim3(kBlock), 0, st, c->orig_of, (int)n, c->loc_orig, d_proj, strength ? d_str : (const double *)nullptr, float_strength, c->rec, c->src, c->str, c->cpt);
    if (hipGetLastError() != hipSuccess) return bail("grid build launch failed");
    // the chunk table of the tile kernel (runs of <= kBlock canonical features of one tile) from the tile starts
    if (hipMemcpy2DAsync(tile_start.data(), sizeof(int), d_cstart, sizeof(int) * (kTile * kTile * kTile), sizeof(int), (size_t)ntile + 1, hipMemcpyDeviceToHost, st) != hipSuccess)
      return bail("download failed");
  } else {
    if (hipMemsetAsync(c->cell_start, 0, sizeof(int) * ((size_t)ncell + 1), st) != hipSuccess) return bail("memset failed");
  }
  c->h_orig_of.assign(m, 0);
  if (n > 0 && hipMemcpyAsync(c->h_orig_of.data(), c->orig_of, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, st) != hipSuccess) return bail("download failed");
  if (hipStreamSynchronize(st) != hipSuccess) return bail("grid build failed");
  std::vector<Chunk> chunks;
  for (long long t = 0; t < ntile; ++t)
    for (int s0 = tile_start[(size_t)t]; s0 < tile_start[(size_t)t + 1]; s0 += kBlock)
      chunks.push_back(Chunk{s0, std::min(kBlock, tile_start[(size_t)t + 1] - s0)});
  c->nchunks = (int)chunks.size();
  if (chunks.size() > max_chunks) return bail("chunk table overflow");
  if (!chunks.empty() && hipMemcpy(c->chunks, chunks.data(), sizeof(Chunk) * chunks.size(), hipMemcpyHostToDevice) != hipSuccess) return bail("upload failed");
  (void)hipFree(scratch);
  *out = c;
  return LRR_ICP_OK;
}

void lrr_icp_cloud_destroy(lrr_icp_cloud *c) {
  if (!c) return;
  if (c->ctx) (void)hipSetDevice(c->ctx->device);
  if (c->vor) lrr_icp_cloud_destroy(c->vor);
  if (c->vor_site_to_feature) (void)hipFree(c->vor_site_to_feature);
  if (c->cov) (void)hipFree(c->cov);
  if (c->slab) (void)hipFree(c->slab);
  delete c;
}

// The occupied voxels of the 1 mm index volume rrl/examples/compute_distance_map.cxx:55-151 builds for a feature
// set: origin = minimum corner of the bounding box, size = ceil(extent), feature i > 0 writes i into the voxel its
// location rounds to (a later feature overwrites an earlier one; 0 is the background value, so feature 0 is never
// stored).  The volume itself (nearest occupied voxel of every voxel) is not materialised: a lookup is the exact
// nearest occupied voxel of the query's voxel, found with the grid search over the site cloud, ties to the lowest
// feature index.  Same choices as the CPU checker of the tests : rounding half up, exact distances
// where the reference uses Danielsson's approximation — parity with the reference binary unpinned.
int lrr_icp_cloud_build_voronoi(lrr_icp_cloud *c) {
  if (!c) return LRR_ICP_ERR_ARG;
  lrr_icp_ctx *ctx = c->ctx;
  if (c->vor) return LRR_ICP_OK;
  if (c->n <= 0) return fail(ctx, LRR_ICP_ERR_ARG, "build_voronoi: empty cloud");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const int64_t n = c->n;
  std::vector<double> loc(3 * (size_t)n);
  HIPCHK(ctx, hipMemcpy(loc.data(), c->loc_orig, sizeof(double) * 3 * (size_t)n, hipMemcpyDeviceToHost));
  double mx[3];
  for (int d = 0; d < 3; ++d) { c->vor_origin[d] = loc[d]; mx[d] = loc[d]; }
  for (int64_t i = 0; i < n; ++i)
    for (int d = 0; d < 3; ++d) {
      c->vor_origin[d] = std::min(c->vor_origin[d], loc[3 * i + d]);
      mx[d] = std::max(mx[d], loc[3 * i + d]);
    }
  for (int d = 0; d < 3; ++d) {
    double sz = std::ceil(mx[d] - c->vor_origin[d]);
    if (sz < 1.0) sz = 1.0;
    if (sz > 2.0e9) return fail(ctx, LRR_ICP_ERR_ARG, "build_voronoi: extent too large for a 1 mm volume");
    c->vor_size[d] = (int)sz;
  }
  auto vindex = [&](int d, double x) {
    double v = std::floor((x - c->vor_origin[d]) + 0.5);
    if (v < 0.0) v = 0.0;
    if (v > (double)(c->vor_size[d] - 1)) v = (double)(c->vor_size[d] - 1);
    return (int64_t)v;
  };
  std::vector<std::pair<int64_t, int64_t>> pairs;  // (voxel, feature)
  pairs.reserve((size_t)n);
  for (int64_t i = 1; i < n; ++i) {
    const int64_t ix = vindex(0, loc[3 * i]), iy = vindex(1, loc[3 * i + 1]), iz = vindex(2, loc[3 * i + 2]);
    pairs.push_back({(iz * c->vor_size[1] + iy) * (int64_t)c->vor_size[0] + ix, i});
  }
  if (pairs.empty()) {  // only feature 0: every lookup reads the background value 0
    pairs.push_back({(vindex(2, loc[2]) * c->vor_size[1] + vindex(1, loc[1])) * (int64_t)c->vor_size[0] + vindex(0, loc[0]), 0});
  }
  std::sort(pairs.begin(), pairs.end());
  std::vector<std::pair<int64_t, int64_t>> sites;  // (feature, voxel): the last writer of every voxel, ascending feature
  for (size_t k = 0; k < pairs.size(); ++k)
    if (k + 1 == pairs.size() || pairs[k + 1].first != pairs[k].first) sites.push_back({pairs[k].second, pairs[k].first});
  std::sort(sites.begin(), sites.end());
  const int64_t ns = (int64_t)sites.size();
  std::vector<double> sloc(3 * (size_t)ns), sproj(9 * (size_t)ns, 0.0);
  for (int64_t k = 0; k < ns; ++k) {
    const int64_t v = sites[(size_t)k].second;
    const int64_t ix = v % c->vor_size[0], iy = (v / c->vor_size[0]) % c->vor_size[1], iz = v / ((int64_t)c->vor_size[0] * c->vor_size[1]);
    sloc[3 * k] = c->vor_origin[0] + (double)ix;
    sloc[3 * k + 1] = c->vor_origin[1] + (double)iy;
    sloc[3 * k + 2] = c->vor_origin[2] + (double)iz;
    sproj[9 * k] = sproj[9 * k + 4] = sproj[9 * k + 8] = 1.0;
  }
  lrr_icp_cloud *sc = nullptr;
  int rc = lrr_icp_cloud_create(ctx, ns, sloc.data(), sproj.data(), nullptr, 0, &sc);
  if (rc) return rc;
  // site canonical index -> canonical index (in this cloud) of the feature it stands for; the site cloud's own
  // "original index" is the site number, ascending in the feature index: its tie rule is the lowest feature index
  std::vector<int> canon_of((size_t)n), s2f((size_t)ns);
  for (int64_t s2 = 0; s2 < n; ++s2) canon_of[(size_t)c->h_orig_of[(size_t)s2]] = (int)s2;
  for (int64_t sc_i = 0; sc_i < ns; ++sc_i) s2f[(size_t)sc_i] = canon_of[(size_t)sites[(size_t)sc->h_orig_of[(size_t)sc_i]].first];
  if (hipMalloc(&c->vor_site_to_feature, sizeof(int) * (size_t)ns) != hipSuccess ||
      hipMemcpy(c->vor_site_to_feature, s2f.data(), sizeof(int) * (size_t)ns, hipMemcpyHostToDevice) != hipSuccess) {
    lrr_icp_cloud_destroy(sc);
    return fail(ctx, LRR_ICP_ERR_HIP, "build_voronoi: upload failed");
  }
  c->h_vor_site_to_orig.resize((size_t)ns);
  for (int64_t sc_i = 0; sc_i < ns; ++sc_i) c->h_vor_site_to_orig[(size_t)sc_i] = (int)sites[(size_t)sc->h_orig_of[(size_t)sc_i]].first;
  c->vor = sc;
  return LRR_ICP_OK;
}

int64_t lrr_icp_cloud_size(const lrr_icp_cloud *c) { return c ? c->n : 0; }

// ---- directions ------------------------------------------------------------------------------
int lrr_icp_dir_create(lrr_icp_ctx *ctx, lrr_icp_cloud *from, lrr_icp_cloud *to, lrr_icp_dir **out) {
  if (!ctx || !from || !to || !out) return fail(ctx, LRR_ICP_ERR_ARG, "dir_create: bad argument");
  lrr_icp_dir *d = new (std::nothrow) lrr_icp_dir;
  if (!d) return fail(ctx, LRR_ICP_ERR_HIP, "out of host memory");
  d->ctx = ctx; d->from = from; d->to = to;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  d->flag_cap = std::max<int64_t>(to->n, 1);
  if (hipMalloc(&d->flag, sizeof(unsigned) * d->flag_cap) != hipSuccess) { delete d; return fail(ctx, LRR_ICP_ERR_HIP, "dir_create: hipMalloc failed"); }
  if (hipMemsetAsync(d->flag, 0, sizeof(unsigned) * d->flag_cap, ctx->stream) != hipSuccess) { lrr_icp_dir_destroy(d); return fail(ctx, LRR_ICP_ERR_HIP, "dir_create: memset failed"); }
  int rc = ensure_dir_capacity(d);
  if (rc) { lrr_icp_dir_destroy(d); return rc; }
  if (hipMemsetAsync(d->slot, 0xff, sizeof(int) * d->cap, ctx->stream) != hipSuccess ||
      hipMemsetAsync(d->m_site, 0xff, sizeof(int) * d->cap, ctx->stream) != hipSuccess ||
      hipMemsetAsync(d->m_to, 0xff, sizeof(int) * d->cap, ctx->stream) != hipSuccess) { lrr_icp_dir_destroy(d); return fail(ctx, LRR_ICP_ERR_HIP, "dir_create: memset failed"); }
  *out = d;
  return LRR_ICP_OK;
}

void lrr_icp_dir_destroy(lrr_icp_dir *d) {
  if (!d) return;
  if (d->ctx) { (void)hipSetDevice(d->ctx->device); (void)hipStreamSynchronize(d->ctx->stream); if (d->ctx->err_pending == d) d->ctx->err_pending = nullptr; }
  void *ptrs[] = {d->slot, d->m_to, d->m_site, d->w, d->wc, d->r2, d->key_g, d->key_s, d->from_loc, d->q_loc, d->flag, d->d2lb};
  for (void *p : ptrs) if (p) (void)hipFree(p);
  delete d;
}

int lrr_icp_dir_set_from(lrr_icp_dir *d, lrr_icp_cloud *from) {
  if (!d || !from) return LRR_ICP_ERR_ARG;
  lrr_icp_ctx *ctx = d->ctx;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  d->from = from;
  d->n = 0;
  d->qloc_in_step = d->qps_in_step = false;
  d->prev_n = -1;
  d->searches = 0;
  d->have_bounds = false;
  d->slot_lazy = false;
  d->from_is_copy = 0;
  int rc = ensure_dir_capacity(d);
  if (rc) return rc;
  HIPCHK(ctx, hipMemsetAsync(d->slot, 0xff, sizeof(int) * d->cap, ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(d->m_to, 0xff, sizeof(int) * d->cap, ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(d->m_site, 0xff, sizeof(int) * d->cap, ctx->stream));
  return LRR_ICP_OK;
}

int64_t lrr_icp_dir_num_matches(const lrr_icp_dir *d) { return d ? d->n : 0; }

void lrr_icp_perm_keys(uint64_t *state, uint32_t keys[4]) {
  for (int r = 0; r < 4; ++r) {
    uint64_t z = (*state += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    z = z ^ (z >> 31);
    keys[r] = (uint32_t)(z >> 32);
  }
}

int64_t lrr_icp_subset_perm(int64_t i, int64_t n, const uint32_t keys[4]) {
  int half = perm_half(n);
  uint32_t mask = (uint32_t)((1u << half) - 1u);
  uint64_t x = (uint64_t)i;
  do {
    uint32_t L = (uint32_t)(x >> half), R = (uint32_t)(x & mask);
    for (int r = 0; r < 4; ++r) {
      uint32_t F = host_fmix32(R ^ keys[r]) & mask;
      uint32_t nL = R;
      R = L ^ F;
      L = nL;
    }
    x = ((uint64_t)L << half) | R;
  } while (x >= (uint64_t)n);
  return (int64_t)x;
}

// what one K1 submission needs after its arguments have been checked
// Subsets the thread-per-SOURCE kernels serve (every source inverts the keyed permutation for itself; both directions in
// one launch): a quarter of the cloud or more, and — above what the coarse pair path takes — an eighth: a source without
// a previous match is bounded by a point of its own cell (ball_search), so a grown subset needs no search from scratch,
// and the thread-per-slot path it would take otherwise forks and joins two streams (32000 of 200k: 72 -> 45 us per direction).
constexpr int64_t kCoarseMax = 8192;  // (16384: the wave-per-query kernel took 51 us for 2 x 16000 queries, the thread-per-source search of that level 37 + 10)
inline bool dense_level(int64_t n_points, int64_t N) { return n_points * 4 >= N || (n_points > kCoarseMax && n_points * 16 >= N); }

struct MatchPrep {
  Keys kk;
  int half = 1;
  const int *d_order = nullptr;
  GridView g;
  QSrc q;
  XfD x;
  int *counts = nullptr;
  int64_t N = 0;
  bool dense = false, steady = false, use_keys = false, empty = false;
  bool voronoi = false;   // Voronoi-lookup semantics: searches the to-cloud's site cloud with voxel-rounded queries
  bool cert = false;      // finest level with second-nearest bounds from the previous search: certificate kernel
  bool short_list = false;  // ... whose list was short last time: the listed sources go straight to the wave-per-query kernel
  bool inline_list = false; // ... or very short: the certificate kernel searches them itself
  bool growth = false;      // ball search right after a dense level grew: sources without a previous match exist
  bool qloc = false;        // the direction's cached match locations were in step with m_to when this search was prepared
  bool qps = false;         // ... and the cached projectors / strengths too
  lrr_icp_xform prev_xf;  // transform of that search
  int method = LRR_ICP_NN_GRID;
};

// Is the certificate pass worth running?  It settles a source when (distance to the previous match) + u < bound, u = how far
// the new transform moves the mapped source; the bound exceeds the previous distance by the gap to the second neighbour,
// at most kBallMarginCells h (what the ball search covers beyond its best).  While a run still moves (u of that order or
// more) nearly every certificate fails and the failures are searched through a list — certificate + list + wave-per-query
// kernels then cost twice a plain ball search of everything (r03_h: 55-86 against 32 us per direction at 200k).  Decided
// from what the host knows: how many sources the last certificate pass listed, and how far the transform has moved since the
// bounds were written (the largest displacement over the corners of the from-cloud's grid).
static bool certificate_pays(const lrr_icp_dir *d, const lrr_icp_xform *xf, int64_t N, double cell, bool quantized) {
  static const bool always = std::getenv("LRR_ICP_CERT_ALWAYS") != nullptr;
  if (always) return true;
  if (d->last_list_count >= 0 && d->last_list_count * 8 > N) return false;  // an eighth of the cloud failed last time
  const lrr_icp_cloud *c = d->from;
  const lrr_icp_xform &p = d->prev_xf;
  double umax = 0.0;
  for (int k = 0; k < 8; ++k) {
    const double x[3] = {c->lo[0] + ((k & 1) ? c->dim[0] * c->h : 0.0), c->lo[1] + ((k & 2) ? c->dim[1] * c->h : 0.0), c->lo[2] + ((k & 4) ? c->dim[2] * c->h : 0.0)};
    double u2 = 0.0;
    for (int r = 0; r < 3; ++r) {
      double a = xf->t[r], b = p.t[r];
      for (int q = 0; q < 3; ++q) { a += xf->A[3 * r + q] * (x[q] - xf->c[q]); b += p.A[3 * r + q] * (x[q] - p.c[q]); }
      u2 += (a - b) * (a - b);
    }
    umax = std::max(umax, std::sqrt(u2));
  }
  // (lookup semantics: a query that stays in its voxel is settled whatever the bound; a displacement of u millimetres moves
  // about 3 u of the mapped points across a voxel face: a tenth of a voxel leaves two thirds of the sources to the certificate)
  if (quantized) return umax <= 0.1;
  return umax <= (double)kBallMarginCells * cell;
}

// argument checks, capacities, explicit order upload, counters: everything before the launches.
// Work is queued on `st`; scratch comes from lane L.
static int match_prepare(lrr_icp_dir *d, const lrr_icp_xform *xf, const uint32_t keys[4], const int64_t *order,
                         int64_t n_points, int method, K1Lane &L, hipStream_t st, MatchPrep &P) {
  if (!d || !xf || n_points < 0) return LRR_ICP_ERR_ARG;
  lrr_icp_ctx *ctx = d->ctx;
  const int64_t N = d->from->n;
  if (n_points > N) return fail(ctx, LRR_ICP_ERR_ARG, "match: n_points exceeds the from-cloud");
  if (!keys && !order && n_points > 0) return fail(ctx, LRR_ICP_ERR_ARG, "match: need keys or an explicit order");
  if (n_points > 0 && d->to->n <= 0) return fail(ctx, LRR_ICP_ERR_ARG, "match: empty to-cloud");
  for (int k = 0; k < 9; ++k) if (!std::isfinite(xf->A[k])) return fail(ctx, LRR_ICP_ERR_ARG, "match: non-finite transform");
  for (int k = 0; k < 3; ++k) if (!std::isfinite(xf->t[k]) || !std::isfinite(xf->c[k])) return fail(ctx, LRR_ICP_ERR_ARG, "match: non-finite transform");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  int rc = ensure_dir_capacity(d);
  if (rc) return rc;
  d->from_is_copy = 0;
  d->n = n_points;
  P.N = N;
  P.empty = n_points == 0;
  P.qloc = d->qloc_in_step && n_points == N;  // (lookup semantics: the kernel marks the cache but does not read it — it holds FEATURE locations)
  P.qps = P.qloc && d->qps_in_step;
  d->qloc_in_step = d->qps_in_step = false;  // whatever search follows may change matches; the inline certificate kernel's launch sites restore them
  d->last_search_cert = false;
  if (P.empty) {
    HIPCHK(ctx, hipMemsetAsync(d->slot, 0xff, sizeof(int) * d->cap, st));
    return LRR_ICP_OK;
  }
  if ((rc = ensure_lists(ctx, L, (size_t)N))) return rc;
  std::memset(&P.kk, 0, sizeof P.kk);
  if (keys) for (int r = 0; r < 4; ++r) P.kk.k[r] = keys[r];
  P.use_keys = keys != nullptr;
  P.half = perm_half(N);
  if (order) {
    if ((size_t)n_points > L.order_cap) {
      if (L.d_order) (void)hipFree(L.d_order);
      L.d_order = nullptr; L.order_cap = 0;
      HIPCHK(ctx, hipMalloc(&L.d_order, sizeof(int) * (size_t)n_points));
      L.order_cap = (size_t)n_points;
    }
    std::vector<int> o32((size_t)n_points);
    std::vector<unsigned char> seen((size_t)N, 0);
    for (int64_t i = 0; i < n_points; ++i) {
      if (order[i] < 0 || order[i] >= N) return fail(ctx, LRR_ICP_ERR_ARG, "match: order index out of range");
      if (seen[(size_t)order[i]]) return fail(ctx, LRR_ICP_ERR_ARG, "match: order holds an index twice");
      seen[(size_t)order[i]] = 1;
      o32[(size_t)i] = (int)order[i];
    }
    HIPCHK(ctx, hipMemcpyAsync(L.d_order, o32.data(), sizeof(int) * (size_t)n_points, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipStreamSynchronize(st));  // o32 is pageable and dies at scope exit
    P.d_order = L.d_order;
  }
  P.method = method == LRR_ICP_NN_AUTO ? LRR_ICP_NN_GRID : method;
  P.voronoi = P.method == LRR_ICP_NN_VORONOI;
  if (P.voronoi) {
    if ((rc = lrr_icp_cloud_build_voronoi(d->to))) return rc;  // built once per cloud
    P.method = LRR_ICP_NN_GRID;
  }
  P.g = P.voronoi ? view_of_voronoi(d->to) : view_of(d->to);
  P.q.src = d->from->src; P.q.raw = nullptr; P.q.want_orig = 0; P.q.pidx = nullptr; P.q.bloc = nullptr; P.q.loc_of = nullptr; P.q.dir = 0;
  P.x = to_dev(xf);
  P.dense = !order && dense_level(n_points, N);  // fine levels: every source feature is looked at
  if (P.voronoi != d->last_voronoi) {  // previous matches stay valid bounds (m_to: features, m_site: sites), second-nearest bounds do not
    d->have_bounds = false;
    d->prev_n = -1;
    d->last_voronoi = P.voronoi;
  }
  if ((rc = fresh_counts(ctx, L, st, &P.counts))) return rc;
  // a level's later iterations: every active source carries its previous match -> ball search
  // ... and so does the first search after a dense level has grown: the sources of the old subset carry a match (any
  // target is a valid bound for the ball search, a recent one a tight bound), the new ones go to the wave-per-query
  // and brute-force phases — cheaper than the tile kernel's search from scratch (135-160 -> 75-90 us per direction)
  static const bool tile_on_growth = std::getenv("LRR_ICP_TILE_ON_GROWTH") != nullptr;
  P.growth = !tile_on_growth && P.method == LRR_ICP_NN_GRID && P.dense && d->searches > 0 && d->prev_n != n_points;
  // (lookup semantics: the tile kernel does not round its queries — the first search of a dense level is a ball search
  // too, whose sources without a previous site are bounded by a site of their own cell or walk the rings)
  if (P.voronoi && P.dense && d->prev_n != n_points) P.growth = true;
  P.steady = P.method == LRR_ICP_NN_GRID && P.dense && (d->prev_n == n_points || P.growth);
  if (P.method == LRR_ICP_NN_GRID) { d->prev_n = P.dense ? n_points : -1; ++d->searches; }
  // bounds are per source and refer to the transform of the search that wrote them: usable only when that
  // search covered every source (n_points == N twice in a row)
  static const bool no_cert = std::getenv("LRR_ICP_NO_CERT") != nullptr;
  P.cert = P.steady && n_points == N && d->have_bounds && !no_cert;
  if (P.cert && !certificate_pays(d, xf, N, P.g.h, P.voronoi)) P.cert = false;  // (the plain ball search leaves fresh bounds: the next search may certify again)
  P.prev_xf = d->prev_xf;
  // the ball kernels (with or without certificate) leave bounds; the wave-per-query kernel leaves one for a source it
  // settles through its previous match and 0 (= search next time) for one it serves without.  Not after a growth
  // search: a third of its sources had no previous match, and a certificate pass that lists them all costs more than
  // one more plain ball search (51 against 32 us per direction)
  d->have_bounds = P.steady && n_points == N && !P.growth;
  d->prev_xf = *xf;
  d->slot_lazy = P.cert;
  if (P.cert) { for (int r = 0; r < 4; ++r) d->slot_keys[r] = P.kk.k[r]; d->slot_half = P.half; }
  P.short_list = P.cert && d->last_list_count >= 0 && d->last_list_count <= kShortList;
  P.inline_list = P.cert && d->last_list_count >= 0 && d->last_list_count <= kInlineList;
  d->cert_inline = false;
  d->cert_list_count = P.cert ? P.counts + 2 : nullptr;
  d->last_list_count = -1;
  return LRR_ICP_OK;
}

static BallArgs ball_args(lrr_icp_dir *d, const MatchPrep &P, K1Lane &L) {
  BallArgs a;
  a.g = P.g; a.src = d->from->src; a.keys = P.kk; a.half = P.half; a.nsrc = (int)P.N; a.n_points = (int)d->n;
  a.xf = P.x; a.to_pt = P.voronoi ? d->to->vor->cpt : d->to->cpt;
  a.slot = d->slot; a.m_to = P.voronoi ? d->m_site : d->m_to; a.w = d->w; a.hard = L.d_hard; a.hard_count = P.counts;
  a.m_feat = P.voronoi ? d->m_to : nullptr; a.s2f = P.voronoi ? d->to->vor_site_to_feature : nullptr;
  a.budget = kBallBudget; a.max_rad = kBallCells * P.g.h;
  a.d2lb = d->d2lb;
  a.xf_prev = to_dev(&P.prev_xf);
  a.list = L.d_active; a.list_count = P.counts + 2;
  a.q_loc = P.qloc ? d->q_loc : nullptr;
  return a;
}
// the wave-per-query arguments of one direction's search: with the lookup semantics the search runs in the to-cloud's site
// cloud around the previous SITE, and the feature behind the site found goes to m_to
static HardArgs hard_args_of(lrr_icp_dir *d, const MatchPrep &P, K1Lane &L, bool with_bounds) {
  HardArgs h = hard_args(L, P.voronoi ? d->to->vor : d->to, P.g, P.q, P.x, P.voronoi ? d->m_site : d->m_to, P.counts, with_bounds ? d->d2lb : nullptr);
  if (P.voronoi) { h.m_feat = d->m_to; h.s2f = d->to->vor_site_to_feature; }
  return h;
}
inline unsigned cert_blocks_for(int64_t n) { return (unsigned)((((n + kCertBlock - 1) / kCertBlock) + 7) / 8 * 8); }
inline unsigned cert_inline_blocks_for(int64_t n) { return (unsigned)((((n + kCertBlock * kCertPer - 1) / (kCertBlock * kCertPer)) + 7) / 8 * 8); }

// the launches of one prepared submission on stream st
static int match_launch(lrr_icp_dir *d, const MatchPrep &P, K1Lane &L, hipStream_t st) {
  lrr_icp_ctx *ctx = d->ctx;
  if (P.empty) return LRR_ICP_OK;
  const int64_t N = P.N, n_points = d->n;
  if (P.method == LRR_ICP_NN_BRUTE) {
    if (P.dense) {
      hipLaunchKernelGGL(k_slots_dense, dim3(blocks_for(N)), dim3(kBlock), 0, st, d->from->src, P.kk, P.half, (int)N, (int)n_points, d->slot, d->w, L.d_active, P.counts + 2);
    } else {
      HIPCHK(ctx, hipMemsetAsync(d->slot, 0xff, sizeof(int) * d->cap, st));
      hipLaunchKernelGGL(k_slots_sparse, dim3(blocks_for(n_points)), dim3(kBlock), 0, st, P.d_order, P.use_keys ? 1 : 0, P.kk, P.half, (int)N, d->from->canon_of, (int)n_points, d->slot, d->w, L.d_active);
    }
    HIPCHK(ctx, hipGetLastError());
    return launch_brute(ctx, L, st, d->to, P.g, P.q, L.d_active, P.x, (int)n_points, d->m_to);
  }
  static const bool no_inline = std::getenv("LRR_ICP_NO_INLINE_CERT") != nullptr;
  if (P.cert && P.inline_list && !no_inline) {  // a short list last time: the certificate kernel serves its failures itself
    ctx->k1_paths[LRR_ICP_K1_CERT_INLINE] += 1;
    hipLaunchKernelGGL(k_match_cert_inline, dim3(cert_inline_blocks_for(N)), dim3(kCertBlock), 0, st, ball_args(d, P, L));
    HIPCHK(ctx, hipGetLastError());
    d->qloc_in_step = P.qloc; d->qps_in_step = P.qps;  // (the kernel marks the cache entry of every match it changes)
    d->last_search_cert = true;
    d->cert_inline = true;
    return LRR_ICP_OK;
  }
  if (P.cert) {
    BallArgs a = ball_args(d, P, L);
    ctx->k1_paths[LRR_ICP_K1_CERT] += 1;
    d->last_search_cert = true;
    hipLaunchKernelGGL(k_match_cert, dim3(cert_blocks_for(N)), dim3(kCertBlock), 0, st, a);
    HIPCHK(ctx, hipGetLastError());
    if (!P.short_list) hipLaunchKernelGGL(k_match_list, dim3(ball_blocks_for(N)), dim3(kBallBlock), 0, st, a);
  } else if (P.steady) {
    ctx->k1_paths[P.growth ? LRR_ICP_K1_BALL_GROWTH : LRR_ICP_K1_BALL] += 1;
    hipLaunchKernelGGL(k_match_ball, dim3(ball_blocks_for(N)), dim3(kBallBlock), 0, st, ball_args(d, P, L));
  } else if (P.dense) {
    ctx->k1_paths[LRR_ICP_K1_TILE] += 1;
    hipLaunchKernelGGL(k_match_tile, dim3(d->from->nchunks), dim3(kBlock), 0, st, P.g, d->from->src, d->from->chunks, P.kk, P.half, (int)N, (int)n_points, P.x, d->to->rec, d->to->orig_of, d->slot, d->m_to, d->w, L.d_hard, P.counts, kTileBudget);
  } else {
    HIPCHK(ctx, hipMemsetAsync(d->slot, 0xff, sizeof(int) * d->cap, st));
    ctx->k1_paths[LRR_ICP_K1_DIRECT] += 1;
    hipLaunchKernelGGL(k_match_direct, dim3(blocks_for(n_points)), dim3(kBlock), 0, st, P.g, P.q, P.d_order, P.use_keys ? 1 : 0, P.kk, P.half, (int)N, d->from->canon_of, P.x, (int)n_points, d->slot,
                       P.voronoi ? d->m_site : d->m_to, d->w, L.d_hard, P.counts);
  }
  HIPCHK(ctx, hipGetLastError());
  const bool direct = !P.cert && !P.steady && !P.dense;
  {
    HardArgs h = hard_args_of(d, P, L, P.steady && n_points == N && !P.growth);
    if (P.cert && ctx->prof) { h.stats = ctx->d_cert_stats; h.seen = (unsigned long long)n_points; }
    if (P.short_list) { h.hard = L.d_active; h.hard_count = P.counts + 2; }  // the certificate kernel's list itself
    hipLaunchKernelGGL(k_match_hard, dim3(hard_blocks_for(n_points)), dim3(kBlock), 0, st, h);
    HIPCHK(ctx, hipGetLastError());
  }
  if (P.voronoi && direct) {  // the thread-per-slot kernel knows sites only: the features behind them (the other kernels write both)
    hipLaunchKernelGGL(k_vor_remap, dim3(blocks_for(N)), dim3(kBlock), 0, st, d->slot, d->m_site, d->m_to, d->to->vor_site_to_feature, (int)N);
    HIPCHK(ctx, hipGetLastError());
  }
  return LRR_ICP_OK;
}

// one K1 submission on lane L and its stream
static int match_on(lrr_icp_dir *d, const lrr_icp_xform *xf, const uint32_t keys[4], const int64_t *order,
                    int64_t n_points, int method, K1Lane &L, bool profile) {
  MatchPrep P;
  int rc = match_prepare(d, xf, keys, order, n_points, method, L, L.stream, P);
  if (rc) return rc;
  ProfScope prof(d->ctx, L.stream, profile && !P.empty ? 1 : 0);
  prof.begin();
  return match_launch(d, P, L, L.stream);
}

int lrr_icp_match(lrr_icp_dir *d, const lrr_icp_xform *xf, const uint32_t keys[4], const int64_t *order,
                  int64_t n_points, int method) {
  if (!d) return LRR_ICP_ERR_ARG;
  return match_on(d, xf, keys, order, n_points, method, d->ctx->lane[0], true);
}

// Both directions of a symmetric estimator in one submission: the forward search runs on the
// context's stream, the backward search on a second stream, forked and joined with events, so the
// short phases of one direction (far queries, overflow) overlap the other's wide kernels.
int lrr_icp_match_pair(lrr_icp_dir *f, const lrr_icp_xform *xf_f, const uint32_t keys_f[4], const int64_t *order_f,
                       int64_t n_f, lrr_icp_dir *b, const lrr_icp_xform *xf_b, const uint32_t keys_b[4],
                       const int64_t *order_b, int64_t n_b, int method) {
  if (!f || !b) return LRR_ICP_ERR_ARG;
  lrr_icp_ctx *ctx = f->ctx;
  if (b->ctx != ctx) return fail(ctx, LRR_ICP_ERR_ARG, "match_pair: directions of different contexts");
  if (f == b) return fail(ctx, LRR_ICP_ERR_ARG, "match_pair: the two directions must differ");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  int rc, rc2;
  // Later iterations of a level (both directions in the ball-search state): every phase handles both
  // directions in ONE launch on the main stream (blockIdx.y = direction) — no fork/join at all.
  const bool both_steady = method != LRR_ICP_NN_BRUTE && !order_f && !order_b && n_f > 0 && n_b > 0 &&
                           dense_level(n_f, f->from->n) && dense_level(n_b, b->from->n) && f->searches > 0 && b->searches > 0;
  if (both_steady) {
    MatchPrep Pf, Pb;
    hipStream_t st = ctx->stream;
    if ((rc = match_prepare(f, xf_f, keys_f, order_f, n_f, method, ctx->lane[0], st, Pf))) return rc;
    if ((rc = match_prepare(b, xf_b, keys_b, order_b, n_b, method, ctx->lane[1], st, Pb))) return rc;
    if (Pf.steady && Pb.steady) {
      ProfScope prof(ctx, st, 2);
      BallArgs af = ball_args(f, Pf, ctx->lane[0]), ab = ball_args(b, Pb, ctx->lane[1]);
      // lists of a few hundred sources at most (as the previous search of each direction counted them): the certificate
      // kernel settles its failures itself — K1 is this one launch
      static const bool no_inline = std::getenv("LRR_ICP_NO_INLINE_CERT") != nullptr;
      if (Pf.cert && Pb.cert && Pf.inline_list && Pb.inline_list && !no_inline) {
        hipEvent_t ev0, ev1;
        prof.attach(&ev0, &ev1);
        if (ev0) hipExtLaunchKernelGGL(k_match_cert_inline2, dim3(cert_inline_blocks_for(std::max(Pf.N, Pb.N)), 2), dim3(kCertBlock), 0, st, ev0, ev1, 0, af, ab);
        else hipLaunchKernelGGL(k_match_cert_inline2, dim3(cert_inline_blocks_for(std::max(Pf.N, Pb.N)), 2), dim3(kCertBlock), 0, st, af, ab);
        HIPCHK(ctx, hipGetLastError());
        ctx->k1_paths[LRR_ICP_K1_CERT_INLINE] += 2;
        f->cert_inline = b->cert_inline = true;  // the host adds this search to the certificate statistics when it learns the counts
        f->qloc_in_step = Pf.qloc; b->qloc_in_step = Pb.qloc;
        f->qps_in_step = Pf.qps; b->qps_in_step = Pb.qps;
        f->last_search_cert = b->last_search_cert = true;
        return LRR_ICP_OK;
      }
      prof.begin();
      if (Pf.cert && Pb.cert) {
        ctx->k1_paths[LRR_ICP_K1_CERT] += 2;
        f->last_search_cert = b->last_search_cert = true;
        hipLaunchKernelGGL(k_match_cert2, dim3(cert_blocks_for(std::max(Pf.N, Pb.N)), 2), dim3(kCertBlock), 0, st, af, ab);
        HIPCHK(ctx, hipGetLastError());
        if (!(Pf.short_list && Pb.short_list)) {
          Pf.short_list = Pb.short_list = false;
          hipLaunchKernelGGL(k_match_list2, dim3(ball_blocks_for(std::max(Pf.N, Pb.N)), 2), dim3(kBallBlock), 0, st, af, ab);
        }
      } else {
        // one of the two has no bounds yet: both take the plain ball search (which writes real slots)
        f->slot_lazy = b->slot_lazy = false;
        ctx->k1_paths[LRR_ICP_K1_BALL] += (Pf.growth ? 0 : 1) + (Pb.growth ? 0 : 1);
        ctx->k1_paths[LRR_ICP_K1_BALL_GROWTH] += (Pf.growth ? 1 : 0) + (Pb.growth ? 1 : 0);
        hipLaunchKernelGGL(k_match_ball2, dim3(ball_blocks_for(std::max(Pf.N, Pb.N)), 2), dim3(kBallBlock), 0, st, af, ab);
      }
      HIPCHK(ctx, hipGetLastError());
      HardArgs hf = hard_args_of(f, Pf, ctx->lane[0], true), hb = hard_args_of(b, Pb, ctx->lane[1], true);
      if (Pf.cert && Pb.cert && ctx->prof) {
        hf.stats = hb.stats = ctx->d_cert_stats;
        hf.seen = (unsigned long long)Pf.N; hb.seen = (unsigned long long)Pb.N;
      }
      if (Pf.cert && Pb.cert && Pf.short_list && Pb.short_list) {  // the certificate kernel's lists themselves
        hf.hard = ctx->lane[0].d_active; hf.hard_count = Pf.counts + 2;
        hb.hard = ctx->lane[1].d_active; hb.hard_count = Pb.counts + 2;
      }
      hipLaunchKernelGGL(k_match_hard2, dim3(hard_blocks_for(std::max(n_f, n_b)), 2), dim3(kBlock), 0, st, hf, hb);
      HIPCHK(ctx, hipGetLastError());
      return LRR_ICP_OK;
    }
    // (not reached: the test above mirrors match_prepare's) fall through to separate launches
    ProfScope prof(ctx, st, 2);
    prof.begin();
    rc = match_launch(f, Pf, ctx->lane[0], st);
    rc2 = match_launch(b, Pb, ctx->lane[1], st);
    return rc ? rc : rc2;
  }
  // Coarse levels (both subsets sparse and small, keyed permutations): one stream, three launches for both directions
  // — list positions + work lists, wave-per-query search (around the previous match where there is one), brute force
  // for what is left.
  if ((method == LRR_ICP_NN_GRID || method == LRR_ICP_NN_AUTO || method == LRR_ICP_NN_VORONOI) && !order_f && !order_b && keys_f && keys_b && n_f > 0 && n_b > 0 &&
      n_f * 4 < f->from->n && n_b * 4 < b->from->n && n_f <= kCoarseMax && n_b <= kCoarseMax && std::getenv("LRR_ICP_NO_COARSE_PAIR") == nullptr) {
    MatchPrep Pf, Pb;
    hipStream_t st = ctx->stream;
    if ((rc = match_prepare(f, xf_f, keys_f, order_f, n_f, method, ctx->lane[0], st, Pf))) return rc;
    if ((rc = match_prepare(b, xf_b, keys_b, order_b, n_b, method, ctx->lane[1], st, Pb))) return rc;
    ProfScope prof(ctx, st, 2);
    prof.begin();
    CoarseArgs cf, cb;
    cf.src = f->from->src; cf.keys = Pf.kk; cf.half = Pf.half; cf.nsrc = (int)Pf.N; cf.n = (int)n_f; cf.slot = f->slot; cf.w = f->w;
    cf.list = ctx->lane[0].d_hard; cf.list_count = Pf.counts;
    cb.src = b->from->src; cb.keys = Pb.kk; cb.half = Pb.half; cb.nsrc = (int)Pb.N; cb.n = (int)n_b; cb.slot = b->slot; cb.w = b->w;
    cb.list = ctx->lane[1].d_hard; cb.list_count = Pb.counts;
    ctx->k1_paths[LRR_ICP_K1_COARSE_PAIR] += 2;
    hipLaunchKernelGGL(k_slots_coarse2, dim3(blocks_for(std::max(Pf.N, Pb.N)), 2), dim3(kBlock), 0, st, cf, cb);
    HIPCHK(ctx, hipGetLastError());
    HardArgs hf = hard_args_of(f, Pf, ctx->lane[0], false), hb = hard_args_of(b, Pb, ctx->lane[1], false);
    hipLaunchKernelGGL(k_match_hard2, dim3(hard_blocks_for(std::max(n_f, n_b)), 2), dim3(kBlock), 0, st, hf, hb);
    HIPCHK(ctx, hipGetLastError());
    return LRR_ICP_OK;
  }
  {
    ProfScope prof(ctx, ctx->stream, 2);
    prof.begin();  // fork .. join on the main stream: both searches
    HIPCHK(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
    HIPCHK(ctx, hipStreamWaitEvent(ctx->lane[1].stream, ctx->ev_fork, 0));
    rc = match_on(f, xf_f, keys_f, order_f, n_f, method, ctx->lane[0], false);
    rc2 = match_on(b, xf_b, keys_b, order_b, n_b, method, ctx->lane[1], false);
    // join even after a failed submission: the main stream must not run ahead of lane 1
    hipError_t e1 = hipEventRecord(ctx->ev_join, ctx->lane[1].stream);
    hipError_t e2 = e1 == hipSuccess ? hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0) : e1;
    if (e2 != hipSuccess && !rc && !rc2) return fail(ctx, LRR_ICP_ERR_HIP, std::string("match_pair join: ") + hipGetErrorString(e2));
  }
  return rc ? rc : rc2;
}

int lrr_icp_premap(lrr_icp_dir *d, const lrr_icp_xform *G) {
  if (!d || !G) return LRR_ICP_ERR_ARG;
  lrr_icp_ctx *ctx = d->ctx;
  if (d->n == 0) { d->from_is_copy = 1; return LRR_ICP_OK; }
  HIPCHK(ctx, hipSetDevice(ctx->device));
  PairSrc ps = pairs_of(d);
  // in place is safe: each thread reads its own 3 doubles before writing them
  hipLaunchKernelGGL(k_premap, dim3(blocks_for(ps.nsrc)), dim3(kBlock), 0, ctx->stream, ps, to_dev(G), d->from_loc);
  HIPCHK(ctx, hipGetLastError());
  d->from_is_copy = 1;
  return LRR_ICP_OK;
}

int lrr_icp_residual_pass(lrr_icp_dir *d, const lrr_icp_xform *xf, const lrr_icp_norm *norm, double *S, double *W) {
  if (!d || !xf) return LRR_ICP_ERR_ARG;
  lrr_icp_ctx *ctx = d->ctx;
  if (S) *S = 0.0;
  if (W) *W = 0.0;
  if (d->n == 0) return LRR_ICP_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  PairSrc ps = pairs_of(d);
  int nb = grid_for(ps.nsrc);
  hipLaunchKernelGGL(k_residual, dim3(nb), dim3(kBlock), 0, ctx->stream, ps, to_dev(xf), to_dev(norm), d->w, d->r2, d->key_s, ctx->d_partials);
  HIPCHK(ctx, hipGetLastError());
  double out[2];
  int rc = finish_reduce(ctx, nb, 2, out);
  if (rc) return rc;
  if (S) *S = out[0];
  if (W) *W = out[1];
  return LRR_ICP_OK;
}

// MUSE order-statistic tables (restated from the published VXL rrel_muse_table; not
// reference-verified: VXL is absent)
static double norm_inv_cdf(double p) {
  static const double a[] = {-3.969683028665376e+01, 2.209460984245205e+02, -2.759285104469687e+02, 1.383577518672690e+02, -3.066479806614716e+01, 2.506628277459239e+00};
  static const double b[] = {-5.447609879822406e+01, 1.615858368580409e+02, -1.556989798598866e+02, 6.680131188771972e+01, -1.328068155288572e+01};
  static const double c[] = {-7.784894002430293e-03, -3.223964580411365e-01, -2.400758277161838e+00, -2.549732539343734e+00, 4.374664141464968e+00, 2.938163982698783e+00};
  static const double dd[] = {7.784695709041462e-03, 3.224671290700398e-01, 2.445134137142996e+00, 3.754408661907416e+00};
  double x;
  if (p < 0.02425) {
    double q = std::sqrt(-2 * std::log(p));
    x = (((((c[0] * q + c[1]) * q + c[2]) * q + c[3]) * q + c[4]) * q + c[5]) / ((((dd[0] * q + dd[1]) * q + dd[2]) * q + dd[3]) * q + 1);
  } else if (p <= 1 - 0.02425) {
    double q = p - 0.5, r = q * q;
    x = (((((a[0] * r + a[1]) * r + a[2]) * r + a[3]) * r + a[4]) * r + a[5]) * q / (((((b[0] * r + b[1]) * r + b[2]) * r + b[3]) * r + b[4]) * r + 1);
  } else {
    double q = std::sqrt(-2 * std::log(1 - p));
    x = -(((((c[0] * q + c[1]) * q + c[2]) * q + c[3]) * q + c[4]) * q + c[5]) / ((((dd[0] * q + dd[1]) * q + dd[2]) * q + dd[3]) * q + 1);
  }
  double e = 0.5 * std::erfc(-x / std::sqrt(2.0)) - p;
  double u = e * std::sqrt(2.0 * M_PI) * std::exp(x * x / 2.0);
  return x - u / (1.0 + x * u / 2.0);
}

// rrel_muset_obj(0,false): the candidate inlier fractions 0.25..0.95 step 0.05 of n residuals with their order-statistic constants
struct MuseCandidate { int64_t k; double ek, sd, div; };
static std::vector<MuseCandidate> muse_candidate_table(int64_t n) {
  std::vector<MuseCandidate> cands;
  for (double frac = 0.25; frac <= 0.95 + 0.00001; frac += 0.05) {
    int64_t k = (int64_t)std::floor(frac * (double)n + 0.5);
    if (k > n) k = n;
    if (k <= 0) k = 1;
    double pk = (double)k / (double)(n + 1), qk = 1.0 - pk;
    double Qk = norm_inv_cdf((1.0 + pk) / 2.0);
    double pQk = std::sqrt(2.0 / M_PI) * std::exp(-Qk * Qk / 2.0);
    double Qp = 1.0 / pQk, Qpp = Qk / (pQk * pQk), Qppp = (1.0 + 2.0 * Qk * Qk) / (pQk * pQk * pQk);
    double np2 = (double)n + 2.0;
    double vrk = (pk * qk / np2) * Qp * Qp + (pk * qk / (np2 * np2)) * (2.0 * (qk - pk) * Qp * Qpp + pk * qk * (Qp * Qppp + Qpp * Qpp / 2.0));
    double sd = std::sqrt(vrk);
    if (Qk / sd < 3.0) continue;  // min_exp_kth_to_stddev_ratio
    double div = (double)(n + 1) * std::sqrt(2.0 / M_PI) * (1.0 - std::exp(-Qk * Qk / 2.0));
    cands.push_back({k, Qk, sd, div});
  }
  return cands;
}
static double muse_argmin(const std::vector<MuseCandidate> &cands, const double *res) {
  bool at_start = true;
  double best_sk = 0.0, best_obj = 0.0;
  for (size_t c = 0; c < cands.size(); ++c) {
    double vk = res[4 * c], sum_less = res[4 * c + 1], cnt_less = res[4 * c + 2];
    // sum of the k smallest = sum of keys below the k-th value + copies of the k-th value
    double sum_k = sum_less + ((double)cands[c].k - cnt_less) * vk;
    double sk = sum_k / cands[c].div;
    double obj = sk * cands[c].sd / cands[c].ek;
    if (at_start || obj < best_obj) { at_start = false; best_sk = sk; best_obj = obj; }
  }
  return best_sk;
}

// MUSE of one or two LONG lists (the first estimate after set_transform at a fine level) in one go: see k_muse_sums
constexpr int64_t kMuseLong = 32768;  // shorter lists: the single-workgroup select (microseconds there)
static int muse_scale_long(lrr_icp_ctx *ctx, lrr_icp_dir *const *dirs, int nd, double *const *sigma) {
  HIPCHK(ctx, hipSetDevice(ctx->device));
  std::vector<MuseCandidate> cands[2];
  MuseList L[2];
  double *d_out = ctx->d_final + kMaxRed;
  int nj = 0, first[2] = {0, 0};
  for (int q = 0; q < nd; ++q) {
    lrr_icp_dir *d = dirs[q];
    cands[q] = muse_candidate_table(d->n);
    if ((int)cands[q].size() > kMuseMaxCand || nj + (int)cands[q].size() > kMaxSelJobs) return fail(ctx, LRR_ICP_ERR_STATE, "muse_scale: candidate table overflow");
    first[q] = nj;
    for (const MuseCandidate &c : cands[q]) {
      SelJob j = make_job(d->key_s, (int)d->from->n, nullptr, 0, c.k - 1, d->n);  // keys of inactive sources are +inf and sort last
      j.out = d_out + 4 * nj;
      ctx->h_jobs[nj++] = j;
    }
    L[q].keys = d->key_s; L[q].n = (int)d->from->n; L[q].sel = d_out + 4 * first[q]; L[q].ncand = (int)cands[q].size();
  }
  if (nd == 1) { L[1] = L[0]; L[1].ncand = 0; L[1].n = 0; }
  int rc = launch_select_kth(ctx, nj);
  if (rc) return rc;
  hipLaunchKernelGGL(k_muse_sums, dim3(kMuseBlocks, nd), dim3(kBlock), 0, ctx->stream, L[0], L[1], ctx->d_partials);
  HIPCHK(ctx, hipGetLastError());
  hipLaunchKernelGGL(k_muse_final, dim3(2 * kMuseMaxCand, nd), dim3(64), 0, ctx->stream, L[0], L[1], (const double *)ctx->d_partials, kMuseBlocks);
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipMemcpyAsync(ctx->h_pinned, d_out, sizeof(double) * 4 * nj, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  for (int q = 0; q < nd; ++q) *sigma[q] = muse_argmin(cands[q], ctx->h_pinned + 4 * first[q]);
  return LRR_ICP_OK;
}

// MUSE of one or two SHORT lists in one submission: a coarse level's few keys are gathered first (the selections then sweep n
// keys instead of the whole cloud's array, mostly +inf for the inactive sources), every candidate's order statistic and the
// sum below it come from a single-workgroup select (k_select: microseconds for a few thousand keys), one copy, one wait
static int muse_scale_short(lrr_icp_ctx *ctx, lrr_icp_dir *const *dirs, int nd, double *const *sigma) {
  HIPCHK(ctx, hipSetDevice(ctx->device));
  std::vector<MuseCandidate> cands[2];
  std::vector<SelJob> jobs;
  size_t need = 0;
  for (int q = 0; q < nd; ++q) need += (size_t)dirs[q]->n;
  if (need > ctx->muse_cap) {
    if (ctx->d_muse) (void)hipFree(ctx->d_muse);
    ctx->d_muse = nullptr; ctx->muse_cap = 0;
    HIPCHK(ctx, hipMalloc(&ctx->d_muse, sizeof(double) * need));
    ctx->muse_cap = need;
  }
  size_t off = 0;
  int first[2] = {0, 0};
  for (int q = 0; q < nd; ++q) {
    lrr_icp_dir *d = dirs[q];
    const int64_t n = d->n;
    const double *keys = d->key_s;
    int nkeys = (int)d->from->n;
    if (n * 4 < d->from->n && !d->slot_lazy) {
      hipLaunchKernelGGL(k_compact_by_slot, dim3(blocks_for(d->from->n)), dim3(kBlock), 0, ctx->stream, d->slot, d->key_s, (int)d->from->n, ctx->d_muse + off);
      HIPCHK(ctx, hipGetLastError());
      keys = ctx->d_muse + off;
      nkeys = (int)n;
      off += (size_t)n;
    }
    cands[q] = muse_candidate_table(n);
    first[q] = (int)jobs.size();
    for (const MuseCandidate &c : cands[q]) jobs.push_back(make_job(keys, nkeys, nullptr, 0, c.k - 1, n));  // keys of inactive sources are +inf and sort last
  }
  std::vector<double> res(4 * jobs.size() + 4);
  int rc = run_select(ctx, jobs, res.data(), true);
  if (rc) return rc;
  for (int q = 0; q < nd; ++q) *sigma[q] = muse_argmin(cands[q], res.data() + 4 * first[q]);
  return LRR_ICP_OK;
}

int lrr_icp_muse_scale(lrr_icp_dir *d, double *sigma) {
  if (!d || !sigma) return LRR_ICP_ERR_ARG;
  lrr_icp_ctx *ctx = d->ctx;
  *sigma = 0.0;
  if (d->n <= 0) return LRR_ICP_OK;
  lrr_icp_dir *one[1] = {d};
  double *out[1] = {sigma};
  return d->n >= kMuseLong ? muse_scale_long(ctx, one, 1, out) : muse_scale_short(ctx, one, 1, out);
}

int lrr_icp_assign_weights(lrr_icp_dir *d, double sigma) {
  if (!d) return LRR_ICP_ERR_ARG;
  lrr_icp_ctx *ctx = d->ctx;
  if (d->n == 0) return LRR_ICP_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  int nsrc = (int)d->from->n;
  hipLaunchKernelGGL(k_assign, dim3(blocks_for(nsrc)), dim3(kBlock), 0, ctx->stream, d->slot, d->r2, sigma, nsrc, d->w);
  HIPCHK(ctx, hipGetLastError());
  return LRR_ICP_OK;
}

int lrr_icp_normalize_stats(lrr_icp_dir *d, lrr_icp_norm *out) {
  if (!d || !out) return LRR_ICP_ERR_ARG;
  lrr_icp_ctx *ctx = d->ctx;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  int64_t n = d->n, nt = d->to->n;
  PairSrc ps = pairs_of(d);
  double r[4];
  int nb = grid_for(ps.nsrc);
  hipLaunchKernelGGL(k_from_sum, dim3(nb), dim3(kBlock), 0, ctx->stream, ps, ctx->d_partials);
  HIPCHK(ctx, hipGetLastError());
  int rc = finish_reduce(ctx, nb, 3, r);
  if (rc) return rc;
  for (int k = 0; k < 3; ++k) out->cm[k] = r[k] / (double)n;
  unsigned gen = 0;
  if ((rc = next_flag_gen(d, &gen))) return rc;
  hipLaunchKernelGGL(k_mark, dim3(blocks_for(ps.nsrc)), dim3(kBlock), 0, ctx->stream, d->slot, d->m_to, ps.nsrc, d->flag, gen);
  HIPCHK(ctx, hipGetLastError());
  int nbt = grid_wide(nt);
  hipLaunchKernelGGL(k_to_sum, dim3(nbt), dim3(kWide), 0, ctx->stream, d->to->cpt, d->flag, gen, (int)nt, ctx->d_partials);
  HIPCHK(ctx, hipGetLastError());
  rc = finish_reduce(ctx, nbt, 4, r);
  if (rc) return rc;
  double nd = r[3];
  for (int k = 0; k < 3; ++k) out->cf[k] = r[k] / nd;
  int nb2 = std::max(nb, nbt);
  hipLaunchKernelGGL(k_radii, dim3(nb2), dim3(kBlock), 0, ctx->stream, ps, d->to->rec, d->flag, gen, (int)nt,
                     out->cm[0], out->cm[1], out->cm[2], out->cf[0], out->cf[1], out->cf[2], ctx->d_partials);
  HIPCHK(ctx, hipGetLastError());
  rc = finish_reduce(ctx, nb2, 2, r);
  if (rc) return rc;
  out->rm = r[0] / (double)n;
  out->rf = r[1] / nd;
  return LRR_ICP_OK;
}

// cdcl_normalize_matches over a MERGED list (cdcl sym .txx:742-755, estimate_LS_backward: the constraints are added
// before the statistics are taken): `own`'s matches plus the role-reversed clones of `rev`'s matches, whose from_ is
// rev's to_ and whose to_ is rev's from_; clones are new objects, so every one of them counts among the distinct to_
int lrr_icp_normalize_stats_merged(lrr_icp_dir *own, lrr_icp_dir *rev, lrr_icp_norm *out) {
  if (!own || !rev || !out) return LRR_ICP_ERR_ARG;
  lrr_icp_ctx *ctx = own->ctx;
  if (rev->ctx != ctx) return fail(ctx, LRR_ICP_ERR_ARG, "normalize_stats_merged: directions of different contexts");
  if (own->n <= 0 || rev->n <= 0) return fail(ctx, LRR_ICP_ERR_STATE, "normalize_stats_merged: both lists need matches");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const double n_own = (double)own->n, n_rev = (double)rev->n;
  PairSrc po = pairs_of(own), pr = pairs_of(rev);
  double r[8], sum_from[3], sum_to[3];
  int rc;
  const int nbo = grid_for(po.nsrc), nbr = grid_for(pr.nsrc), nt = (int)own->to->n, nbt = grid_wide(nt);
  hipLaunchKernelGGL(k_from_sum, dim3(nbo), dim3(kBlock), 0, ctx->stream, po, ctx->d_partials);
  HIPCHK(ctx, hipGetLastError());
  if ((rc = finish_reduce(ctx, nbo, 3, r))) return rc;
  for (int k = 0; k < 3; ++k) sum_from[k] = r[k];
  unsigned gen = 0;
  if ((rc = next_flag_gen(own, &gen))) return rc;
  hipLaunchKernelGGL(k_mark, dim3(blocks_for(po.nsrc)), dim3(kBlock), 0, ctx->stream, own->slot, own->m_to, po.nsrc, own->flag, gen);
  HIPCHK(ctx, hipGetLastError());
  hipLaunchKernelGGL(k_to_sum, dim3(nbt), dim3(kWide), 0, ctx->stream, own->to->cpt, own->flag, gen, nt, ctx->d_partials);
  HIPCHK(ctx, hipGetLastError());
  if ((rc = finish_reduce(ctx, nbt, 4, r))) return rc;
  const double nd = r[3];
  for (int k = 0; k < 3; ++k) sum_to[k] = r[k];
  hipLaunchKernelGGL(k_pair_stats, dim3(nbr), dim3(kBlock), 0, ctx->stream, pr, rev->to->rec, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, ctx->d_partials);
  HIPCHK(ctx, hipGetLastError());
  if ((rc = finish_reduce(ctx, nbr, 8, r))) return rc;
  for (int k = 0; k < 3; ++k) {
    out->cm[k] = (sum_from[k] + r[3 + k]) / (n_own + n_rev);  // clones' from_ = rev's to_
    out->cf[k] = (sum_to[k] + r[k]) / (nd + n_rev);           // clones' to_ = rev's from_
  }
  const int nb2 = std::max(nbo, nbt);
  hipLaunchKernelGGL(k_radii, dim3(nb2), dim3(kBlock), 0, ctx->stream, po, own->to->rec, own->flag, gen, nt,
                     out->cm[0], out->cm[1], out->cm[2], out->cf[0], out->cf[1], out->cf[2], ctx->d_partials);
  HIPCHK(ctx, hipGetLastError());
  double ro[2];
  if ((rc = finish_reduce(ctx, nb2, 2, ro))) return rc;
  hipLaunchKernelGGL(k_pair_stats, dim3(nbr), dim3(kBlock), 0, ctx->stream, pr, rev->to->rec, out->cf[0], out->cf[1], out->cf[2],
                     out->cm[0], out->cm[1], out->cm[2], ctx->d_partials);
  HIPCHK(ctx, hipGetLastError());
  if ((rc = finish_reduce(ctx, nbr, 8, r))) return rc;
  out->rm = (ro[0] + r[7]) / (n_own + n_rev);
  out->rf = (ro[1] + r[6]) / (nd + n_rev);
  return LRR_ICP_OK;
}

int lrr_icp_clone_weights(lrr_icp_dir *d) {
  if (!d) return LRR_ICP_ERR_ARG;
  lrr_icp_ctx *ctx = d->ctx;
  if (d->n == 0) return LRR_ICP_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  int nsrc = (int)d->from->n;
  hipLaunchKernelGGL(k_copy_w, dim3(blocks_for(nsrc)), dim3(kBlock), 0, ctx->stream, d->w, d->wc, nsrc);
  HIPCHK(ctx, hipGetLastError());
  return LRR_ICP_OK;
}

int lrr_icp_weight_by_strength(lrr_icp_dir *a, lrr_icp_dir *b, double *grad_scale) {
  if (!a) return LRR_ICP_ERR_ARG;
  lrr_icp_ctx *ctx = a->ctx;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  double cnt_a = 0.0, cnt_b = 0.0;
  int rc;
  const int na = (int)a->from->n, nbsrc = b ? (int)b->from->n : 0;
  if (a->n > 0) {
    int nb = grid_for(na);
    hipLaunchKernelGGL(k_strength, dim3(nb), dim3(kBlock), 0, ctx->stream, pairs_of(a), a->from->float_strength, a->w, a->key_g, ctx->d_partials);
    HIPCHK(ctx, hipGetLastError());
    if ((rc = finish_reduce(ctx, nb, 1, &cnt_a))) return rc;
  }
  if (b && b->n > 0) {
    int nb = grid_for(nbsrc);
    hipLaunchKernelGGL(k_strength, dim3(nb), dim3(kBlock), 0, ctx->stream, pairs_of(b), b->from->float_strength, b->wc, b->key_g, ctx->d_partials);
    HIPCHK(ctx, hipGetLastError());
    if ((rc = finish_reduce(ctx, nb, 1, &cnt_b))) return rc;
  }
  long long m = (long long)(cnt_a + cnt_b);
  double gs = 0.0;
  if (m > 1) {
    // masked / inactive entries carry +inf keys and sort last, so rank k among the m finite keys is rank k overall
    std::vector<SelJob> jobs(1, make_job(a->n > 0 ? a->key_g : nullptr, a->n > 0 ? na : 0,
                                         (b && b->n > 0) ? b->key_g : nullptr, (b && b->n > 0) ? nbsrc : 0, mad_rank(m), m));
    jobs[0].state = 6;
    double res[4];
    if ((rc = run_select(ctx, jobs, res))) return rc;
    gs = mad_from_kth(m, res[0]);
  }
  double f = 1.0 / gs;
  if (a->n > 0) { hipLaunchKernelGGL(k_scale_w, dim3(blocks_for(na)), dim3(kBlock), 0, ctx->stream, a->slot, a->w, na, f); HIPCHK(ctx, hipGetLastError()); }
  if (b && b->n > 0) { hipLaunchKernelGGL(k_scale_w, dim3(blocks_for(nbsrc)), dim3(kBlock), 0, ctx->stream, b->slot, b->wc, nbsrc, f); HIPCHK(ctx, hipGetLastError()); }
  if (grad_scale) *grad_scale = gs;
  return LRR_ICP_OK;
}

int lrr_icp_weight_spatially(lrr_icp_dir *a, const lrr_icp_norm *norm_a, lrr_icp_dir *b, const lrr_icp_norm *norm_b, double *spatial_scale) {
  if (!a) return LRR_ICP_ERR_ARG;
  lrr_icp_ctx *ctx = a->ctx;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const int na = (int)a->from->n, nbsrc = b ? (int)b->from->n : 0;
  if (a->n > 0) {
    hipLaunchKernelGGL(k_spatial, dim3(blocks_for(na)), dim3(kBlock), 0, ctx->stream, pairs_of(a), to_dev(norm_a), 0, a->w, a->key_s);
    HIPCHK(ctx, hipGetLastError());
  }
  if (b && b->n > 0) {
    hipLaunchKernelGGL(k_spatial, dim3(blocks_for(nbsrc)), dim3(kBlock), 0, ctx->stream, pairs_of(b), to_dev(norm_b), 1, b->wc, b->key_s);
    HIPCHK(ctx, hipGetLastError());
  }
  long long m = (long long)a->n + (b ? (long long)b->n : 0);
  double ss = 0.0;
  int rc;
  if (m > 1) {
    std::vector<SelJob> jobs(1, make_job(a->n > 0 ? a->key_s : nullptr, a->n > 0 ? na : 0,
                                         (b && b->n > 0) ? b->key_s : nullptr, (b && b->n > 0) ? nbsrc : 0, mad_rank(m), m));
    jobs[0].state = 7;
    double res[4];
    if ((rc = run_select(ctx, jobs, res))) return rc;
    ss = mad_from_kth(m, res[0]);
  }
  double f = 1.0 / ss;
  if (a->n > 0) { hipLaunchKernelGGL(k_scale_w, dim3(blocks_for(na)), dim3(kBlock), 0, ctx->stream, a->slot, a->w, na, f); HIPCHK(ctx, hipGetLastError()); }
  if (b && b->n > 0) { hipLaunchKernelGGL(k_scale_w, dim3(blocks_for(nbsrc)), dim3(kBlock), 0, ctx->stream, b->slot, b->wc, nbsrc, f); HIPCHK(ctx, hipGetLastError()); }
  if (spatial_scale) *spatial_scale = ss;
  return LRR_ICP_OK;
}

int lrr_icp_accumulate(lrr_icp_dir *d, int dof, const double jac_centre[3], const lrr_icp_norm *norm, int flags, double *AtA, double *Atb) {
  if (!d || !AtA || !Atb || !jac_centre || (dof != 6 && dof != 12)) return LRR_ICP_ERR_ARG;
  lrr_icp_ctx *ctx = d->ctx;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  std::memset(AtA, 0, sizeof(double) * dof * dof);
  std::memset(Atb, 0, sizeof(double) * dof);
  if (d->n == 0) return LRR_ICP_OK;
  int nt = dof * (dof + 1) / 2, K = nt + dof;
  PairSrc ps = pairs_of(d);
  // 90 fp64 accumulators per thread at dof 12: fewer, longer blocks
  int nb = (int)std::min<int64_t>(std::max<int64_t>(1, ((int64_t)ps.nsrc + kBlock - 1) / kBlock), 256);
  NormD nm = to_dev(norm);
  const double *w = (flags & LRR_ICP_ACC_CLONE_W) ? d->wc : d->w;
  int rigid = (flags & LRR_ICP_ACC_RIGID_INCREMENT) ? 1 : 0, rev = (flags & LRR_ICP_ACC_REVERSED) ? 1 : 0;
  bool premul = (flags & LRR_ICP_ACC_PREMUL_SKIP) != 0;
  if (dof == 12 && premul) hipLaunchKernelGGL((k_accumulate<12, true>), dim3(nb), dim3(kBlock), 0, ctx->stream, ps, nm, w, jac_centre[0], jac_centre[1], jac_centre[2], rigid, rev, ctx->d_partials);
  else if (dof == 12) hipLaunchKernelGGL((k_accumulate<12, false>), dim3(nb), dim3(kBlock), 0, ctx->stream, ps, nm, w, jac_centre[0], jac_centre[1], jac_centre[2], rigid, rev, ctx->d_partials);
  else if (premul) hipLaunchKernelGGL((k_accumulate<6, true>), dim3(nb), dim3(kBlock), 0, ctx->stream, ps, nm, w, jac_centre[0], jac_centre[1], jac_centre[2], rigid, rev, ctx->d_partials);
  else hipLaunchKernelGGL((k_accumulate<6, false>), dim3(nb), dim3(kBlock), 0, ctx->stream, ps, nm, w, jac_centre[0], jac_centre[1], jac_centre[2], rigid, rev, ctx->d_partials);
  HIPCHK(ctx, hipGetLastError());
  double r[kMaxRed];
  int rc = finish_reduce(ctx, nb, K, r);
  if (rc) return rc;
  int t = 0;
  for (int a = 0; a < dof; ++a)
    for (int b = a; b < dof; ++b) { AtA[a * dof + b] = r[t]; AtA[b * dof + a] = r[t]; ++t; }
  for (int a = 0; a < dof; ++a) Atb[a] = r[nt + a];
  return LRR_ICP_OK;
}

// The error metrics in two halves: launch queues k_errors (its blocks store their rows in mapped host memory, tagged with
// the submission's number) and returns; collect adds the rows in row order as they arrive (after 50 ms without them:
// stream synchronisation, which also reports a failed kernel).  Between the two the caller may queue more work on the
// context — the forward classes launch RMS_before, run the whole estimate, and collect it afterwards.  One launch is
// outstanding per context: a second one first moves the first one's sums into its direction.
static int errors_drain(lrr_icp_ctx *ctx) {
  lrr_icp_dir *d = ctx->err_pending;
  if (!d) return LRR_ICP_OK;
  ctx->err_pending = nullptr;
  const volatile double *rows = ctx->h_rows;
  const double tag = ctx->err_tag;
  const auto t0 = std::chrono::steady_clock::now();
  bool synced = false;
  double r[3] = {0.0, 0.0, 0.0};
  for (int b = 0; b < ctx->err_nb; ++b) {
    for (int k = 0; k < 3; ++k) {
      unsigned spin = 0;
      double v = 0.0;
      while (!chunk_take(rows + 8 * b + 2 * k, tag, &v)) {
        if ((++spin & 1023u) != 0 || std::chrono::steady_clock::now() - t0 < std::chrono::milliseconds(50)) continue;
        if (synced) return fail(ctx, LRR_ICP_ERR_HIP, "error sums did not arrive");
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        synced = true;
      }
      r[k] += v;
    }
  }
  for (int k = 0; k < 3; ++k) d->err_stash[k] = r[k];
  d->err_stashed = true;
  return LRR_ICP_OK;
}

int lrr_icp_errors_launch(lrr_icp_dir *d, const lrr_icp_xform *xf) {
  if (!d || !xf) return LRR_ICP_ERR_ARG;
  lrr_icp_ctx *ctx = d->ctx;
  { int rc = errors_drain(ctx); if (rc) return rc; }
  d->err_stashed = false;
  if (d->n <= 0) { d->err_stash[0] = d->err_stash[1] = d->err_stash[2] = 0.0; d->err_stashed = true; return LRR_ICP_OK; }
  HIPCHK(ctx, hipSetDevice(ctx->device));
  PairSrc ps = pairs_of(d);
  const int nb = grid_wide(ps.nsrc);
  const unsigned long long seq = ++ctx->seq;
  const double tag = seq_tag(seq);
  hipLaunchKernelGGL(k_errors, dim3(nb), dim3(kWide), 0, ctx->stream, ps, to_dev(xf), to_dev((const lrr_icp_norm *)nullptr), d->w, ctx->d_partials, ctx->h_rows_dev, tag);
  HIPCHK(ctx, hipGetLastError());
  ctx->err_pending = d; ctx->err_tag = tag; ctx->err_nb = nb;
  return LRR_ICP_OK;
}

int lrr_icp_errors_collect(lrr_icp_dir *d, double *sum_r2, double *sum_wr, double *sum_w) {
  if (!d) return LRR_ICP_ERR_ARG;
  lrr_icp_ctx *ctx = d->ctx;
  if (ctx->err_pending == d) { int rc = errors_drain(ctx); if (rc) return rc; }
  if (!d->err_stashed) return fail(ctx, LRR_ICP_ERR_STATE, "errors_collect: no launch outstanding for this direction");
  d->err_stashed = false;
  if (sum_r2) *sum_r2 = d->err_stash[0];
  if (sum_wr) *sum_wr = d->err_stash[1];
  if (sum_w) *sum_w = d->err_stash[2];
  return LRR_ICP_OK;
}

int lrr_icp_errors(lrr_icp_dir *d, const lrr_icp_xform *xf, double *sum_r2, double *sum_wr, double *sum_w) {
  int rc = lrr_icp_errors_launch(d, xf);
  return rc ? rc : lrr_icp_errors_collect(d, sum_r2, sum_wr, sum_w);
}

// the 12 x 12 normal matrix and right-hand side from the kKron sums of accumulate_pair_kron
static void kron_expand(const double *S, double *AtA, double *Atb) {
  static const int sym4[4][4] = {{0, 1, 2, 6}, {1, 3, 4, 7}, {2, 4, 5, 8}, {6, 7, 8, 9}};
  static const int sym3[3][3] = {{0, 1, 2}, {1, 3, 4}, {2, 4, 5}};
  for (int a = 0; a < 12; ++a) {
    const int da = a < 9 ? a / 3 : a - 9, ka = a < 9 ? a % 3 : 3;
    for (int b = 0; b < 12; ++b) {
      const int db = b < 9 ? b / 3 : b - 9, kb = b < 9 ? b % 3 : 3;
      AtA[a * 12 + b] = S[6 * sym4[ka][kb] + sym3[da][db]];
    }
    Atb[a] = S[60 + a];
  }
}

int lrr_icp_estimate_symmetric(lrr_icp_dir *F, lrr_icp_dir *B, const lrr_icp_xform *xfF, const lrr_icp_xform *xfB,
                               const lrr_icp_sym_opts *o, lrr_icp_sym_out *out) {
  if (!F || !B || !xfF || !xfB || !o || !out || (o->dof != 6 && o->dof != 12)) return LRR_ICP_ERR_ARG;
  lrr_icp_ctx *ctx = F->ctx;
  if (B->ctx != ctx) return fail(ctx, LRR_ICP_ERR_ARG, "estimate_symmetric: directions of different contexts");
  if (F->n <= 0 || B->n <= 0) return fail(ctx, LRR_ICP_ERR_STATE, "estimate_symmetric: both directions need matches");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  double *P1 = ctx->d_fused + kFusedP1, *P2 = ctx->d_fused + kFusedP2, *P3 = ctx->d_fused + kFusedP3;
  double *P4 = ctx->d_fused + kFusedP4, *P6 = ctx->d_fused + kFusedP6, *OUT = ctx->d_fused + kFusedOut;
  double *SC = OUT, *SEL = OUT + SEL_OFF, *FIN = OUT + FIN_OFF;
  DirArgs aF = dir_args(F), aB = dir_args(B);
  const int nsF = aF.ps.nsrc, nsB = aB.ps.nsrc;
  int nbF = grid_wide_pair(nsF), nbB = grid_wide_pair(nsB), nt = (int)F->to->n, nbT = grid_wide(nt);
  int nb3 = std::max(nbF, nbT);
  NormD ident = to_dev((const lrr_icp_norm *)nullptr);
  // F1.  The matched targets' records per source: streamed from the cache while it is in step with the matches (steady state: the
  // certificate kernel keeps it so), stored in full when the last search was a certificate pass (the next estimate will stream
  // them), locations only otherwise (a run that still moves pays no extra 56 B per match)
  auto cache_bits = [](const lrr_icp_dir *d) {
    const bool full = d->n == d->from->n && !d->from_is_copy;
    return !full ? 0 : d->qps_in_step ? (int)kMatchCached : d->last_search_cert ? (int)kMatchCacheFill : 0;
  };
  const int bitsF = cache_bits(F), bitsB = cache_bits(B);
  aF.float_product |= bitsF; aB.float_product |= bitsB;
  if (o->normalize) { int rc = next_flag_gen(F, &aF.gen); if (rc) return rc; }
  hipLaunchKernelGGL(k_f_residual, dim3(nbF + nbB), dim3(kWide), 0, st, aF, aB, nbF, nbB, to_dev(xfF), to_dev(xfB), ident, o->normalize,
                     o->scale_mode_f == LRR_ICP_SCALE_MUSE ? 1 : 0, o->scale_mode_b == LRR_ICP_SCALE_MUSE ? 1 : 0, P1);
  HIPCHK(ctx, hipGetLastError());
  // (F1 stored — or found — the location of every active source's match, and with the bits above the rest of its record)
  F->qloc_in_step = F->n == F->from->n && !F->from_is_copy;
  B->qloc_in_step = B->n == B->from->n && !B->from_is_copy;
  F->qps_in_step = bitsF != 0;
  B->qps_in_step = bitsB != 0;
  // scale modes; MUSE needs the sorted residuals: its tables and argmin stay on the host (first
  // estimate after set_transform only)
  WeightOpts wo;
  wo.rrl = o->rrl; wo.normalize = o->normalize; wo.strength_f = o->strength_f; wo.strength_b = o->strength_b;
  wo.mode_f = o->scale_mode_f; wo.mode_b = o->scale_mode_b; wo.inj_f = o->injected_f; wo.inj_b = o->injected_b;
  if (wo.mode_f == LRR_ICP_SCALE_MUSE && wo.mode_b == LRR_ICP_SCALE_MUSE && (F->n >= kMuseLong) == (B->n >= kMuseLong)) {
    lrr_icp_dir *both[2] = {F, B};  // both lists: their order statistics and sums in the same launches, one wait
    double *out2[2] = {&wo.inj_f, &wo.inj_b};
    int rc = F->n >= kMuseLong ? muse_scale_long(ctx, both, 2, out2) : muse_scale_short(ctx, both, 2, out2);
    if (rc) return rc;
    wo.mode_f = wo.mode_b = LRR_ICP_SCALE_INJECTED;
  }
  if (wo.mode_f == LRR_ICP_SCALE_MUSE) { int rc = lrr_icp_muse_scale(F, &wo.inj_f); if (rc) return rc; wo.mode_f = LRR_ICP_SCALE_INJECTED; }
  if (wo.mode_b == LRR_ICP_SCALE_MUSE) { int rc = lrr_icp_muse_scale(B, &wo.inj_b); if (rc) return rc; wo.mode_b = LRR_ICP_SCALE_INJECTED; }
  if (o->normalize) {
    hipLaunchKernelGGL(k_to_sum, dim3(nbT), dim3(kWide), 0, st, F->to->cpt, F->flag, aF.gen, nt, P2);
    HIPCHK(ctx, hipGetLastError());
    hipLaunchKernelGGL(k_f_radii, dim3(nb3), dim3(kWide), 0, st, aF, nbF, P1, P2, nbT, F->to->cpt, nt, P3, SC);
    HIPCHK(ctx, hipGetLastError());
  }
  // The order statistics ride along with the weight pass (FOLD) when the scales have settled: same lists and sizes as
  // the last two estimates and all four results within 1e-3 of their predecessors.  Otherwise the three-kernel windowed
  // select (wide window: survives any jump).
  const bool same = ctx->sel_dirs[0] == (const void *)F && ctx->sel_dirs[1] == (const void *)B && ctx->sel_n[0] == F->n && ctx->sel_n[1] == B->n;
  if (!same) ctx->sel_runs = 0;
  bool fold = same && ctx->sel_runs >= 2 && o->dof == 12 && std::getenv("LRR_ICP_NO_FOLD") == nullptr;
  for (int k = 0; k < 4 && fold; ++k)
    if (!(std::fabs(ctx->sel_last[k] - ctx->sel_prev[k]) <= 1e-3 * std::fabs(ctx->sel_last[k]))) fold = false;
  // Short lists (the coarse levels of the schedule: a few thousand matches among 200k sources) ride along as well, with the window
  // wide open: the weight pass gathers EVERY finite key (a block meets a few dozen: kFoldStage holds them, kSelCap all of them) and
  // the final block ranks them — two launches instead of four, and nothing to miss.
  static const bool no_fold_all = std::getenv("LRR_ICP_NO_FOLD_ALL") != nullptr;
  const bool fold_all = !fold && o->dof == 12 && !no_fold_all && F->n <= kFoldAllMax && B->n <= kFoldAllMax;
  if (fold_all) fold = true;
  FoldArgs fa;
  fa.prev = ctx->d_selprev; fa.win = fold_all ? nullptr : ctx->d_selwin; fa.coll = ctx->d_selcoll; fa.cursor = ctx->d_selcursor; fa.P5 = ctx->d_fused + kFusedP5;
  if (fold) hipLaunchKernelGGL((k_f_weights<true>), dim3(nbF + nbB), dim3(kWide), 0, st, aF, aB, nbF, nbB, wo, P1, P3, nb3, SC, P4, fa);
  else hipLaunchKernelGGL((k_f_weights<false>), dim3(nbF + nbB), dim3(kWide), 0, st, aF, aB, nbF, nbB, wo, P1, P3, nb3, SC, P4, fa);
  HIPCHK(ctx, hipGetLastError());
  // order statistics: ranks of the strength jobs come from the device-side counts
  SelJob *hj = ctx->h_jobs;
  int nj = 0;
  auto job = [&](const double *k0, int n0, const double *k1, int n1, long long m, bool counted, int a0, int a1, int b0, int b1, int col,
                 int ra0, int ra1, int rb0, int rb1) {
    SelJob j = make_job(k0, n0, k1, n1, -1, m);
    j.out = SEL + 4 * nj;
    j.cnt = counted ? P4 : nullptr; j.cnt_a0 = a0; j.cnt_a1 = a1; j.cnt_b0 = b0; j.cnt_b1 = b1;
    j.row_a0 = ra0; j.row_a1 = ra1; j.row_b0 = rb0; j.row_b1 = rb1; j.col = col;
    j.state = (o->rrl ? 0 : 4) + nj;
    hj[nj++] = j;
  };
  if (o->rrl) {
    job(F->key_g, nsF, nullptr, 0, 0, true, 0, nbF, 0, 0, 0, 0, nbF, 0, 0);
    job(B->key_g, nsB, nullptr, 0, 0, true, nbF, nbF + nbB, 0, 0, 0, nbF, nbF + nbB, 0, 0);
    job(F->key_s, nsF, nullptr, 0, F->n, false, 0, 0, 0, 0, 1, 0, nbF, 0, 0);
    job(B->key_s, nsB, nullptr, 0, B->n, false, 0, 0, 0, 0, 1, nbF, nbF + nbB, 0, 0);
  } else {
    job(F->key_g, nsF, B->key_g, nsB, 0, true, 0, nbF, nbF, nbF + nbB, 0, 0, nbF, nbF, nbF + nbB);
    job(F->key_s, nsF, B->key_s, nsB, F->n + B->n, false, 0, 0, 0, 0, 1, 0, nbF, nbF, nbF + nbB);
  }
  if (fold) {
    SelPack pk;
    for (int j = 0; j < kSelPack; ++j) pk.j[j] = hj[std::min(j, nj - 1)];
    hipLaunchKernelGGL(k_selw_fold_final, dim3(nj), dim3(kSelThreads), 0, st, pk, ctx->d_selprev, fold_all ? (double *)nullptr : ctx->d_selwin, (const double *)(ctx->d_fused + kFusedP5),
                       (const unsigned long long *)ctx->d_selcoll, ctx->d_selcursor);
    HIPCHK(ctx, hipGetLastError());
  } else {
    int rc = launch_select_kth(ctx, nj);
    if (rc) return rc;
  }
  // F6 + final reduction
  const bool kron = o->dof == 12;
  const int acc_cap = kron ? kKronBlocks : kAccBlocks;
  int nbAF = std::min(grid_for(nsF), acc_cap), nbAB = std::min(grid_for(nsB), acc_cap);
  const double *c = o->jac_centre;
  const int nt2 = o->dof * (o->dof + 1) / 2, K = kron ? kKron : nt2 + o->dof;
  if (kron && o->rrl) hipLaunchKernelGGL((k_f_accumulate_kron<true>), dim3(nbAF + nbAB), dim3(kBlock), 0, st, aF, aB, nbAF, nbAB, wo, SC, SEL, c[0], c[1], c[2], o->rigid_increment, P6);
  else if (kron) hipLaunchKernelGGL((k_f_accumulate_kron<false>), dim3(nbAF + nbAB), dim3(kBlock), 0, st, aF, aB, nbAF, nbAB, wo, SC, SEL, c[0], c[1], c[2], o->rigid_increment, P6);
  else if (o->rrl) hipLaunchKernelGGL((k_f_accumulate<6, true>), dim3(nbAF + nbAB), dim3(kBlock), 0, st, aF, aB, nbAF, nbAB, wo, SC, SEL, c[0], c[1], c[2], o->rigid_increment, P6);
  else hipLaunchKernelGGL((k_f_accumulate<6, false>), dim3(nbAF + nbAB), dim3(kBlock), 0, st, aF, aB, nbAF, nbAB, wo, SC, SEL, c[0], c[1], c[2], o->rigid_increment, P6);
  HIPCHK(ctx, hipGetLastError());
  const unsigned long long seq = ++ctx->seq;
  hipLaunchKernelGGL(k_reduce_partials, dim3(K), dim3(64), 0, st, P6, nbAF + nbAB, K, FIN, (const double *)OUT, (int)FIN_OFF, ctx->h_fused_dev, ctx->d_done, ctx->h_seq_dev, seq,
                     F->cert_list_count, B->cert_list_count, kron ? nbAF + nbAB : 0, seq_tag(seq), (const int *)ctx->d_fault);
  HIPCHK(ctx, hipGetLastError());
  { int rc = wait_chunks(ctx, ctx->h_fused, (int)FIN_OFF + K + 3, seq_tag(seq)); if (rc) return rc; }
  const double *h = ctx->fused_vals.data();
  if (h[FIN_OFF + K + 2] != 0.0) return match_fault(ctx, "estimate_symmetric");
  if (F->cert_list_count) F->last_list_count = (int)h[FIN_OFF + K];
  if (B->cert_list_count) B->last_list_count = (int)h[FIN_OFF + K + 1];
  note_inline_cert(ctx, F);
  note_inline_cert(ctx, B);
  F->cert_list_count = B->cert_list_count = nullptr;  // the counter slots are recycled
  for (int k = 0; k < 3; ++k) { out->norm.cm[k] = h[SC_CM + k]; out->norm.cf[k] = h[SC_CF + k]; }
  out->norm.rm = h[SC_RM]; out->norm.rf = h[SC_RF];
  out->sigma_f = h[SC_SIGF]; out->sigma_b = h[SC_SIGB];
  auto mad = [&](int slot) { const double *sl = h + SEL_OFF + 4 * slot; return sl[3] > 1.0 ? 1.4826 * (1 + 5.0 / (sl[3] - 1.0)) * sl[0] : 0.0; };
  if (o->rrl) { out->grad_scale[0] = mad(0); out->grad_scale[1] = mad(1); out->spatial_scale[0] = mad(2); out->spatial_scale[1] = mad(3); }
  else { out->grad_scale[0] = out->grad_scale[1] = mad(0); out->spatial_scale[0] = out->spatial_scale[1] = mad(1); }
  for (int k = 0; k < 4; ++k) {  // the selected values themselves: what the folded selection's windows follow
    ctx->sel_prev[k] = ctx->sel_last[k];
    ctx->sel_last[k] = h[SEL_OFF + 4 * std::min(k, nj - 1)];
  }
  ctx->sel_dirs[0] = F; ctx->sel_dirs[1] = B; ctx->sel_n[0] = F->n; ctx->sel_n[1] = B->n;
  if (ctx->sel_runs < 1000) ++ctx->sel_runs;
  if (kron) { kron_expand(h + FIN_OFF, out->AtA, out->Atb); return LRR_ICP_OK; }
  int t = 0;
  for (int a = 0; a < o->dof; ++a)
    for (int b = a; b < o->dof; ++b) { out->AtA[a * o->dof + b] = h[FIN_OFF + t]; out->AtA[b * o->dof + a] = h[FIN_OFF + t]; ++t; }
  for (int a = 0; a < o->dof; ++a) out->Atb[a] = h[FIN_OFF + nt2 + a];
  return LRR_ICP_OK;
}

// test hook: the stored match of from-feature `from_idx` becomes `value` as it is (also outside the to-cloud)
int lrr_icp_debug_set_match(lrr_icp_dir *d, int64_t from_idx, int64_t value) {
  if (!d || from_idx < 0 || from_idx >= d->from->n) return LRR_ICP_ERR_ARG;
  lrr_icp_ctx *ctx = d->ctx;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  int s = 0;
  HIPCHK(ctx, hipMemcpy(&s, d->from->canon_of + from_idx, sizeof(int), hipMemcpyDeviceToHost));
  const int v = (int)value;
  HIPCHK(ctx, hipMemcpy(d->m_to + s, &v, sizeof(int), hipMemcpyHostToDevice));
  d->qloc_in_step = d->qps_in_step = false;
  return LRR_ICP_OK;
}

int lrr_icp_cloud_set_shapes(lrr_icp_cloud *c, const uint8_t *shape, const uint8_t *ndirs, const double *dirs6) {
  if (!c || (c->n > 0 && (!shape || !ndirs || !dirs6))) return LRR_ICP_ERR_ARG;
  lrr_icp_ctx *ctx = c->ctx;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const int64_t n = c->n;
  std::vector<ShapeRec> recs((size_t)std::max<int64_t>(n, 1));
  for (int64_t sidx = 0; sidx < n; ++sidx) {
    const int i = c->h_orig_of[(size_t)sidx];
    const double *a = dirs6 + 6 * (int64_t)i, *b = a + 3;
    ShapeRec r;
    r.shape = shape[i]; r.ndirs = ndirs[i];
    for (int k = 0; k < 3; ++k) r.sd[k] = r.ndirs >= 1 ? a[k] : 0.0;
    // vnl_cross_3d(directions_[0], directions_[1])
    r.td[0] = r.ndirs >= 2 ? a[1] * b[2] - a[2] * b[1] : 0.0;
    r.td[1] = r.ndirs >= 2 ? a[2] * b[0] - a[0] * b[2] : 0.0;
    r.td[2] = r.ndirs >= 2 ? a[0] * b[1] - a[1] * b[0] : 0.0;
    recs[(size_t)sidx] = r;
  }
  HIPCHK(ctx, hipMemcpy(c->shp, recs.data(), sizeof(ShapeRec) * recs.size(), hipMemcpyHostToDevice));
  c->has_shapes = true;
  return LRR_ICP_OK;
}

int lrr_icp_cloud_set_shape_vectors(lrr_icp_cloud *c, const uint8_t *shape, const double *sheet3, const double *tube3) {
  if (!c || (c->n > 0 && (!shape || !sheet3 || !tube3))) return LRR_ICP_ERR_ARG;
  lrr_icp_ctx *ctx = c->ctx;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const int64_t n = c->n;
  std::vector<ShapeRec> recs((size_t)std::max<int64_t>(n, 1));
  for (int64_t sidx = 0; sidx < n; ++sidx) {
    const int i = c->h_orig_of[(size_t)sidx];
    ShapeRec r;
    r.shape = shape[i]; r.ndirs = 2;
    for (int k = 0; k < 3; ++k) { r.sd[k] = sheet3[3 * (int64_t)i + k]; r.td[k] = tube3[3 * (int64_t)i + k]; }
    recs[(size_t)sidx] = r;
  }
  HIPCHK(ctx, hipMemcpy(c->shp, recs.data(), sizeof(ShapeRec) * recs.size(), hipMemcpyHostToDevice));
  c->has_shapes = true;
  return LRR_ICP_OK;
}

static int angles_impl(lrr_icp_dir *d, const XfD &x, int tube, int unmapped, double *sum_wa, double *sum_w);
int lrr_icp_angles_unmapped(lrr_icp_dir *d, const double R[9], int tube, double *sum_wa, double *sum_w) {
  if (!d || !R || !sum_wa || !sum_w) return LRR_ICP_ERR_ARG;
  XfD x = identity_xf();
  for (int k = 0; k < 9; ++k) x.A[k] = R[k];
  return angles_impl(d, x, tube, 1, sum_wa, sum_w);
}
int lrr_icp_angles(lrr_icp_dir *d, const lrr_icp_xform *xf, int tube, double *sum_wa, double *sum_w) {
  if (!d || !xf || !sum_wa || !sum_w) return LRR_ICP_ERR_ARG;
  return angles_impl(d, to_dev(xf), tube, 0, sum_wa, sum_w);
}
static int angles_impl(lrr_icp_dir *d, const XfD &x, int tube, int unmapped, double *sum_wa, double *sum_w) {
  lrr_icp_ctx *ctx = d->ctx;
  *sum_wa = 0.0; *sum_w = 0.0;
  if (!d->from->has_shapes || !d->to->has_shapes) return fail(ctx, LRR_ICP_ERR_STATE, "angles: shapes and directions were not uploaded (lrr_icp_cloud_set_shapes)");
  if (d->n == 0) return LRR_ICP_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  PairSrc ps = pairs_of(d);
  int nb = grid_for(ps.nsrc);
  hipLaunchKernelGGL(k_angles, dim3(nb), dim3(kBlock), 0, ctx->stream, ps, d->from->shp, d->to->shp, x, d->w, tube ? 1 : 0, ctx->d_partials, unmapped);
  HIPCHK(ctx, hipGetLastError());
  double r[2];
  int rc = finish_reduce(ctx, nb, 2, r);
  if (rc) return rc;
  *sum_wa = r[0]; *sum_w = r[1];
  return LRR_ICP_OK;
}

int lrr_icp_transfer_covariance(lrr_icp_dir *d, int dof, const double jac_centre[3], const double *covar, double out9[9]) {
  if (!d || !jac_centre || !covar || !out9 || (dof != 6 && dof != 12)) return LRR_ICP_ERR_ARG;
  lrr_icp_ctx *ctx = d->ctx;
  for (int k = 0; k < 9; ++k) out9[k] = 0.0;
  if (d->n == 0) return LRR_ICP_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  PairSrc ps = pairs_of(d);
  int nb = grid_for(ps.nsrc);
  CovarArg C;
  std::memset(&C, 0, sizeof C);
  std::memcpy(C.c, covar, sizeof(double) * (size_t)dof * dof);
  if (dof == 12) hipLaunchKernelGGL((k_transfer_cov<12>), dim3(nb), dim3(kBlock), 0, ctx->stream, ps, C, jac_centre[0], jac_centre[1], jac_centre[2], ctx->d_partials);
  else hipLaunchKernelGGL((k_transfer_cov<6>), dim3(nb), dim3(kBlock), 0, ctx->stream, ps, C, jac_centre[0], jac_centre[1], jac_centre[2], ctx->d_partials);
  HIPCHK(ctx, hipGetLastError());
  return finish_reduce(ctx, nb, 9, out9);
}

int lrr_icp_median_error(lrr_icp_dir *d, const lrr_icp_xform *xf, double *median) {
  if (!d || !xf || !median) return LRR_ICP_ERR_ARG;
  lrr_icp_ctx *ctx = d->ctx;
  *median = 0.0;
  const int64_t n = d->n;
  if (n < 2) return LRR_ICP_OK;  // the reference reads past the end of its list here
  HIPCHK(ctx, hipSetDevice(ctx->device));
  PairSrc ps = pairs_of(d);
  // a diagnostic outside the iteration: its residuals go to scratch of their own, not to the arrays
  // an estimate in progress may still need
  double *tmp = nullptr;
  HIPCHK(ctx, hipMalloc(&tmp, sizeof(double) * 2 * (size_t)ps.nsrc));
  int nb = grid_for(ps.nsrc);
  hipLaunchKernelGGL(k_residual, dim3(nb), dim3(kBlock), 0, ctx->stream, ps, to_dev(xf), to_dev((const lrr_icp_norm *)nullptr), d->w, tmp, tmp + ps.nsrc, ctx->d_partials);
  int rc = hipGetLastError() == hipSuccess ? LRR_ICP_OK : fail(ctx, LRR_ICP_ERR_HIP, "median_error: launch failed");
  double res[4] = {0, 0, 0, 0};
  if (!rc) {
    // inactive sources carry +inf and sort last: rank (n-1)/2 + 1 among the n active ones is that rank overall
    std::vector<SelJob> jobs(1, make_job(tmp + ps.nsrc, ps.nsrc, nullptr, 0, (n - 1) / 2 + 1, n));
    rc = run_select(ctx, jobs, res);
  }
  (void)hipStreamSynchronize(ctx->stream);
  (void)hipFree(tmp);
  if (!rc) *median = res[0];
  return rc;
}

int lrr_icp_estimate_forward(lrr_icp_dir *F, const lrr_icp_xform *xf, const lrr_icp_fwd_opts *o, lrr_icp_sym_out *out) {
  if (!F || !xf || !o || !out || (o->dof != 6 && o->dof != 12)) return LRR_ICP_ERR_ARG;
  lrr_icp_ctx *ctx = F->ctx;
  if (F->n <= 0) return fail(ctx, LRR_ICP_ERR_STATE, "estimate_forward: the direction needs matches");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  double *P1 = ctx->d_fused + kFusedP1, *P2 = ctx->d_fused + kFusedP2, *P3 = ctx->d_fused + kFusedP3;
  double *P4 = ctx->d_fused + kFusedP4, *P6 = ctx->d_fused + kFusedP6, *OUT = ctx->d_fused + kFusedOut;
  double *SC = OUT, *SEL = OUT + SEL_OFF, *FIN = OUT + FIN_OFF;
  DirArgs aF = dir_args(F);
  const int nsF = aF.ps.nsrc, nbF = grid_wide(nsF), nt = (int)F->to->n, nbT = grid_wide(nt), nb3 = std::max(nbF, nbT);
  if (o->normalize) {
    int rc = next_flag_gen(F, &aF.gen);
    if (rc) return rc;
    hipLaunchKernelGGL(k_fw_stats, dim3(nbF), dim3(kWide), 0, st, aF, P1);
    HIPCHK(ctx, hipGetLastError());
    hipLaunchKernelGGL(k_to_sum, dim3(nbT), dim3(kWide), 0, st, F->to->cpt, F->flag, aF.gen, nt, P2);
    HIPCHK(ctx, hipGetLastError());
    hipLaunchKernelGGL(k_f_radii, dim3(nb3), dim3(kWide), 0, st, aF, nbF, P1, P2, nbT, F->to->cpt, nt, P3, SC);
    HIPCHK(ctx, hipGetLastError());
  }
  FwdOpts fo;
  fo.normalize = o->normalize; fo.strength = o->strength; fo.rigid = o->rigid_transform;
  fo.mode = o->scale_mode; fo.injected = o->injected;
  hipLaunchKernelGGL(k_fw_residual, dim3(nbF), dim3(kWide), 0, st, aF, fo, to_dev(xf), P3, nb3, SC, P1);
  HIPCHK(ctx, hipGetLastError());
  F->qloc_in_step = F->n == F->from->n && !F->from_is_copy;
  F->qps_in_step = false;  // (this pass stores locations only, over the marks of the certificate kernel too)
  // MUSE needs the sorted residuals: its tables and argmin stay on the host (first estimate only)
  if (fo.mode == LRR_ICP_SCALE_MUSE) { int rc = lrr_icp_muse_scale(F, &fo.injected); if (rc) return rc; fo.mode = LRR_ICP_SCALE_INJECTED; }
  // the order statistic rides along with the weight pass once the scale has settled (as in lrr_icp_estimate_symmetric)
  const bool same = ctx->sel_dirs[0] == (const void *)F && ctx->sel_dirs[1] == nullptr && ctx->sel_n[0] == F->n;
  if (!same) ctx->sel_runs = 0;
  bool fold = o->strength && same && ctx->sel_runs >= 2 && std::getenv("LRR_ICP_NO_FOLD") == nullptr &&
              std::fabs(ctx->sel_last[0] - ctx->sel_prev[0]) <= 1e-3 * std::fabs(ctx->sel_last[0]);
  FoldArgs fa;
  fa.prev = ctx->d_selprev; fa.win = ctx->d_selwin; fa.coll = ctx->d_selcoll; fa.cursor = ctx->d_selcursor; fa.P5 = ctx->d_fused + kFusedP5;
  if (fold) hipLaunchKernelGGL((k_fw_weights<true>), dim3(nbF), dim3(kWide), 0, st, aF, fo, P1, nbF, SC, P4, fa);
  else hipLaunchKernelGGL((k_fw_weights<false>), dim3(nbF), dim3(kWide), 0, st, aF, fo, P1, nbF, SC, P4, fa);
  HIPCHK(ctx, hipGetLastError());
  if (o->strength) {
    SelJob j = make_job(F->key_g, nsF, nullptr, 0, -1, 0);
    j.out = SEL;
    j.cnt = P4; j.cnt_a0 = 0; j.cnt_a1 = nbF; j.cnt_b0 = 0; j.cnt_b1 = 0;
    j.row_a0 = 0; j.row_a1 = nbF; j.row_b0 = 0; j.row_b1 = 0; j.col = 0;
    j.state = 8;
    ctx->h_jobs[0] = j;
    if (fold) {
      SelPack pk;
      for (int q = 0; q < kSelPack; ++q) pk.j[q] = j;
      hipLaunchKernelGGL(k_selw_fold_final, dim3(1), dim3(kSelThreads), 0, st, pk, ctx->d_selprev, ctx->d_selwin, (const double *)(ctx->d_fused + kFusedP5),
                         (const unsigned long long *)ctx->d_selcoll, ctx->d_selcursor);
      HIPCHK(ctx, hipGetLastError());
    } else {
      int rc = launch_select_kth(ctx, 1);
      if (rc) return rc;
    }
  }
  const bool kron = o->dof == 12;
  const int nbA = std::min(grid_for(nsF), kron ? 2 * kKronBlocks : kAccBlocks);
  const double *c = o->jac_centre;
  const int nt2 = o->dof * (o->dof + 1) / 2, K = kron ? kKron : nt2 + o->dof;
  if (kron) hipLaunchKernelGGL(k_fw_accumulate_kron, dim3(nbA), dim3(kBlock), 0, st, aF, fo, SC, SEL, c[0], c[1], c[2], o->rigid_increment, P6);
  else hipLaunchKernelGGL((k_fw_accumulate<6>), dim3(nbA), dim3(kBlock), 0, st, aF, fo, SC, SEL, c[0], c[1], c[2], o->rigid_increment, P6);
  HIPCHK(ctx, hipGetLastError());
  const unsigned long long seq = ++ctx->seq;
  hipLaunchKernelGGL(k_reduce_partials, dim3(K), dim3(64), 0, st, P6, nbA, K, FIN, (const double *)OUT, (int)FIN_OFF, ctx->h_fused_dev, ctx->d_done, ctx->h_seq_dev, seq,
                     F->cert_list_count, (const int *)nullptr, kron ? nbA : 0, seq_tag(seq), (const int *)ctx->d_fault);
  HIPCHK(ctx, hipGetLastError());
  { int rc = wait_chunks(ctx, ctx->h_fused, (int)FIN_OFF + K + 3, seq_tag(seq)); if (rc) return rc; }
  if (ctx->fused_vals[(size_t)FIN_OFF + K + 2] != 0.0) return match_fault(ctx, "estimate_forward");
  const double *h = ctx->fused_vals.data();
  if (F->cert_list_count) F->last_list_count = (int)h[FIN_OFF + K];
  note_inline_cert(ctx, F);
  F->cert_list_count = nullptr;
  for (int k = 0; k < 3; ++k) { out->norm.cm[k] = h[SC_CM + k]; out->norm.cf[k] = h[SC_CF + k]; }
  out->norm.rm = h[SC_RM]; out->norm.rf = h[SC_RF];
  out->sigma_f = h[SC_SIGF]; out->sigma_b = 0.0;
  const double *sl = h + SEL_OFF;
  out->grad_scale[0] = out->grad_scale[1] = (o->strength && sl[3] > 1.0) ? 1.4826 * (1 + 5.0 / (sl[3] - 1.0)) * sl[0] : 0.0;
  out->spatial_scale[0] = out->spatial_scale[1] = 0.0;
  ctx->sel_prev[0] = ctx->sel_last[0];
  ctx->sel_last[0] = o->strength ? sl[0] : 0.0;
  ctx->sel_dirs[0] = F; ctx->sel_dirs[1] = nullptr; ctx->sel_n[0] = F->n; ctx->sel_n[1] = 0;
  if (ctx->sel_runs < 1000) ++ctx->sel_runs;
  if (kron) { kron_expand(h + FIN_OFF, out->AtA, out->Atb); return LRR_ICP_OK; }
  int t = 0;
  for (int a = 0; a < o->dof; ++a)
    for (int b = a; b < o->dof; ++b) { out->AtA[a * o->dof + b] = h[FIN_OFF + t]; out->AtA[b * o->dof + a] = h[FIN_OFF + t]; ++t; }
  for (int a = 0; a < o->dof; ++a) out->Atb[a] = h[FIN_OFF + nt2 + a];
  return LRR_ICP_OK;
}

int lrr_icp_get_matches(lrr_icp_dir *d, int64_t *from_idx, int64_t *to_idx, double *w) {
  if (!d) return LRR_ICP_ERR_ARG;
  lrr_icp_ctx *ctx = d->ctx;
  int64_t n = d->n;
  if (n == 0) return LRR_ICP_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  long long *df = nullptr, *dt = nullptr;
  double *dw = nullptr;
  int rc = LRR_ICP_OK;
  if (hipMalloc(&df, sizeof(long long) * n) != hipSuccess || hipMalloc(&dt, sizeof(long long) * n) != hipSuccess ||
      hipMalloc(&dw, sizeof(double) * n) != hipSuccess)
    rc = fail(ctx, LRR_ICP_ERR_HIP, "get_matches: hipMalloc failed");
  if (!rc) {
    int nsrc = (int)d->from->n;
    Keys kk;
    for (int r = 0; r < 4; ++r) kk.k[r] = d->slot_keys[r];
    hipLaunchKernelGGL(k_export, dim3(blocks_for(nsrc)), dim3(kBlock), 0, ctx->stream, d->slot, d->m_to, d->w, d->from->orig_of, d->to->orig_of, nsrc, df, dt, dw,
                       d->slot_lazy ? 1 : 0, kk, d->slot_half);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) rc = fail(ctx, LRR_ICP_ERR_HIP, "get_matches: export failed");
  }
  static_assert(sizeof(long long) == sizeof(int64_t), "int64 layout");
  if (!rc && from_idx && hipMemcpy(from_idx, df, sizeof(int64_t) * n, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(ctx, LRR_ICP_ERR_HIP, "get_matches: download failed");
  if (!rc && to_idx && hipMemcpy(to_idx, dt, sizeof(int64_t) * n, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(ctx, LRR_ICP_ERR_HIP, "get_matches: download failed");
  if (!rc && w && hipMemcpy(w, dw, sizeof(double) * n, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(ctx, LRR_ICP_ERR_HIP, "get_matches: download failed");
  if (df) (void)hipFree(df);
  if (dt) (void)hipFree(dt);
  if (dw) (void)hipFree(dw);
  return rc;
}

int lrr_icp_nearest(lrr_icp_cloud *cloud, const double *query, int64_t nq, int method, int64_t *idx) {
  if (!cloud || nq < 0 || nq > 0x3fffffff || (nq > 0 && (!query || !idx))) return LRR_ICP_ERR_ARG;
  lrr_icp_ctx *ctx = cloud->ctx;
  if (nq == 0) return LRR_ICP_OK;
  if (cloud->n <= 0) { for (int64_t i = 0; i < nq; ++i) idx[i] = -1; return LRR_ICP_OK; }
  HIPCHK(ctx, hipSetDevice(ctx->device));
  K1Lane &L = ctx->lane[0];
  int rc = ensure_lists(ctx, L, (size_t)nq);
  if (rc) return rc;
  double *dq = nullptr;
  int *dto = nullptr;
  HIPCHK(ctx, hipMalloc(&dq, sizeof(double) * 3 * nq));
  if (hipMalloc(&dto, sizeof(int) * nq) != hipSuccess) { (void)hipFree(dq); return fail(ctx, LRR_ICP_ERR_HIP, "nearest: hipMalloc failed"); }
  if (hipMemcpy(dq, query, sizeof(double) * 3 * nq, hipMemcpyHostToDevice) != hipSuccess) rc = fail(ctx, LRR_ICP_ERR_HIP, "nearest: upload failed");
  int *counts = nullptr;
  if (!rc) rc = fresh_counts(ctx, L, ctx->stream, &counts);
  if (!rc) {
    const bool voronoi = method == LRR_ICP_NN_VORONOI;
    if (voronoi) rc = lrr_icp_cloud_build_voronoi(cloud);
    GridView g = (voronoi && !rc) ? view_of_voronoi(cloud) : view_of(cloud);
    QSrc q;
    q.src = nullptr; q.raw = dq; q.want_orig = voronoi ? 0 : 1; q.pidx = nullptr; q.bloc = nullptr; q.loc_of = nullptr; q.dir = 0;
    Keys kk;
    std::memset(&kk, 0, sizeof kk);
    XfD x = identity_xf();
    if (method == LRR_ICP_NN_AUTO) method = LRR_ICP_NN_GRID;
    ProfScope prof(ctx, ctx->stream, 1);
    prof.begin();
    if (rc) {
    } else if (method == LRR_ICP_NN_BRUTE) rc = launch_brute(ctx, L, ctx->stream, cloud, g, q, nullptr, x, (int)nq, dto);
    else {
      hipLaunchKernelGGL(k_match_direct, dim3(blocks_for(nq)), dim3(kBlock), 0, ctx->stream, g, q, (const int *)nullptr, 0, kk, 1, (int)nq, (const int *)nullptr, x, (int)nq, (int *)nullptr, dto, (double *)nullptr, L.d_hard, counts);
      if (hipGetLastError() != hipSuccess) rc = fail(ctx, LRR_ICP_ERR_HIP, "nearest: launch failed");
      if (!rc) rc = launch_hard_phases(ctx, L, ctx->stream, voronoi ? cloud->vor : cloud, g, q, x, (int)nq, dto, counts, false);
    }
  }
  std::vector<int> tmp((size_t)nq);
  if (!rc && (hipStreamSynchronize(ctx->stream) != hipSuccess || hipMemcpy(tmp.data(), dto, sizeof(int) * nq, hipMemcpyDeviceToHost) != hipSuccess))
    rc = fail(ctx, LRR_ICP_ERR_HIP, "nearest: download failed");
  if (!rc && method == LRR_ICP_NN_VORONOI) for (int64_t i = 0; i < nq; ++i) idx[i] = cloud->h_vor_site_to_orig[(size_t)tmp[(size_t)i]];  // site -> feature
  else if (!rc) for (int64_t i = 0; i < nq; ++i) idx[i] = tmp[(size_t)i];
  (void)hipFree(dq); (void)hipFree(dto);
  return rc;
}

}  // extern "C"

#include "lrr_icp_batch.inc"
#include "lrr_icp_cdc.inc"

// lrr_icp.hip — hand-written HIP kernels (gfx950 / CDNA4) + the C-ABI of include/lrr_icp.h.
//
// The per-iteration hot path of LRR's robust ICP:
//   K1  map moving features + exact nearest fixed feature      (k_match_grid / k_match_brute)
//   K2  projected residuals, robust scale, Beaton-Tukey weights (k_residual, k_select, k_assign)
//   K5  normalisation statistics                                (k_from_sum, k_to_sum, ...)
//   K3  weighted normal equations J^T P J, J^T P b              (k_accumulate)
//   K4  error metrics                                           (k_errors)
// Everything is fp64: the reference is double throughout (cdcl/cdcl_feature.h:21) and "match
// sets exactly" is a parity requirement.  This is distance search + small reductions: no MFMA.
// The file is compiled with -ffp-contract=off so that per-pair quantities are bit-identical to
// the CPU restatement used as checker; the only fused operations are the explicit fma() calls
// of dist2().  Reductions are two-stage with a fixed block->partial->final order, so results
// are run-to-run deterministic (no floating-point atomics anywhere).
#include <hip/hip_runtime.h>

#include <algorithm>
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
constexpr int kMaxBlocks = 512;      // reduction grids: 2 blocks / CU on 256 CUs
constexpr int kMaxRed = 96;          // max doubles reduced per pass (78 + 12 for dof 12)
constexpr int kRecStride = 10;       // feature record: x y z pxx pxy pxz pyy pyz pzz strength
constexpr int kScanUnroll = 8;       // candidate loads in flight per lane in the grid search
constexpr int kEasyBudget = 48;      // candidate points a lane may scan in the thread-per-query phase
constexpr int kRingEasy = 1;         // rings walked by the thread-per-query phase of the grid search
constexpr int kRingMax = 8;          // rings walked by the wave-per-query phase before the brute-force fallback
constexpr int kSelThreads = 1024;
constexpr int kMaxSelJobs = 32;
constexpr int kProfEvents = 4096;
// fused scratch: P1 (2*512 x 5), P2 (512 x 4), P3 (512 x 2), P4 (2*512 x 1), P6 (2*512 x 96), output block
constexpr int kFusedP1 = 0, kFusedP2 = kFusedP1 + 2 * 512 * 5, kFusedP3 = kFusedP2 + 512 * 4,
              kFusedP4 = kFusedP3 + 512 * 2, kFusedP6 = kFusedP4 + 2 * 512, kFusedOut = kFusedP6 + 2 * 512 * 96,
              kFusedDoubles = kFusedOut + 256;

struct GridPoint { double x, y, z; long long idx; };  // 32 B: one candidate = two 16-B loads
struct SelState { unsigned long long prefix, mask; long long k, m; };  // radix-select progress of one job

struct SelJob {
  const double *k0; int n0;
  const double *k1; int n1;
  long long k;     // 0-based rank; < 0: MAD rank (m-1)/2+1 of the m counted keys (cnt rows below)
  double *out;     // out[0] = k-th smallest, out[1] = sum of keys < it, out[2] = their count, out[3] = m
  const double *cnt;               // per-block partial counts (1 double per row) or nullptr
  int cnt_a0, cnt_a1, cnt_b0, cnt_b1;  // two row ranges whose sum is m
  long long m_fixed;               // m when cnt == nullptr
};

}  // namespace

struct lrr_icp_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  double *d_partials = nullptr;  // kMaxBlocks * kMaxRed
  double *d_final = nullptr;     // kMaxRed + select outputs
  double *h_pinned = nullptr;    // results staging
  SelJob *d_jobs = nullptr;
  SelJob *h_jobs = nullptr;
  SelState *d_selstate = nullptr;
  int *d_selhist = nullptr;      // kMaxSelJobs x 2048
  double *d_fused = nullptr;     // P1..P6 partial tables + the fused output block
  double *h_fused = nullptr;
  // brute-force partial argmin scratch
  double *d_bd = nullptr; int *d_bi = nullptr; size_t brute_cap = 0;
  int *d_overflow = nullptr, *d_hard = nullptr; size_t overflow_cap = 0; int *d_overflow_count = nullptr;  // [0] hard, [1] overflow
  int *d_order = nullptr; size_t order_cap = 0;
  // match-kernel profile (event pairs resolved lazily)
  bool prof = false;
  std::vector<hipEvent_t> pe;
  int pe_used = 0;
  double match_ms = 0.0;
  int64_t match_launches = 0;
};

struct lrr_icp_cloud {
  lrr_icp_ctx *ctx = nullptr;
  int64_t n = 0;
  int float_strength = 0;
  double *rec = nullptr;                       // n x 10, original order (gather target)
  GridPoint *gp = nullptr;                     // points sorted by grid cell (+ original index)
  int *cell_start = nullptr;                   // ncell + 1
  double lo[3] = {0, 0, 0}, h = 1.0, inv_h = 1.0, tol = 0.0;
  int dim[3] = {1, 1, 1};
};

struct lrr_icp_dir {
  lrr_icp_ctx *ctx = nullptr;
  lrr_icp_cloud *from = nullptr, *to = nullptr;
  int64_t cap = 0, n = 0;
  int *m_from = nullptr, *m_to = nullptr;
  double *w = nullptr, *wc = nullptr, *r2 = nullptr, *key_g = nullptr, *key_s = nullptr;
  double *from_loc = nullptr;  // 3 per match: pre-mapped copies
  int from_is_copy = 0;
  unsigned *flag = nullptr; int64_t flag_cap = 0;
};

namespace {

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

struct GridView {
  const GridPoint *gp;   // points sorted by grid cell
  const int *cell_start;
  double lox, loy, loz, h, inv_h, tol;
  int dx, dy, dz, n;
};

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

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// Block-wide sum of K per-thread values; result for value k lands in partials[blockIdx.x*K + k].
// Fixed order: lanes by shuffle tree, then the 4 waves in index order.
template <int K>
__device__ __forceinline__ void block_reduce_store(double (&v)[K], double *partials) {
  __shared__ double red[kBlock / 64][K];
  int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    double s = wave_sum(v[k]);
    if (lane == 0) red[wv][k] = s;
  }
  __syncthreads();
  for (int k = threadIdx.x; k < K; k += kBlock) {
    double s = red[0][k];
#pragma unroll
    for (int w2 = 1; w2 < kBlock / 64; ++w2) s += red[w2][k];
    partials[(size_t)blockIdx.x * K + k] = s;
  }
}

// one wavefront per output value: lane l sums blocks l, l+64, ... in order, then the fixed
// shuffle tree — deterministic for a given (nblocks, K)
__global__ __launch_bounds__(64) void k_reduce_partials(const double *partials, int nblocks, int K, double *out) {
  int k = blockIdx.x;
  double s = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += 64) s += partials[(size_t)b * K + k];
  s = wave_sum(s);
  if (threadIdx.x == 0) out[k] = s;
}

// ---------------------------------------------------------------------------------------------
// K1: exact nearest neighbour
// ---------------------------------------------------------------------------------------------
struct QuerySrc {
  const double *rec;   // query records
  int stride;          // doubles per record (10 for a cloud, 3 for raw points)
  const int *order;    // explicit permutation or nullptr
  int use_keys;        // keyed permutation when order == nullptr
  int nsrc;            // size of the source set (domain of the permutation)
  int half;
  Keys keys;
};

__device__ __forceinline__ int query_index(const QuerySrc &q, int i) {
  if (q.order) return q.order[i];
  if (q.use_keys) return subset_perm(i, q.nsrc, q.half, q.keys);
  return i;
}

__device__ __forceinline__ void scan_range(const GridView &g, int s, int e, double yx, double yy,
                                           double yz, double &bd, int &bi) {
  // kScanUnroll candidates are loaded before any is compared: the loads are independent, so a lane
  // keeps that many 32-B requests in flight instead of paying one L2 round trip per point
  for (int j = s; j < e; j += kScanUnroll) {
    GridPoint pt[kScanUnroll];
#pragma unroll
    for (int u = 0; u < kScanUnroll; ++u) pt[u] = g.gp[min(j + u, e - 1)];
#pragma unroll
    for (int u = 0; u < kScanUnroll; ++u) {
      if (j + u < e) {
        double d = dist2(yx, yy, yz, pt[u].x, pt[u].y, pt[u].z);
        int oi = (int)pt[u].idx;
        if (d < bd || (d == bd && oi < bi)) { bd = d; bi = oi; }
      }
    }
  }
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

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

// distance from coordinate v to the slab [lo, hi] widened by tol (cells own their points up to
// rounding: tol covers the difference between floor((x-lo)/h) and the exact cell walls)
__device__ __forceinline__ double slab_gap(double v, double lo, double hi, double tol) {
  double g = v < lo ? lo - v : (v > hi ? v - hi : 0.0);
  g -= tol;
  return g > 0.0 ? g : 0.0;
}

// Exact nearest neighbour by ring search over the uniform grid.  Ring r = the cells at Chebyshev
// distance r from the query's cell.  Inside a ring a row / cell is skipped when its box is
// farther than the best distance so far (strictly: an equally distant point with a lower index
// must still be seen); after ring r every unvisited point lies beyond an interior face of the
// (2r+1)^3 block, so the search stops once best < (distance to the nearest such face - tol)^2.
// Returns false when the query is not settled inside rings 0..ring_max or would need more than
// `budget` candidate points: such queries go to the wave-per-query kernel, so no lane of this
// kernel ever walks a long serial chain (kernel time = slowest lane).
__device__ __forceinline__ bool grid_search(const GridView &g, double yx, double yy, double yz, int ring_max,
                                            int budget, int &best) {
  int cx = clampi((int)floor((yx - g.lox) * g.inv_h), 0, g.dx - 1);
  int cy = clampi((int)floor((yy - g.loy) * g.inv_h), 0, g.dy - 1);
  int cz = clampi((int)floor((yz - g.loz) * g.inv_h), 0, g.dz - 1);
  double bd = INFINITY;
  int bi = 0x7fffffff;
  bool done = false;
#define LRR_SCAN(S, E)                                   \
  {                                                      \
    int s_ = (S), e_ = (E);                              \
    budget -= e_ - s_;                                   \
    if (budget < 0) return false;                        \
    scan_range(g, s_, e_, yx, yy, yz, bd, bi);           \
  }
  for (int r = 0; r <= ring_max && !done; ++r) {
    int x0 = max(cx - r, 0), x1 = min(cx + r, g.dx - 1);
    int y0 = max(cy - r, 0), y1 = min(cy + r, g.dy - 1);
    int z0 = max(cz - r, 0), z1 = min(cz + r, g.dz - 1);
    for (int z = z0; z <= z1; ++z) {
      double gz_ = slab_gap(yz, g.loz + z * g.h, g.loz + (z + 1) * g.h, g.tol);
      double dz2 = gz_ * gz_;
      if (dz2 > bd) continue;
      for (int y = y0; y <= y1; ++y) {
        double gy_ = slab_gap(yy, g.loy + y * g.h, g.loy + (y + 1) * g.h, g.tol);
        double dyz2 = dz2 + gy_ * gy_;
        if (dyz2 > bd) continue;
        int row = (z * g.dy + y) * g.dx;
        bool shell = (z - cz == r) || (cz - z == r) || (y - cy == r) || (cy - y == r);
        if (shell) {
          int xa = x0, xb = x1;
          if (bd < INFINITY) {  // only the cells whose x-slab meets [yx - rad, yx + rad]
            double rad = sqrt(bd - dyz2) + 2.0 * g.tol;
            xa = max(xa, (int)floor((yx - rad - g.lox) * g.inv_h));
            xb = min(xb, (int)floor((yx + rad - g.lox) * g.inv_h));
          }
          if (xa <= xb) LRR_SCAN(g.cell_start[row + xa], g.cell_start[row + xb + 1])
        } else {
          int xc = cx - r;
          if (xc >= 0) {
            double gx_ = slab_gap(yx, g.lox + xc * g.h, g.lox + (xc + 1) * g.h, g.tol);
            if (dyz2 + gx_ * gx_ <= bd) LRR_SCAN(g.cell_start[row + xc], g.cell_start[row + xc + 1])
          }
          xc = cx + r;
          if (xc <= g.dx - 1) {
            double gx_ = slab_gap(yx, g.lox + xc * g.h, g.lox + (xc + 1) * g.h, g.tol);
            if (dyz2 + gx_ * gx_ <= bd) LRR_SCAN(g.cell_start[row + xc], g.cell_start[row + xc + 1])
          }
        }
      }
    }
    double lb = INFINITY;
    if (cx - r > 0) lb = fmin(lb, yx - (g.lox + (cx - r) * g.h));
    if (cx + r < g.dx - 1) lb = fmin(lb, (g.lox + (cx + r + 1) * g.h) - yx);
    if (cy - r > 0) lb = fmin(lb, yy - (g.loy + (cy - r) * g.h));
    if (cy + r < g.dy - 1) lb = fmin(lb, (g.loy + (cy + r + 1) * g.h) - yy);
    if (cz - r > 0) lb = fmin(lb, yz - (g.loz + (cz - r) * g.h));
    if (cz + r < g.dz - 1) lb = fmin(lb, (g.loz + (cz + r + 1) * g.h) - yz);
    if (lb == INFINITY) done = true;  // the block covers the whole grid
    else {
      lb -= g.tol;
      if (lb > 0.0 && bd < lb * lb) done = true;
    }
  }
#undef LRR_SCAN
  best = (bi == 0x7fffffff) ? 0 : bi;  // 0x7fffffff only with non-finite input
  return done;
}

// Phase-1 search specialised for rings 0..1 with ONE round trip for all cell boundaries: the 4
// cell_start entries around the query's column are fetched for each of the (up to) 9 rows of the
// 3x3x3 block before anything is scanned, so a row that survives the box pruning costs only its
// point loads.  Order: own cell, ring-0 stopping test, then the other rows / cells of ring 1.
// Returns false (-> wave-per-query kernel) when ring 1 does not settle the query or more than
// `budget` candidates would have to be scanned by this lane.
__device__ __forceinline__ bool grid_search_easy(const GridView &g, double yx, double yy, double yz, int budget,
                                                 int &best) {
  int cx = clampi((int)floor((yx - g.lox) * g.inv_h), 0, g.dx - 1);
  int cy = clampi((int)floor((yy - g.loy) * g.inv_h), 0, g.dy - 1);
  int cz = clampi((int)floor((yz - g.loz) * g.inv_h), 0, g.dz - 1);
  // st[dz+1][dy+1][k] = cell_start[row + clamp(cx - 1 + k)], k = 0..3 (clamped reads stay inside
  // the row's own [row, row + dx] entries, which always exist)
  int st[3][3][4];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      int z = cz - 1 + a, y = cy - 1 + b;
      bool ok = z >= 0 && z < g.dz && y >= 0 && y < g.dy;
      int row = ok ? (z * g.dy + y) * g.dx : 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) st[a][b][k] = ok ? g.cell_start[row + clampi(cx - 1 + k, 0, g.dx)] : 0;
    }
  double bd = INFINITY;
  int bi = 0x7fffffff;
#define LRR_SCAN(S, E)                                   \
  {                                                      \
    int s_ = (S), e_ = (E);                              \
    budget -= e_ - s_;                                   \
    if (budget < 0) return false;                        \
    scan_range(g, s_, e_, yx, yy, yz, bd, bi);           \
  }
  // ring 0: the query's own cell = entries k = 1..2 of the centre row
  LRR_SCAN(st[1][1][1], st[1][1][2])
  bool done = false;
  {
    double lb = INFINITY;
    if (cx > 0) lb = fmin(lb, yx - (g.lox + cx * g.h));
    if (cx < g.dx - 1) lb = fmin(lb, (g.lox + (cx + 1) * g.h) - yx);
    if (cy > 0) lb = fmin(lb, yy - (g.loy + cy * g.h));
    if (cy < g.dy - 1) lb = fmin(lb, (g.loy + (cy + 1) * g.h) - yy);
    if (cz > 0) lb = fmin(lb, yz - (g.loz + cz * g.h));
    if (cz < g.dz - 1) lb = fmin(lb, (g.loz + (cz + 1) * g.h) - yz);
    if (lb == INFINITY) done = true;
    else { lb -= g.tol; if (lb > 0.0 && bd < lb * lb) done = true; }
  }
  if (!done) {
    // ring 1
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      int z = cz - 1 + a;
      if (z < 0 || z >= g.dz) continue;
      double gz_ = slab_gap(yz, g.loz + z * g.h, g.loz + (z + 1) * g.h, g.tol);
      double dz2 = gz_ * gz_;
      if (dz2 > bd) continue;
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        int y = cy - 1 + b;
        if (y < 0 || y >= g.dy) continue;
        double gy_ = slab_gap(yy, g.loy + y * g.h, g.loy + (y + 1) * g.h, g.tol);
        double dyz2 = dz2 + gy_ * gy_;
        if (dyz2 > bd) continue;
        if (a != 1 || b != 1) {  // shell row: the x-run cx-1..cx+1, narrowed by the best distance
          int xa = max(cx - 1, 0), xb = min(cx + 1, g.dx - 1);
          if (bd < INFINITY) {
            double rad = sqrt(bd - dyz2) + 2.0 * g.tol;
            xa = max(xa, (int)floor((yx - rad - g.lox) * g.inv_h));
            xb = min(xb, (int)floor((yx + rad - g.lox) * g.inv_h));
          }
          if (xa <= xb) LRR_SCAN(st[a][b][xa - (cx - 1)], st[a][b][xb + 1 - (cx - 1)])
        } else {  // centre row: the two end cells
          if (cx - 1 >= 0) {
            double gx_ = slab_gap(yx, g.lox + (cx - 1) * g.h, g.lox + cx * g.h, g.tol);
            if (dyz2 + gx_ * gx_ <= bd) LRR_SCAN(st[1][1][0], st[1][1][1])
          }
          if (cx + 1 <= g.dx - 1) {
            double gx_ = slab_gap(yx, g.lox + (cx + 1) * g.h, g.lox + (cx + 2) * g.h, g.tol);
            if (dyz2 + gx_ * gx_ <= bd) LRR_SCAN(st[1][1][2], st[1][1][3])
          }
        }
      }
    }
    double lb = INFINITY;
    if (cx - 1 > 0) lb = fmin(lb, yx - (g.lox + (cx - 1) * g.h));
    if (cx + 1 < g.dx - 1) lb = fmin(lb, (g.lox + (cx + 2) * g.h) - yx);
    if (cy - 1 > 0) lb = fmin(lb, yy - (g.loy + (cy - 1) * g.h));
    if (cy + 1 < g.dy - 1) lb = fmin(lb, (g.loy + (cy + 2) * g.h) - yy);
    if (cz - 1 > 0) lb = fmin(lb, yz - (g.loz + (cz - 1) * g.h));
    if (cz + 1 < g.dz - 1) lb = fmin(lb, (g.loz + (cz + 2) * g.h) - yz);
    if (lb == INFINITY) done = true;
    else { lb -= g.tol; if (lb > 0.0 && bd < lb * lb) done = true; }
  }
#undef LRR_SCAN
  best = (bi == 0x7fffffff) ? 0 : bi;
  return done;
}

// K1, query order = match-list order (explicit orders, coarse levels, stand-alone queries)
__global__ __launch_bounds__(kBlock) void k_match_grid(GridView g, QuerySrc qs, XfD xf, int n,
                                                       int *m_from, int *m_to, double *w_out,
                                                       int *overflow, int *overflow_count) {
  int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  int p = query_index(qs, i);
  const double *pr = qs.rec + (size_t)p * qs.stride;
  double yx, yy, yz;
  map_loc(xf, pr[0], pr[1], pr[2], yx, yy, yz);
  if (m_from) m_from[i] = p;
  if (w_out) w_out[i] = 1.0;
  int bi;
  if (grid_search_easy(g, yx, yy, yz, kEasyBudget, bi)) m_to[i] = bi;
  else {
    m_to[i] = -1;
    overflow[atomicAdd(overflow_count, 1)] = i;
  }
}

// K1, spatially coherent form for the fine levels: thread s takes the s-th point of the
// from-cloud in ITS grid order (coalesced loads; neighbouring lanes query neighbouring cells of
// the target grid, so they share cache lines and ring counts) and writes the result into match
// slot j = perm^-1(point), which keeps the match list identical to the direct form.
__global__ __launch_bounds__(kBlock) void k_match_grid_coherent(GridView g, GridView gfrom, Keys keys,
                                                                int half, int n_points, XfD xf,
                                                                int *m_from, int *m_to, double *w_out,
                                                                int *overflow, int *overflow_count) {
  int s = blockIdx.x * kBlock + threadIdx.x;
  if (s >= gfrom.n) return;
  GridPoint src = gfrom.gp[s];
  int p = (int)src.idx;
  int j = subset_perm_inv(p, gfrom.n, half, keys);
  if (j >= n_points) return;
  double yx, yy, yz;
  map_loc(xf, src.x, src.y, src.z, yx, yy, yz);
  m_from[j] = p;
  w_out[j] = 1.0;  // temporarily set weight to 1.0 (cdcl .txx:645)
  int bi;
  if (grid_search_easy(g, yx, yy, yz, kEasyBudget, bi)) m_to[j] = bi;
  else {
    m_to[j] = -1;
    overflow[atomicAdd(overflow_count, 1)] = j;
  }
}

// K1 phase 2: the queries phase 1 handed over (far from any target feature, or next to cells too
// dense for one lane).  One WAVEFRONT per query.  For ring r the up to 2(2r+1)^2 cell runs (a
// whole x-run for shell rows, the two end cells for interior rows) are taken 64 at a time: each
// lane box-prunes one run and fetches its [start, end) range, a wave prefix sum turns the runs
// into one candidate list, and the lanes stride over THAT list (binary search in LDS from
// candidate number to run) — every lane busy whatever the point distribution.  Lanes are merged
// by a lexicographic (distance, index) shuffle reduction; the stopping rule is grid_search's,
// so the result is the same exact nearest neighbour.
__global__ __launch_bounds__(kBlock) void k_match_hard(GridView g, QuerySrc qs, XfD xf, const int *hard,
                                                       const int *hard_count, int *m_to, int *overflow,
                                                       int *overflow_count) {
  __shared__ int seg_s[kBlock / 64][64], seg_pre[kBlock / 64][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int wave = (blockIdx.x * kBlock + threadIdx.x) >> 6, nwaves = (gridDim.x * kBlock) >> 6;
  const int cnt = *hard_count;
  for (int o = wave; o < cnt; o += nwaves) {
    int i = hard[o];
    int p = query_index(qs, i);
    const double *pr = qs.rec + (size_t)p * qs.stride;
    double yx, yy, yz;
    map_loc(xf, pr[0], pr[1], pr[2], yx, yy, yz);
    int cx = clampi((int)floor((yx - g.lox) * g.inv_h), 0, g.dx - 1);
    int cy = clampi((int)floor((yy - g.loy) * g.inv_h), 0, g.dy - 1);
    int cz = clampi((int)floor((yz - g.loz) * g.inv_h), 0, g.dz - 1);
    double bd = INFINITY;   // wave-uniform best so far
    int bi = 0x7fffffff;
    bool done = false;
    for (int r = 0; r <= kRingMax && !done; ++r) {
      const int side = 2 * r + 1, nitems = 2 * side * side;
      for (int base = 0; base < nitems; base += 64) {
        // one cell run per lane
        int t = base + lane, s0 = 0, e0 = 0;
        if (t < nitems) {
          int rowi = t >> 1, which = t & 1;
          int z = cz - r + rowi / side, y = cy - r + rowi % side;
          if (z >= 0 && z < g.dz && y >= 0 && y < g.dy) {
            double gz_ = slab_gap(yz, g.loz + z * g.h, g.loz + (z + 1) * g.h, g.tol);
            double gy_ = slab_gap(yy, g.loy + y * g.h, g.loy + (y + 1) * g.h, g.tol);
            double dyz2 = gz_ * gz_ + gy_ * gy_;
            if (dyz2 <= bd) {
              int row = (z * g.dy + y) * g.dx;
              bool shell = (z - cz == r) || (cz - z == r) || (y - cy == r) || (cy - y == r);
              if (shell) {
                if (which == 0) {
                  int xa = max(cx - r, 0), xb = min(cx + r, g.dx - 1);
                  if (bd < INFINITY) {
                    double rad = sqrt(bd - dyz2) + 2.0 * g.tol;
                    xa = max(xa, (int)floor((yx - rad - g.lox) * g.inv_h));
                    xb = min(xb, (int)floor((yx + rad - g.lox) * g.inv_h));
                  }
                  if (xa <= xb) { s0 = g.cell_start[row + xa]; e0 = g.cell_start[row + xb + 1]; }
                }
              } else if (r > 0) {
                int xc = which == 0 ? cx - r : cx + r;
                if (xc >= 0 && xc <= g.dx - 1) {
                  double gx_ = slab_gap(yx, g.lox + xc * g.h, g.lox + (xc + 1) * g.h, g.tol);
                  if (dyz2 + gx_ * gx_ <= bd) { s0 = g.cell_start[row + xc]; e0 = g.cell_start[row + xc + 1]; }
                }
              }
            }
          }
        }
        int c0 = e0 - s0, incl = c0;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
          int v = __shfl_up(incl, off, 64);
          if (lane >= off) incl += v;
        }
        int total = __shfl(incl, 63, 64);
        if (total > 0) {
          seg_s[wv][lane] = s0;
          seg_pre[wv][lane] = incl - c0;
          __builtin_amdgcn_wave_barrier();
          double ld = INFINITY;  // this lane's best in the chunk
          int li = 0x7fffffff;
          for (int c = lane; c < total; c += 64) {
            int lo = 0, hi = 64;  // last run whose exclusive prefix is <= c
#pragma unroll
            for (int step = 0; step < 6; ++step) {
              int mid = (lo + hi) >> 1;
              if (seg_pre[wv][mid] <= c) lo = mid; else hi = mid;
            }
            GridPoint pt = g.gp[seg_s[wv][lo] + (c - seg_pre[wv][lo])];
            double d = dist2(yx, yy, yz, pt.x, pt.y, pt.z);
            int oi = (int)pt.idx;
            if (d < ld || (d == ld && oi < li)) { ld = d; li = oi; }
          }
          __builtin_amdgcn_wave_barrier();
          // merge the lanes with the running best: smallest distance, ties to the lowest index
#pragma unroll
          for (int off = 32; off > 0; off >>= 1) {
            double d2 = __shfl_xor(ld, off, 64);
            int i2 = __shfl_xor(li, off, 64);
            if (d2 < ld || (d2 == ld && i2 < li)) { ld = d2; li = i2; }
          }
          if (ld < bd || (ld == bd && li < bi)) { bd = ld; bi = li; }
        }
      }
      double lb = INFINITY;
      if (cx - r > 0) lb = fmin(lb, yx - (g.lox + (cx - r) * g.h));
      if (cx + r < g.dx - 1) lb = fmin(lb, (g.lox + (cx + r + 1) * g.h) - yx);
      if (cy - r > 0) lb = fmin(lb, yy - (g.loy + (cy - r) * g.h));
      if (cy + r < g.dy - 1) lb = fmin(lb, (g.loy + (cy + r + 1) * g.h) - yy);
      if (cz - r > 0) lb = fmin(lb, yz - (g.loz + (cz - r) * g.h));
      if (cz + r < g.dz - 1) lb = fmin(lb, (g.loz + (cz + r + 1) * g.h) - yz);
      if (lb == INFINITY) done = true;
      else {
        lb -= g.tol;
        if (lb > 0.0 && bd < lb * lb) done = true;
      }
    }
    if (lane == 0) {
      if (done) m_to[i] = (bi == 0x7fffffff) ? 0 : bi;
      else overflow[atomicAdd(overflow_count, 1)] = i;
    }
  }
}

// brute-force fallback for the (rare) queries the ring search did not settle: one block each
__global__ __launch_bounds__(kBlock) void k_match_overflow(GridView g, QuerySrc qs, XfD xf,
                                                           const int *overflow,
                                                           const int *overflow_count, int *m_to) {
  __shared__ double sd[kBlock];
  __shared__ int si[kBlock];
  int cnt = *overflow_count;
  for (int o = blockIdx.x; o < cnt; o += gridDim.x) {
    int i = overflow[o];
    int p = query_index(qs, i);
    const double *pr = qs.rec + (size_t)p * qs.stride;
    double yx, yy, yz;
    map_loc(xf, pr[0], pr[1], pr[2], yx, yy, yz);
    double bd = INFINITY;
    int bi = 0x7fffffff;
    for (int j = threadIdx.x; j < g.n; j += kBlock) {
      GridPoint pt = g.gp[j];
      double d = dist2(yx, yy, yz, pt.x, pt.y, pt.z);
      int oi = (int)pt.idx;
      if (d < bd || (d == bd && oi < bi)) { bd = d; bi = oi; }
    }
    sd[threadIdx.x] = bd; si[threadIdx.x] = bi;
    __syncthreads();
    for (int s = kBlock / 2; s > 0; s >>= 1) {
      if (threadIdx.x < s) {
        double d2 = sd[threadIdx.x + s]; int i2 = si[threadIdx.x + s];
        if (d2 < sd[threadIdx.x] || (d2 == sd[threadIdx.x] && i2 < si[threadIdx.x])) { sd[threadIdx.x] = d2; si[threadIdx.x] = i2; }
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) m_to[i] = (si[0] == 0x7fffffff) ? 0 : si[0];
    __syncthreads();
  }
}

// All-vs-all matcher: each block owns kBlock*Q queries (Q per thread, in registers) and streams
// one chunk of the target cloud through LDS in tiles; every lane reads the same LDS address
// (broadcast, conflict-free).  blockIdx.y splits the target cloud so that small query counts
// still fill the chip; partial (d2, idx) pairs are merged lexicographically by k_brute_merge.
constexpr int kBruteQ = 2;
constexpr int kBruteTile = 1024;

__global__ __launch_bounds__(kBlock) void k_match_brute(GridView g, QuerySrc qs, XfD xf, int n,
                                                        int chunk, int *m_from, double *pd,
                                                        int *pi) {
  __shared__ double sx[kBruteTile], sy[kBruteTile], sz[kBruteTile];
  __shared__ int sidx[kBruteTile];
  double yx[kBruteQ], yy[kBruteQ], yz[kBruteQ], bd[kBruteQ];
  int bi[kBruteQ], qi[kBruteQ];
#pragma unroll
  for (int q = 0; q < kBruteQ; ++q) {
    int i = (blockIdx.x * kBruteQ + q) * kBlock + threadIdx.x;
    qi[q] = i;
    bd[q] = INFINITY; bi[q] = 0x7fffffff;
    yx[q] = yy[q] = yz[q] = 0.0;
    if (i < n) {
      int p = query_index(qs, i);
      const double *pr = qs.rec + (size_t)p * qs.stride;
      map_loc(xf, pr[0], pr[1], pr[2], yx[q], yy[q], yz[q]);
      if (m_from && blockIdx.y == 0) m_from[i] = p;
    }
  }
  int begin = blockIdx.y * chunk, end = min(begin + chunk, g.n);
  for (int base = begin; base < end; base += kBruteTile) {
    int tn = min(kBruteTile, end - base);
    __syncthreads();
    for (int j = threadIdx.x; j < tn; j += kBlock) {
      GridPoint pt = g.gp[base + j];
      sx[j] = pt.x; sy[j] = pt.y; sz[j] = pt.z;
      sidx[j] = (int)pt.idx;
    }
    __syncthreads();
    for (int j = 0; j < tn; ++j) {
      double px = sx[j], py = sy[j], pz = sz[j];
      int oi = sidx[j];
#pragma unroll
      for (int q = 0; q < kBruteQ; ++q) {
        double d = dist2(yx[q], yy[q], yz[q], px, py, pz);
        if (d < bd[q] || (d == bd[q] && oi < bi[q])) { bd[q] = d; bi[q] = oi; }
      }
    }
  }
#pragma unroll
  for (int q = 0; q < kBruteQ; ++q)
    if (qi[q] < n) {
      pd[(size_t)blockIdx.y * n + qi[q]] = bd[q];
      pi[(size_t)blockIdx.y * n + qi[q]] = bi[q];
    }
}

__global__ void k_brute_merge(const double *pd, const int *pi, int n, int S, int *m_to, double *w_out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (w_out) w_out[i] = 1.0;
  double bd = pd[i];
  int bi = pi[i];
  for (int s = 1; s < S; ++s) {
    double d = pd[(size_t)s * n + i];
    int oi = pi[(size_t)s * n + i];
    if (d < bd || (d == bd && oi < bi)) { bd = d; bi = oi; }
  }
  m_to[i] = (bi == 0x7fffffff) ? 0 : bi;
}

// ---------------------------------------------------------------------------------------------
// per-match passes
// ---------------------------------------------------------------------------------------------
struct PairSrc {
  const double *from_rec, *to_rec;
  const int *m_from, *m_to;
  const double *from_loc;  // pre-mapped copies or nullptr
};

__device__ __forceinline__ void load_from(const PairSrc &ps, int i, double &p0, double &p1, double &p2) {
  if (ps.from_loc) {
    p0 = ps.from_loc[3 * (size_t)i]; p1 = ps.from_loc[3 * (size_t)i + 1]; p2 = ps.from_loc[3 * (size_t)i + 2];
  } else {
    const double *r = ps.from_rec + (size_t)ps.m_from[i] * kRecStride;
    p0 = r[0]; p1 = r[1]; p2 = r[2];
  }
}

// rigid pre-map (cdcl .txx:467-482)
__global__ void k_premap(PairSrc ps, XfD G, int n, double *out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double p0, p1, p2, y0, y1, y2;
  load_from(ps, i, p0, p1, p2);
  map_loc(G, p0, p1, p2, y0, y1, y2);
  out[3 * (size_t)i] = y0; out[3 * (size_t)i + 1] = y1; out[3 * (size_t)i + 2] = y2;
}

__device__ __forceinline__ double pair_residual2(const PairSrc &ps, int i, const XfD &xf, const NormD &nm) {
  double p0, p1, p2;
  load_from(ps, i, p0, p1, p2);
  p0 = (p0 - nm.cm[0]) / nm.rm; p1 = (p1 - nm.cm[1]) / nm.rm; p2 = (p2 - nm.cm[2]) / nm.rm;
  const double *q = ps.to_rec + (size_t)ps.m_to[i] * kRecStride;
  double q0 = (q[0] - nm.cf[0]) / nm.rf, q1 = (q[1] - nm.cf[1]) / nm.rf, q2 = (q[2] - nm.cf[2]) / nm.rf;
  double y0, y1, y2;
  map_loc(xf, p0, p1, p2, y0, y1, y2);
  return proj_residual2(y0, y1, y2, q0, q1, q2, q[3], q[4], q[5], q[6], q[7], q[8]);
}

// K2 pass 1 (cdcl .txx:677-708; rrl .txx:594-617)
__global__ __launch_bounds__(kBlock) void k_residual(PairSrc ps, XfD xf, NormD nm, int n,
                                                     const double *w_prev, double *r2_out,
                                                     double *dist_out, double *partials) {
  double acc[2] = {0.0, 0.0};
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    double r2 = pair_residual2(ps, i, xf, nm);
    double wp = w_prev[i];
    acc[0] += wp * r2;
    acc[1] += wp;
    r2_out[i] = r2;
    dist_out[i] = sqrt(r2);
  }
  block_reduce_store<2>(acc, partials);
}

// K2 pass 2 (cdcl .txx:744-765, weight_BT cdcl_utils.h:142-152)
__global__ void k_assign(const double *r2, double scale, int n, double *w) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double scale2 = scale * scale;
  double u = r2[i] / scale2;
  double bt = 0.0;
  if (u <= 16.0) { double temp = 1 - u * 0.0625; bt = temp * temp; }
  w[i] = bt / scale2;
}

// K4 (cdcl .txx:196-268)
__global__ __launch_bounds__(kBlock) void k_errors(PairSrc ps, XfD xf, NormD nm, int n,
                                                   const double *w, double *partials) {
  double acc[3] = {0.0, 0.0, 0.0};
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    double r2 = pair_residual2(ps, i, xf, nm);
    double wi = w[i];
    acc[0] += r2;
    acc[1] += sqrt(r2) * wi;
    acc[2] += wi;
  }
  block_reduce_store<3>(acc, partials);
}

// K5 statistics (cdcl .txx:1175-1213)
__global__ __launch_bounds__(kBlock) void k_from_sum(PairSrc ps, int n, double *partials) {
  double acc[3] = {0.0, 0.0, 0.0};
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    double p0, p1, p2;
    load_from(ps, i, p0, p1, p2);
    acc[0] += p0; acc[1] += p1; acc[2] += p2;
  }
  block_reduce_store<3>(acc, partials);
}

__global__ void k_mark(const int *m_to, int n, unsigned *flag) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) flag[m_to[i]] = 1u;  // benign race: every writer stores 1
}

__global__ __launch_bounds__(kBlock) void k_to_sum(const double *to_rec, const unsigned *flag, int nt,
                                                   double *partials) {
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  for (int j = blockIdx.x * kBlock + threadIdx.x; j < nt; j += gridDim.x * kBlock) {
    if (flag[j]) {
      const double *q = to_rec + (size_t)j * kRecStride;
      acc[0] += q[0]; acc[1] += q[1]; acc[2] += q[2]; acc[3] += 1.0;
    }
  }
  block_reduce_store<4>(acc, partials);
}

__global__ __launch_bounds__(kBlock) void k_radii(PairSrc ps, int n, const double *to_rec,
                                                  const unsigned *flag, int nt, double cm0, double cm1,
                                                  double cm2, double cf0, double cf1, double cf2,
                                                  double *partials) {
  double acc[2] = {0.0, 0.0};
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    double p0, p1, p2;
    load_from(ps, i, p0, p1, p2);
    p0 -= cm0; p1 -= cm1; p2 -= cm2;
    acc[0] += sqrt((p0 * p0 + p1 * p1) + p2 * p2);
  }
  for (int j = blockIdx.x * kBlock + threadIdx.x; j < nt; j += gridDim.x * kBlock) {
    if (flag[j]) {
      const double *q = to_rec + (size_t)j * kRecStride;
      double q0 = q[0] - cf0, q1 = q[1] - cf1, q2 = q[2] - cf2;
      acc[1] += sqrt((q0 * q0 + q1 * q1) + q2 * q2);
    }
  }
  block_reduce_store<2>(acc, partials);
}

// weight_by_strength, first loop (cdcl .txx:832-851; rrl .txx:728-743).  reversed: the list is a
// role-reversed clone list (the product is symmetric, only the weight buffer differs).
__global__ __launch_bounds__(kBlock) void k_strength(PairSrc ps, int n, int float_product, double *w,
                                                     double *key, double *partials) {
  double acc[1] = {0.0};
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    double wi = w[i];
    double k = INFINITY;
    if (wi != 0) {
      double sp = ps.from_rec[(size_t)ps.m_from[i] * kRecStride + 9];
      double sq = ps.to_rec[(size_t)ps.m_to[i] * kRecStride + 9];
      double g = float_product ? (double)((float)sp * (float)sq) : sp * sq;
      if (g > 90000.0) g = 90000.0;
      w[i] = wi * g;
      k = g;
      acc[0] += 1.0;
    }
    key[i] = k;
  }
  block_reduce_store<1>(acc, partials);
}

// weight_spatially, first loop (cdcl .txx:977-990; rrl .txx:760-770).  use_from: the to_ of a
// role-reversed clone is this direction's from_.
__global__ void k_spatial(PairSrc ps, NormD nm, int n, int use_from, double *w, double *key) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double q0, q1, q2;
  if (use_from) {
    load_from(ps, i, q0, q1, q2);
    q0 = (q0 - nm.cm[0]) / nm.rm; q1 = (q1 - nm.cm[1]) / nm.rm; q2 = (q2 - nm.cm[2]) / nm.rm;
  } else {
    const double *q = ps.to_rec + (size_t)ps.m_to[i] * kRecStride;
    q0 = (q[0] - nm.cf[0]) / nm.rf; q1 = (q[1] - nm.cf[1]) / nm.rf; q2 = (q[2] - nm.cf[2]) / nm.rf;
  }
  double d = sqrt((q0 * q0 + q1 * q1) + q2 * q2);
  double sw = 1.0 / (1.0 + 5.0 * d * d);
  w[i] *= sw;
  key[i] = sw;
}

__global__ void k_scale_w(double *w, int n, double f) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) w[i] *= f;
}

// K3 — normal equations.  Upper triangle + rhs per thread in registers, grid-stride over the
// matches, fixed-order block reduction.  Per-pair products follow the reference's operation
// order (generic: (J^T P) J w, cdcl .txx:1061-1062; PREMUL: (J^T (P w)) J, rrl sym .txx:468-536)
// so that only the ORDER of the sum differs from the CPU path.
template <int DOF> struct Tri { static constexpr int N = DOF * (DOF + 1) / 2; };

// per-pair update of the upper triangle (NT values) and the right-hand side (DOF values).
// (jx,jy,jz): point the Jacobian is taken at; (b0,b1,b2): right-hand-side point; P: symmetric
// projector (6 unique); (c0,c1,c2): centre of the transform.
template <int DOF, bool PREMUL>
__device__ __forceinline__ void accumulate_pair(double *acc, double wi, double jx, double jy, double jz,
                                                double b0, double b1, double b2, double *P, double c0,
                                                double c1, double c2, int rigid_increment) {
  constexpr int NT = Tri<DOF>::N;
  if (rigid_increment) { b0 = b0 - jx; b1 = b1 - jy; b2 = b2 - jz; }
  if (PREMUL) {
#pragma unroll
    for (int k = 0; k < 6; ++k) P[k] = P[k] * wi;
  }
  const double Pm[3][3] = {{P[0], P[1], P[2]}, {P[1], P[3], P[4]}, {P[2], P[4], P[5]}};
  double l0 = jx - c0, l1 = jy - c1, l2 = jz - c2;
  if (DOF == 12) {
    // affine Jacobian: J[d][3d+k] = l_k, J[d][9+d] = 1 (cdcl_trans_affine.txx:141-161);
    // one non-zero per column, so (J^T P)[a][c] = lt[k_a] * P[d_a][c] exactly
    const double lt[4] = {l0, l1, l2, 1.0};
    int t = 0;
#pragma unroll
    for (int a = 0; a < 12; ++a) {
      const int da = a < 9 ? a / 3 : a - 9, ka = a < 9 ? a % 3 : 3;
      double Ma0 = lt[ka] * Pm[da][0], Ma1 = lt[ka] * Pm[da][1], Ma2 = lt[ka] * Pm[da][2];
      const double Ma[3] = {Ma0, Ma1, Ma2};
#pragma unroll
      for (int b = a; b < 12; ++b) {
        const int db = b < 9 ? b / 3 : b - 9, kb = b < 9 ? b % 3 : 3;
        double v = Ma[db] * lt[kb];
        acc[t++] += PREMUL ? v : v * wi;
      }
      double r = (Ma0 * b0 + Ma1 * b1) + Ma2 * b2;
      acc[NT + a] += PREMUL ? r : r * wi;
    }
  } else {
    // rigid small-angle Jacobian [-[l]x , I] (cdcl_trans_rigid3d.cxx:197-225)
    const double J[3][6] = {{0.0, l2, -l1, 1.0, 0.0, 0.0}, {-l2, 0.0, l0, 0.0, 1.0, 0.0}, {l1, -l0, 0.0, 0.0, 0.0, 1.0}};
    int t = 0;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      double Ma[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) Ma[c] = (J[0][a] * Pm[0][c] + J[1][a] * Pm[1][c]) + J[2][a] * Pm[2][c];
#pragma unroll
      for (int b = a; b < 6; ++b) {
        double v = (Ma[0] * J[0][b] + Ma[1] * J[1][b]) + Ma[2] * J[2][b];
        acc[t++] += PREMUL ? v : v * wi;
      }
      double r = (Ma[0] * b0 + Ma[1] * b1) + Ma[2] * b2;
      acc[NT + a] += PREMUL ? r : r * wi;
    }
  }
}

// loads one match in normalised coordinates and orients it (reversed: role-reversed term)
__device__ __forceinline__ void load_oriented(const PairSrc &ps, int i, const NormD &nm, int reversed,
                                              double &jx, double &jy, double &jz, double &b0, double &b1,
                                              double &b2, double *P) {
  double p0, p1, p2;
  load_from(ps, i, p0, p1, p2);
  p0 = (p0 - nm.cm[0]) / nm.rm; p1 = (p1 - nm.cm[1]) / nm.rm; p2 = (p2 - nm.cm[2]) / nm.rm;
  const double *qr = ps.to_rec + (size_t)ps.m_to[i] * kRecStride;
  double q0 = (qr[0] - nm.cf[0]) / nm.rf, q1 = (qr[1] - nm.cf[1]) / nm.rf, q2 = (qr[2] - nm.cf[2]) / nm.rf;
  if (!reversed) {
#pragma unroll
    for (int k = 0; k < 6; ++k) P[k] = qr[3 + k];
    jx = p0; jy = p1; jz = p2;
    b0 = q0; b1 = q1; b2 = q2;
  } else {
    const double *pr = ps.from_rec + (size_t)ps.m_from[i] * kRecStride;
#pragma unroll
    for (int k = 0; k < 6; ++k) P[k] = pr[3 + k];
    jx = q0; jy = q1; jz = q2;
    b0 = p0; b1 = p1; b2 = p2;
  }
}

template <int DOF, bool PREMUL>
__global__ __launch_bounds__(kBlock) void k_accumulate(PairSrc ps, NormD nm, int n, const double *w,
                                                       double c0, double c1, double c2,
                                                       int rigid_increment, int reversed,
                                                       double *partials) {
  constexpr int NT = Tri<DOF>::N;
  double acc[NT + DOF];
#pragma unroll
  for (int k = 0; k < NT + DOF; ++k) acc[k] = 0.0;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    double wi = w[i];
    if (PREMUL && wi < 1e-4) continue;
    double jx, jy, jz, b0, b1, b2, P[6];
    load_oriented(ps, i, nm, reversed, jx, jy, jz, b0, b1, b2, P);
    accumulate_pair<DOF, PREMUL>(acc, wi, jx, jy, jz, b0, b1, b2, P, c0, c1, c2, rigid_increment);
  }
  block_reduce_store<NT + DOF>(acc, partials);
}

// ---------------------------------------------------------------------------------------------
// exact order statistics: one 1024-thread block per job, 6-pass MSB radix select on the
// (non-negative) double bit patterns with an LDS histogram; optional sum/count of smaller keys
// (needed by the trimmed MUSE scale).  Stands in for std::nth_element / std::sort.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long job_key(const SelJob &j, int i) {
  double v = i < j.n0 ? j.k0[i] : j.k1[i - j.n0];
  return (unsigned long long)__double_as_longlong(fabs(v));
}

__global__ __launch_bounds__(kSelThreads) void k_select(const SelJob *jobs) {
  __shared__ int hist[2048];
  __shared__ int wave_tot[kSelThreads / 64];
  __shared__ int sel_bin, sel_before;
  __shared__ double red_s[kSelThreads / 64];
  __shared__ double red_c[kSelThreads / 64];
  SelJob job = jobs[blockIdx.x];
  int n = job.n0 + job.n1;
  long long k = job.k, m = job.m_fixed;
  if (job.cnt) {  // uniform branch: the whole block sums the partial counts in a fixed order
    double c = 0.0;
    for (int b = job.cnt_a0 + (int)threadIdx.x; b < job.cnt_a1; b += kSelThreads) c += job.cnt[b];
    for (int b = job.cnt_b0 + (int)threadIdx.x; b < job.cnt_b1; b += kSelThreads) c += job.cnt[b];
    c = wave_sum(c);
    if ((threadIdx.x & 63) == 0) red_c[threadIdx.x >> 6] = c;
    __syncthreads();
    double tot = 0.0;
    for (int w2 = 0; w2 < kSelThreads / 64; ++w2) tot += red_c[w2];
    __syncthreads();
    m = (long long)tot;
  }
  if (k < 0) k = m > 1 ? (m - 1) / 2 + 1 : -1;
  if (n <= 0 || k < 0 || k >= n) {
    if (threadIdx.x == 0) { job.out[0] = 0.0; job.out[1] = 0.0; job.out[2] = 0.0; job.out[3] = (double)m; }
    return;
  }
  unsigned long long prefix = 0, mask = 0;
  const int shifts[6] = {53, 42, 31, 20, 9, 0};
  const int widths[6] = {11, 11, 11, 11, 11, 9};
  for (int pass = 0; pass < 6; ++pass) {
    int shift = shifts[pass], bins = 1 << widths[pass];
    for (int b = threadIdx.x; b < 2048; b += kSelThreads) hist[b] = 0;
    __syncthreads();
    for (int base = 0; base < n; base += kSelThreads) {
      int i = base + threadIdx.x;
      bool act = false;
      int dgt = 0;
      if (i < n) {
        unsigned long long key = job_key(job, i);
        act = (key & mask) == prefix;
        dgt = (int)((key >> shift) & (unsigned long long)(bins - 1));
      }
      // keys of similar magnitude share their leading digits: let one lane add the whole group
      // (two rounds), the stragglers use their own LDS atomic
      int lane = threadIdx.x & 63;
#pragma unroll
      for (int round = 0; round < 2; ++round) {
        unsigned long long m = __ballot(act);
        if (m == 0ull) break;
        int leader = __ffsll((long long)m) - 1;
        int d0 = __shfl(dgt, leader, 64);
        unsigned long long same = __ballot(act && dgt == d0);
        if (lane == leader) atomicAdd(&hist[d0], (int)__popcll(same));
        if (dgt == d0) act = false;
      }
      if (act) atomicAdd(&hist[dgt], 1);
    }
    __syncthreads();
    // each thread owns 2 bins; block-wide exclusive scan of the pair sums
    int h0 = hist[2 * threadIdx.x], h1 = hist[2 * threadIdx.x + 1];
    int s = h0 + h1, incl = s;
    int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      int v = __shfl_up(incl, off, 64);
      if (lane >= off) incl += v;
    }
    if (lane == 63) wave_tot[wv] = incl;
    __syncthreads();
    int before = 0;
    for (int w2 = 0; w2 < wv; ++w2) before += wave_tot[w2];
    int excl = before + incl - s;
    if (s > 0 && k >= excl && k < excl + s) {
      if (k < excl + h0) { sel_bin = 2 * threadIdx.x; sel_before = excl; }
      else { sel_bin = 2 * threadIdx.x + 1; sel_before = excl + h0; }
    }
    __syncthreads();
    prefix |= (unsigned long long)sel_bin << shift;
    mask |= (unsigned long long)(bins - 1) << shift;
    k -= sel_before;
    __syncthreads();
  }
  double value = __longlong_as_double((long long)prefix);
  double sum = 0.0, cnt = 0.0;
  for (int i = threadIdx.x; i < n; i += kSelThreads) {
    unsigned long long key = job_key(job, i);
    if (key < prefix) { sum += __longlong_as_double((long long)key); cnt += 1.0; }
  }
  sum = wave_sum(sum); cnt = wave_sum(cnt);
  if ((threadIdx.x & 63) == 0) { red_s[threadIdx.x >> 6] = sum; red_c[threadIdx.x >> 6] = cnt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double s2 = 0.0, c2 = 0.0;
    for (int w2 = 0; w2 < kSelThreads / 64; ++w2) { s2 += red_s[w2]; c2 += red_c[w2]; }
    job.out[0] = value; job.out[1] = s2; job.out[2] = c2; job.out[3] = (double)m;
  }
}

// ---------------------------------------------------------------------------------------------
// fused symmetric estimate: six kernels, no host round trip in between.  Scalars (scales,
// normalisation statistics, MAD factors) are re-derived in each consumer kernel's prologue from
// the producer's per-block partials, summed by every block in the same fixed order.
// ---------------------------------------------------------------------------------------------
struct DirArgs {
  PairSrc ps;
  int n;
  double *w, *wc, *r2, *key_s, *key_g;
  unsigned *flag;
  int float_product;
};

// scalar block layout (doubles) inside the fused output buffer
enum { SC_CM = 0, SC_RM = 3, SC_CF = 4, SC_RF = 7, SC_ND = 8, SC_SIGF = 9, SC_SIGB = 10, SC_SIZE = 16,
       SEL_OFF = 16, SEL_SIZE = 16, FIN_OFF = 32, FUSED_OUT = 32 + kMaxRed };

// sum of column k over rows [r0, r1) of a row-major partials table; same value in every thread
__device__ __forceinline__ double block_total(const double *P, int r0, int r1, int K, int k) {
  __shared__ double sh[kBlock / 64];
  double s = 0.0;
  for (int b = r0 + (int)threadIdx.x; b < r1; b += kBlock) s += P[(size_t)b * K + k];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  double tot = sh[0];
#pragma unroll
  for (int w2 = 1; w2 < kBlock / 64; ++w2) tot += sh[w2];
  __syncthreads();
  return tot;
}

// F1: residuals of both lists (un-normalised coordinates: rrl sym .txx:701-705), sums for the
// RMS scale, centroid sums and distinct-target marks of the forward list
__global__ __launch_bounds__(kBlock) void k_f_residual(DirArgs F, DirArgs B, int nbF, int nbB, XfD xfF,
                                                       XfD xfB, NormD ident, int want_stats, double *P1) {
  const bool isF = (int)blockIdx.x < nbF;
  const DirArgs &D = isF ? F : B;
  const XfD &xf = isF ? xfF : xfB;
  const int lb = isF ? blockIdx.x : blockIdx.x - nbF, nb = isF ? nbF : nbB;
  double acc[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
  for (int i = lb * kBlock + threadIdx.x; i < D.n; i += nb * kBlock) {
    double r2 = pair_residual2(D.ps, i, xf, ident);
    double wp = D.w[i];
    acc[0] += wp * r2;
    acc[1] += wp;
    D.r2[i] = r2;
    D.key_s[i] = sqrt(r2);
    if (isF && want_stats) {
      double p0, p1, p2;
      load_from(D.ps, i, p0, p1, p2);
      acc[2] += p0; acc[3] += p1; acc[4] += p2;
      D.flag[D.ps.m_to[i]] = 1u;
    }
  }
  block_reduce_store<5>(acc, P1);
}

// F3: mean radii about the centroids (cdcl .txx:1195-1213)
__global__ __launch_bounds__(kBlock) void k_f_radii(DirArgs F, int nbF, const double *P1, const double *P2,
                                                    int nbT, const double *to_rec, int nt, double *P3,
                                                    double *SC) {
  double cm[3], cf[3];
  for (int k = 0; k < 3; ++k) cm[k] = block_total(P1, 0, nbF, 5, 2 + k) / (double)F.n;
  double nd = block_total(P2, 0, nbT, 4, 3);
  for (int k = 0; k < 3; ++k) cf[k] = block_total(P2, 0, nbT, 4, k) / nd;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    for (int k = 0; k < 3; ++k) { SC[SC_CM + k] = cm[k]; SC[SC_CF + k] = cf[k]; }
    SC[SC_ND] = nd;
  }
  double acc[2] = {0.0, 0.0};
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < F.n; i += gridDim.x * kBlock) {
    double p0, p1, p2;
    load_from(F.ps, i, p0, p1, p2);
    p0 -= cm[0]; p1 -= cm[1]; p2 -= cm[2];
    acc[0] += sqrt((p0 * p0 + p1 * p1) + p2 * p2);
  }
  for (int j = blockIdx.x * kBlock + threadIdx.x; j < nt; j += gridDim.x * kBlock) {
    if (F.flag[j]) {
      const double *q = to_rec + (size_t)j * kRecStride;
      double q0 = q[0] - cf[0], q1 = q[1] - cf[1], q2 = q[2] - cf[2];
      acc[1] += sqrt((q0 * q0 + q1 * q1) + q2 * q2);
    }
  }
  block_reduce_store<2>(acc, P3);
}

struct WeightOpts {
  int rrl, normalize, strength_f, strength_b;
  int mode_f, mode_b;       // LRR_ICP_SCALE_*; MUSE arrives here as an injected value
  double inj_f, inj_b;
};

__device__ __forceinline__ NormD norm_from_scalars(const double *SC, bool flipped) {
  NormD nm;
  if (!flipped) {
    for (int k = 0; k < 3; ++k) { nm.cm[k] = SC[SC_CM + k]; nm.cf[k] = SC[SC_CF + k]; }
    nm.rm = SC[SC_RM]; nm.rf = SC[SC_RF];
  } else {  // backward list: flipped centres and radii (rrl sym .txx:719-720)
    for (int k = 0; k < 3; ++k) { nm.cm[k] = SC[SC_CF + k]; nm.cf[k] = SC[SC_CM + k]; }
    nm.rm = SC[SC_RF]; nm.rf = SC[SC_RM];
  }
  return nm;
}

// F4: Beaton-Tukey weights, strength gain and spatial keys.  The multiplications by 1/MADscale
// are deferred to F6, in the reference's order: w = BT/s^2 ; w *= g ; w *= 1/gs ; w *= sw ; w *= 1/ss.
__global__ __launch_bounds__(kBlock) void k_f_weights(DirArgs F, DirArgs B, int nbF, int nbB, WeightOpts o,
                                                      const double *P1, const double *P3, int nb3,
                                                      double *SC, double *P4) {
  const bool isF = (int)blockIdx.x < nbF;
  const DirArgs &D = isF ? F : B;
  const int lb = isF ? blockIdx.x : blockIdx.x - nbF, nb = isF ? nbF : nbB;
  const int r0 = isF ? 0 : nbF, r1 = isF ? nbF : nbF + nbB;
  const int mode = isF ? o.mode_f : o.mode_b;
  double scale;
  if (mode == LRR_ICP_SCALE_RMS) {
    double S = block_total(P1, r0, r1, 5, 0), W = block_total(P1, r0, r1, 5, 1);
    scale = sqrt(S / W);
  } else {
    scale = isF ? o.inj_f : o.inj_b;
  }
  __shared__ double sc_norm[8];
  if (o.normalize) {
    double rm = block_total(P3, 0, nb3, 2, 0) / (double)F.n;
    double rf = block_total(P3, 0, nb3, 2, 1) / SC[SC_ND];
    if (threadIdx.x == 0) {
      for (int k = 0; k < 3; ++k) { sc_norm[k] = SC[SC_CM + k]; sc_norm[4 + k] = SC[SC_CF + k]; }
      sc_norm[3] = rm; sc_norm[7] = rf;
      if (blockIdx.x == 0) { SC[SC_RM] = rm; SC[SC_RF] = rf; }
    }
  } else if (threadIdx.x == 0) {
    for (int k = 0; k < 8; ++k) sc_norm[k] = (k == 3 || k == 7) ? 1.0 : 0.0;
    if (blockIdx.x == 0) for (int k = 0; k < 8; ++k) SC[k] = sc_norm[k];
  }
  if (lb == 0 && threadIdx.x == 0) SC[isF ? SC_SIGF : SC_SIGB] = scale;
  __syncthreads();
  NormD nm = norm_from_scalars(sc_norm, !isF);
  const bool clone = !o.rrl && !isF;                 // cdcl: backward matches enter as role-reversed clones
  const bool strength = o.rrl ? (isF ? o.strength_f : o.strength_b) : o.strength_f;
  double *wdst = clone ? D.wc : D.w;
  const double scale2 = scale * scale;
  double acc[1] = {0.0};
  for (int i = lb * kBlock + threadIdx.x; i < D.n; i += nb * kBlock) {
    double u = D.r2[i] / scale2, bt = 0.0;
    if (u <= 16.0) { double temp = 1 - u * 0.0625; bt = temp * temp; }
    double wi = bt / scale2;
    if (clone) D.w[i] = wi;  // the backward list itself keeps its Beaton-Tukey weights (cdcl sym .txx:665)
    double key = INFINITY;
    if (strength && wi != 0) {
      double sp = D.ps.from_rec[(size_t)D.ps.m_from[i] * kRecStride + 9];
      double sq = D.ps.to_rec[(size_t)D.ps.m_to[i] * kRecStride + 9];
      double g = D.float_product ? (double)((float)sp * (float)sq) : sp * sq;
      if (g > 90000.0) g = 90000.0;
      wi = wi * g;
      key = g;
      acc[0] += 1.0;
    }
    D.key_g[i] = key;
    double q0, q1, q2;
    if (clone) {  // the clone's to_ is this list's from_
      load_from(D.ps, i, q0, q1, q2);
      q0 = (q0 - nm.cm[0]) / nm.rm; q1 = (q1 - nm.cm[1]) / nm.rm; q2 = (q2 - nm.cm[2]) / nm.rm;
    } else {
      const double *q = D.ps.to_rec + (size_t)D.ps.m_to[i] * kRecStride;
      q0 = (q[0] - nm.cf[0]) / nm.rf; q1 = (q[1] - nm.cf[1]) / nm.rf; q2 = (q[2] - nm.cf[2]) / nm.rf;
    }
    double d = sqrt((q0 * q0 + q1 * q1) + q2 * q2);
    D.key_s[i] = 1.0 / (1.0 + 5.0 * d * d);
    wdst[i] = wi;
  }
  block_reduce_store<1>(acc, P4);
}

__device__ __forceinline__ double mad_factor(const double *sel) {  // 1 / rrel_util_median_abs_dev_scale
  double m = sel[3];
  double gs = m > 1.0 ? 1.4826 * (1 + 5.0 / (m - 1.0)) * sel[0] : 0.0;
  return 1.0 / gs;
}

// F6: final weights and the normal equations of both lists (backward list role-reversed)
template <int DOF, bool PREMUL>
__global__ __launch_bounds__(kBlock) void k_f_accumulate(DirArgs F, DirArgs B, int nbF, int nbB, WeightOpts o,
                                                         const double *SC, const double *SEL, double c0,
                                                         double c1, double c2, int rigid_increment,
                                                         double *P6) {
  constexpr int NT = Tri<DOF>::N;
  const bool isF = (int)blockIdx.x < nbF;
  const DirArgs &D = isF ? F : B;
  const int lb = isF ? blockIdx.x : blockIdx.x - nbF, nb = isF ? nbF : nbB;
  NormD nm = norm_from_scalars(SC, !isF);
  const bool clone = !o.rrl && !isF;
  const bool strength = o.rrl ? (isF ? o.strength_f : o.strength_b) : o.strength_f;
  // selection slots: rrl {F.g, B.g, F.s, B.s}; cdcl {merged g, merged s}
  const double fg = strength ? mad_factor(SEL + 4 * (o.rrl ? (isF ? 0 : 1) : 0)) : 1.0;
  const double fs = mad_factor(SEL + 4 * (o.rrl ? (isF ? 2 : 3) : 1));
  double *wbuf = clone ? D.wc : D.w;
  double acc[NT + DOF];
#pragma unroll
  for (int k = 0; k < NT + DOF; ++k) acc[k] = 0.0;
  for (int i = lb * kBlock + threadIdx.x; i < D.n; i += nb * kBlock) {
    double wi = wbuf[i];
    if (strength) wi = wi * fg;
    wi = wi * D.key_s[i];
    wi = wi * fs;
    wbuf[i] = wi;
    if (PREMUL && wi < 1e-4) continue;
    double jx, jy, jz, b0, b1, b2, P[6];
    load_oriented(D.ps, i, nm, isF ? 0 : 1, jx, jy, jz, b0, b1, b2, P);
    accumulate_pair<DOF, PREMUL>(acc, wi, jx, jy, jz, b0, b1, b2, P, c0, c1, c2, rigid_increment);
  }
  block_reduce_store<NT + DOF>(acc, P6);
}

// ---------------------------------------------------------------------------------------------
// the same 6-pass MSB radix select spread over the chip: per pass one histogram kernel (many
// blocks per job, LDS-private histograms flushed with integer atomics: order-independent, so
// still deterministic) and one pick kernel (one block per job scans the 2048 bins, narrows the
// prefix and clears the histogram).  Used for the MAD scales, where only the k-th value is needed.
// ---------------------------------------------------------------------------------------------
constexpr int kSelHistBlocks = 64;   // blocks per job and pass

__global__ __launch_bounds__(kBlock) void k_sel_init(const SelJob *jobs, SelState *st, int *hist) {
  SelJob job = jobs[blockIdx.x];
  long long m = job.m_fixed;
  if (job.cnt) {
    double tot = block_total(job.cnt, job.cnt_a0, job.cnt_a1, 1, 0) + block_total(job.cnt, job.cnt_b0, job.cnt_b1, 1, 0);
    m = (long long)tot;
  }
  long long k = job.k;
  if (k < 0) k = m > 1 ? (m - 1) / 2 + 1 : -1;
  if (k >= (long long)job.n0 + job.n1) k = -1;
  for (int b = threadIdx.x; b < 2048; b += kBlock) hist[blockIdx.x * 2048 + b] = 0;
  if (threadIdx.x == 0) { SelState s0; s0.prefix = 0; s0.mask = 0; s0.k = k; s0.m = m; st[blockIdx.x] = s0; }
}

__global__ __launch_bounds__(kBlock) void k_sel_hist(const SelJob *jobs, const SelState *st, int *hist, int shift, int bins) {
  __shared__ int lh[2048];
  SelJob job = jobs[blockIdx.y];
  SelState s0 = st[blockIdx.y];
  if (s0.k < 0) return;
  for (int b = threadIdx.x; b < 2048; b += kBlock) lh[b] = 0;
  __syncthreads();
  int n = job.n0 + job.n1;
  int per = (n + gridDim.x - 1) / gridDim.x;
  int lo = blockIdx.x * per, hi = min(lo + per, n);
  int lane = threadIdx.x & 63;
  for (int base = lo; base < hi; base += kBlock) {
    int i = base + threadIdx.x;
    bool act = false;
    int dgt = 0;
    if (i < hi) {
      unsigned long long key = job_key(job, i);
      act = (key & s0.mask) == s0.prefix;
      dgt = (int)((key >> shift) & (unsigned long long)(bins - 1));
    }
#pragma unroll
    for (int round = 0; round < 2; ++round) {
      unsigned long long mm = __ballot(act);
      if (mm == 0ull) break;
      int leader = __ffsll((long long)mm) - 1;
      int d0 = __shfl(dgt, leader, 64);
      unsigned long long same = __ballot(act && dgt == d0);
      if (lane == leader) atomicAdd(&lh[d0], (int)__popcll(same));
      if (dgt == d0) act = false;
    }
    if (act) atomicAdd(&lh[dgt], 1);
  }
  __syncthreads();
  for (int b = threadIdx.x; b < bins; b += kBlock) {
    int v = lh[b];
    if (v) atomicAdd(&hist[blockIdx.y * 2048 + b], v);
  }
}

__global__ __launch_bounds__(kSelThreads) void k_sel_pick(const SelJob *jobs, SelState *st, int *hist, int shift, int bins, int last) {
  __shared__ int wave_tot[kSelThreads / 64];
  __shared__ int sel_bin, sel_before;
  SelJob job = jobs[blockIdx.x];
  SelState s0 = st[blockIdx.x];
  int *h = hist + blockIdx.x * 2048;
  if (s0.k < 0) {
    if (last && threadIdx.x == 0) { job.out[0] = 0.0; job.out[1] = 0.0; job.out[2] = 0.0; job.out[3] = (double)s0.m; }
    return;
  }
  int h0 = h[2 * threadIdx.x], h1 = h[2 * threadIdx.x + 1];
  int s = h0 + h1, incl = s;
  int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    int v = __shfl_up(incl, off, 64);
    if (lane >= off) incl += v;
  }
  if (lane == 63) wave_tot[wv] = incl;
  __syncthreads();
  int before = 0;
  for (int w2 = 0; w2 < wv; ++w2) before += wave_tot[w2];
  int excl = before + incl - s;
  long long k = s0.k;
  if (s > 0 && k >= excl && k < excl + s) {
    if (k < excl + h0) { sel_bin = 2 * threadIdx.x; sel_before = excl; }
    else { sel_bin = 2 * threadIdx.x + 1; sel_before = excl + h0; }
  }
  __syncthreads();
  h[2 * threadIdx.x] = 0; h[2 * threadIdx.x + 1] = 0;  // ready for the next pass
  if (threadIdx.x == 0) {
    s0.prefix |= (unsigned long long)sel_bin << shift;
    s0.mask |= (unsigned long long)(bins - 1) << shift;
    s0.k = k - sel_before;
    st[blockIdx.x] = s0;
    if (last) { job.out[0] = __longlong_as_double((long long)s0.prefix); job.out[1] = 0.0; job.out[2] = 0.0; job.out[3] = (double)s0.m; }
  }
}

__global__ void k_copy_w(const double *src, double *dst, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i];
}

__global__ void k_fill(double *dst, int n, double v) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = v;
}

__global__ void k_widen_idx(const int *src, int n, long long *dst) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i];
}

// ---------------------------------------------------------------------------------------------
// host helpers
// ---------------------------------------------------------------------------------------------
inline int grid_for(int64_t n) {
  int64_t b = (n + kBlock - 1) / kBlock;
  if (b < 1) b = 1;
  return (int)std::min<int64_t>(b, kMaxBlocks);
}
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
  g.gp = c->gp; g.cell_start = c->cell_start;
  g.lox = c->lo[0]; g.loy = c->lo[1]; g.loz = c->lo[2];
  g.h = c->h; g.inv_h = c->inv_h; g.tol = c->tol;
  g.dx = c->dim[0]; g.dy = c->dim[1]; g.dz = c->dim[2];
  g.n = (int)c->n;
  return g;
}

PairSrc pairs_of(const lrr_icp_dir *d) {
  PairSrc ps;
  ps.from_rec = d->from->rec; ps.to_rec = d->to->rec;
  ps.m_from = d->m_from; ps.m_to = d->m_to;
  ps.from_loc = d->from_is_copy ? d->from_loc : nullptr;
  return ps;
}

// finish a block-partials reduction: final sum on device, copy K doubles to host, synchronise
int finish_reduce(lrr_icp_ctx *ctx, int nblocks, int K, double *host_out) {
  hipLaunchKernelGGL(k_reduce_partials, dim3(K), dim3(64), 0, ctx->stream, ctx->d_partials, nblocks, K, ctx->d_final);
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipMemcpyAsync(ctx->h_pinned, ctx->d_final, sizeof(double) * K, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  std::memcpy(host_out, ctx->h_pinned, sizeof(double) * K);
  return LRR_ICP_OK;
}

// k-th values of nj jobs already uploaded to ctx->d_jobs: 1 + 6 x 2 asynchronous launches
int launch_select_kth(lrr_icp_ctx *ctx, int nj) {
  static const int shifts[6] = {53, 42, 31, 20, 9, 0};
  static const int widths[6] = {11, 11, 11, 11, 11, 9};
  hipLaunchKernelGGL(k_sel_init, dim3(nj), dim3(kBlock), 0, ctx->stream, ctx->d_jobs, ctx->d_selstate, ctx->d_selhist);
  HIPCHK(ctx, hipGetLastError());
  for (int pass = 0; pass < 6; ++pass) {
    hipLaunchKernelGGL(k_sel_hist, dim3(kSelHistBlocks, nj), dim3(kBlock), 0, ctx->stream, ctx->d_jobs, ctx->d_selstate, ctx->d_selhist, shifts[pass], 1 << widths[pass]);
    HIPCHK(ctx, hipGetLastError());
    hipLaunchKernelGGL(k_sel_pick, dim3(nj), dim3(kSelThreads), 0, ctx->stream, ctx->d_jobs, ctx->d_selstate, ctx->d_selhist, shifts[pass], 1 << widths[pass], pass == 5 ? 1 : 0);
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
  HIPCHK(ctx, hipMemcpyAsync(ctx->d_jobs, ctx->h_jobs, sizeof(SelJob) * nj, hipMemcpyHostToDevice, ctx->stream));
  if (need_sums) {  // trimmed MUSE sums: single-block kernel that also accumulates the keys below the k-th
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

// rrel_util_median_abs_dev_scale, dof = 1 (cdcl .txx:520-536): k-th of m values, 0 if m <= 1
inline long long mad_rank(long long m) { return (m - 1) / 2 + 1; }
inline double mad_from_kth(long long m, double kth) { return 1.4826 * (1 + 5.0 / (double)(m - 1)) * kth; }

int ensure_dir_capacity(lrr_icp_dir *d, int64_t n) {
  lrr_icp_ctx *ctx = d->ctx;
  if (n <= d->cap) return LRR_ICP_OK;
  int64_t cap = std::max<int64_t>(n, 1024);
  void *ptrs[] = {d->m_from, d->m_to, d->w, d->wc, d->r2, d->key_g, d->key_s, d->from_loc};
  for (void *p : ptrs) if (p) (void)hipFree(p);
  d->cap = 0;
  HIPCHK(ctx, hipMalloc(&d->m_from, sizeof(int) * cap));
  HIPCHK(ctx, hipMalloc(&d->m_to, sizeof(int) * cap));
  HIPCHK(ctx, hipMalloc(&d->w, sizeof(double) * cap));
  HIPCHK(ctx, hipMalloc(&d->wc, sizeof(double) * cap));
  HIPCHK(ctx, hipMalloc(&d->r2, sizeof(double) * cap));
  HIPCHK(ctx, hipMalloc(&d->key_g, sizeof(double) * cap));
  HIPCHK(ctx, hipMalloc(&d->key_s, sizeof(double) * cap));
  HIPCHK(ctx, hipMalloc(&d->from_loc, sizeof(double) * 3 * cap));
  d->cap = cap;
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

// launch the exact NN search of n queries against `to`; results in m_to (and m_from when given)
int launch_match(lrr_icp_ctx *ctx, lrr_icp_cloud *to, const lrr_icp_cloud *from_cloud, const QuerySrc &qs,
                 const XfD &xf, int n, int method, int *m_from, int *m_to, double *w_out) {
  if (n <= 0) return LRR_ICP_OK;
  if (to->n <= 0) {
    HIPCHK(ctx, hipMemsetAsync(m_to, 0xff, sizeof(int) * n, ctx->stream));
    return LRR_ICP_OK;
  }
  GridView g = view_of(to);
  bool prof = ctx->prof && ctx->pe_used + 2 <= (int)ctx->pe.size();
  if (prof) HIPCHK(ctx, hipEventRecord(ctx->pe[ctx->pe_used], ctx->stream));
  if (method == LRR_ICP_NN_BRUTE) {
    int bx = (n + kBlock * kBruteQ - 1) / (kBlock * kBruteQ);
    int S = std::max(1, std::min(64, 1024 / bx));
    int max_chunks = (int)((to->n + kBruteTile - 1) / kBruteTile);
    S = std::min(S, std::max(1, max_chunks));
    int chunk = (int)((to->n + S - 1) / S);
    chunk = ((chunk + kBruteTile - 1) / kBruteTile) * kBruteTile;
    S = (int)((to->n + chunk - 1) / chunk);
    size_t need = (size_t)S * (size_t)n;
    if (need > ctx->brute_cap) {
      if (ctx->d_bd) (void)hipFree(ctx->d_bd);
      if (ctx->d_bi) (void)hipFree(ctx->d_bi);
      ctx->brute_cap = 0;
      HIPCHK(ctx, hipMalloc(&ctx->d_bd, sizeof(double) * need));
      HIPCHK(ctx, hipMalloc(&ctx->d_bi, sizeof(int) * need));
      ctx->brute_cap = need;
    }
    hipLaunchKernelGGL(k_match_brute, dim3(bx, S), dim3(kBlock), 0, ctx->stream, g, qs, xf, n, chunk, m_from, ctx->d_bd, ctx->d_bi);
    HIPCHK(ctx, hipGetLastError());
    hipLaunchKernelGGL(k_brute_merge, dim3(blocks_for(n)), dim3(kBlock), 0, ctx->stream, ctx->d_bd, ctx->d_bi, n, S, m_to, w_out);
    HIPCHK(ctx, hipGetLastError());
  } else {
    if ((size_t)n > ctx->overflow_cap) {
      if (ctx->d_overflow) (void)hipFree(ctx->d_overflow);
      if (ctx->d_hard) (void)hipFree(ctx->d_hard);
      ctx->overflow_cap = 0;
      HIPCHK(ctx, hipMalloc(&ctx->d_overflow, sizeof(int) * (size_t)n));
      HIPCHK(ctx, hipMalloc(&ctx->d_hard, sizeof(int) * (size_t)n));
      ctx->overflow_cap = (size_t)n;
    }
    int *hard_count = ctx->d_overflow_count, *over_count = ctx->d_overflow_count + 1;
    HIPCHK(ctx, hipMemsetAsync(ctx->d_overflow_count, 0, 2 * sizeof(int), ctx->stream));
    // phase 1: rings 0..kRingEasy, one thread per query; unsettled queries -> hard list
    if (from_cloud && qs.use_keys && !qs.order && m_from && w_out && (int64_t)n * 4 >= from_cloud->n) {
      GridView gf = view_of(from_cloud);
      hipLaunchKernelGGL(k_match_grid_coherent, dim3(blocks_for(from_cloud->n)), dim3(kBlock), 0, ctx->stream, g, gf, qs.keys, qs.half, n, xf, m_from, m_to, w_out, ctx->d_hard, hard_count);
    } else {
      hipLaunchKernelGGL(k_match_grid, dim3(blocks_for(n)), dim3(kBlock), 0, ctx->stream, g, qs, xf, n, m_from, m_to, w_out, ctx->d_hard, hard_count);
    }
    HIPCHK(ctx, hipGetLastError());
    // phase 2: one wavefront per hard query; phase 3: brute force for what is still open.
    // Both read their list length on the device and fall through when it is 0.
    int hard_blocks = (int)std::min<int64_t>(4096, std::max<int64_t>(1, ((int64_t)n + 3) / 4));
    hipLaunchKernelGGL(k_match_hard, dim3(hard_blocks), dim3(kBlock), 0, ctx->stream, g, qs, xf, ctx->d_hard, hard_count, m_to, ctx->d_overflow, over_count);
    HIPCHK(ctx, hipGetLastError());
    hipLaunchKernelGGL(k_match_overflow, dim3(256), dim3(kBlock), 0, ctx->stream, g, qs, xf, ctx->d_overflow, over_count, m_to);
    HIPCHK(ctx, hipGetLastError());
  }
  if (prof) {
    HIPCHK(ctx, hipEventRecord(ctx->pe[ctx->pe_used + 1], ctx->stream));
    ctx->pe_used += 2;
  }
  ctx->match_launches += 1;
  return LRR_ICP_OK;
}

int resolve_profile(lrr_icp_ctx *ctx) {
  if (ctx->pe_used == 0) return LRR_ICP_OK;
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  for (int i = 0; i + 1 < ctx->pe_used; i += 2) {
    float ms = 0.f;
    HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->pe[i], ctx->pe[i + 1]));
    ctx->match_ms += ms;
  }
  ctx->pe_used = 0;
  return LRR_ICP_OK;
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
  if ((e = hipHostMalloc(&ctx->h_pinned, sizeof(double) * (kMaxRed + 4 * kMaxSelJobs))) != hipSuccess) return bail(e);
  if ((e = hipMalloc(&ctx->d_jobs, sizeof(SelJob) * kMaxSelJobs)) != hipSuccess) return bail(e);
  if ((e = hipHostMalloc(&ctx->h_jobs, sizeof(SelJob) * kMaxSelJobs)) != hipSuccess) return bail(e);
  if ((e = hipMalloc(&ctx->d_overflow_count, 2 * sizeof(int))) != hipSuccess) return bail(e);
  if ((e = hipMalloc(&ctx->d_fused, sizeof(double) * kFusedDoubles)) != hipSuccess) return bail(e);
  if ((e = hipMalloc(&ctx->d_selstate, sizeof(SelState) * kMaxSelJobs)) != hipSuccess) return bail(e);
  if ((e = hipMalloc(&ctx->d_selhist, sizeof(int) * kMaxSelJobs * 2048)) != hipSuccess) return bail(e);
  if ((e = hipHostMalloc(&ctx->h_fused, sizeof(double) * 256)) != hipSuccess) return bail(e);
  *out = ctx;
  return LRR_ICP_OK;
}

void lrr_icp_destroy(lrr_icp_ctx *ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  for (hipEvent_t ev : ctx->pe) (void)hipEventDestroy(ev);
  if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
  if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
  if (ctx->d_partials) (void)hipFree(ctx->d_partials);
  if (ctx->d_final) (void)hipFree(ctx->d_final);
  if (ctx->h_pinned) (void)hipHostFree(ctx->h_pinned);
  if (ctx->d_jobs) (void)hipFree(ctx->d_jobs);
  if (ctx->h_jobs) (void)hipHostFree(ctx->h_jobs);
  if (ctx->d_bd) (void)hipFree(ctx->d_bd);
  if (ctx->d_bi) (void)hipFree(ctx->d_bi);
  if (ctx->d_overflow) (void)hipFree(ctx->d_overflow);
  if (ctx->d_hard) (void)hipFree(ctx->d_hard);
  if (ctx->d_overflow_count) (void)hipFree(ctx->d_overflow_count);
  if (ctx->d_order) (void)hipFree(ctx->d_order);
  if (ctx->d_fused) (void)hipFree(ctx->d_fused);
  if (ctx->d_selstate) (void)hipFree(ctx->d_selstate);
  if (ctx->d_selhist) (void)hipFree(ctx->d_selhist);
  if (ctx->h_fused) (void)hipHostFree(ctx->h_fused);
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
  if (reset) { ctx->match_ms = 0.0; ctx->match_launches = 0; }
  if (enable && ctx->pe.empty()) {
    ctx->pe.resize(kProfEvents);
    for (auto &ev : ctx->pe) HIPCHK(ctx, hipEventCreate(&ev));
  }
  ctx->prof = enable != 0;
  return LRR_ICP_OK;
}

// ---- clouds ----------------------------------------------------------------------------------
int lrr_icp_cloud_create(lrr_icp_ctx *ctx, int64_t n, const double *loc, const double *proj,
                         const double *strength, int float_strength, lrr_icp_cloud **out) {
  if (!ctx || !out || n < 0 || n > 0x3fffffff || (n > 0 && (!loc || !proj))) return fail(ctx, LRR_ICP_ERR_ARG, "cloud_create: bad argument");
  *out = nullptr;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  // records; the error projector must be symmetric (nn^T, I - tt^T, I: features/itkFeatureImageFilter.txx:1053-1072)
  std::vector<double> rec((size_t)std::max<int64_t>(n, 1) * kRecStride);
  for (int64_t i = 0; i < n; ++i) {
    const double *P = proj + 9 * i;
    if (P[1] != P[3] || P[2] != P[6] || P[5] != P[7]) return fail(ctx, LRR_ICP_ERR_ARG, "cloud_create: error_projector_ must be symmetric");
    double *r = &rec[(size_t)i * kRecStride];
    r[0] = loc[3 * i]; r[1] = loc[3 * i + 1]; r[2] = loc[3 * i + 2];
    r[3] = P[0]; r[4] = P[1]; r[5] = P[2]; r[6] = P[4]; r[7] = P[5]; r[8] = P[8];
    double s = strength ? strength[i] : 0.0;
    r[9] = float_strength ? (double)(float)s : s;
  }
  lrr_icp_cloud *c = new (std::nothrow) lrr_icp_cloud;
  if (!c) return fail(ctx, LRR_ICP_ERR_HIP, "out of host memory");
  c->ctx = ctx; c->n = n; c->float_strength = float_strength;
  // uniform grid, ~2 points per cell; counting sort by cell id
  double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
  for (int64_t i = 0; i < n; ++i)
    for (int d = 0; d < 3; ++d) {
      double v = loc[3 * i + d];
      if (!std::isfinite(v)) { delete c; return fail(ctx, LRR_ICP_ERR_ARG, "cloud_create: non-finite location"); }
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
    double ppc = 0.5;  // target points per cell (surfaces make occupied cells far denser)
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
  std::vector<int> cell((size_t)std::max<int64_t>(n, 1)), start((size_t)ncell + 1, 0);
  for (int64_t i = 0; i < n; ++i) {
    int cc[3];
    for (int d = 0; d < 3; ++d) {
      int v = (int)std::floor((loc[3 * i + d] - lo[d]) * c->inv_h);
      cc[d] = v < 0 ? 0 : (v > dim[d] - 1 ? dim[d] - 1 : v);
    }
    int id = (cc[2] * dim[1] + cc[1]) * dim[0] + cc[0];
    cell[(size_t)i] = id;
    start[(size_t)id + 1]++;
  }
  for (int64_t k = 0; k < ncell; ++k) start[(size_t)k + 1] += start[(size_t)k];
  std::vector<int> cursor(start.begin(), start.end() - 1);
  std::vector<GridPoint> gp((size_t)std::max<int64_t>(n, 1));
  for (int64_t i = 0; i < n; ++i) {  // ascending i: original order kept inside a cell
    int pos = cursor[(size_t)cell[(size_t)i]]++;
    gp[(size_t)pos] = GridPoint{loc[3 * i], loc[3 * i + 1], loc[3 * i + 2], (long long)i};
  }
  auto cleanup = [&](int code, const std::string &msg) { lrr_icp_cloud_destroy(c); return fail(ctx, code, msg); };
  size_t m = (size_t)std::max<int64_t>(n, 1);
  if (hipMalloc(&c->rec, sizeof(double) * m * kRecStride) != hipSuccess ||
      hipMalloc(&c->gp, sizeof(GridPoint) * m) != hipSuccess ||
      hipMalloc(&c->cell_start, sizeof(int) * ((size_t)ncell + 1)) != hipSuccess)
    return cleanup(LRR_ICP_ERR_HIP, "cloud_create: hipMalloc failed");
  if (hipMemcpy(c->rec, rec.data(), sizeof(double) * m * kRecStride, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(c->gp, gp.data(), sizeof(GridPoint) * m, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(c->cell_start, start.data(), sizeof(int) * ((size_t)ncell + 1), hipMemcpyHostToDevice) != hipSuccess)
    return cleanup(LRR_ICP_ERR_HIP, "cloud_create: upload failed");
  *out = c;
  return LRR_ICP_OK;
}

void lrr_icp_cloud_destroy(lrr_icp_cloud *c) {
  if (!c) return;
  if (c->ctx) (void)hipSetDevice(c->ctx->device);
  void *ptrs[] = {c->rec, c->gp, c->cell_start};
  for (void *p : ptrs) if (p) (void)hipFree(p);
  delete c;
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
  *out = d;
  return LRR_ICP_OK;
}

void lrr_icp_dir_destroy(lrr_icp_dir *d) {
  if (!d) return;
  if (d->ctx) { (void)hipSetDevice(d->ctx->device); (void)hipStreamSynchronize(d->ctx->stream); }
  void *ptrs[] = {d->m_from, d->m_to, d->w, d->wc, d->r2, d->key_g, d->key_s, d->from_loc, d->flag};
  for (void *p : ptrs) if (p) (void)hipFree(p);
  delete d;
}

int lrr_icp_dir_set_from(lrr_icp_dir *d, lrr_icp_cloud *from) {
  if (!d || !from) return LRR_ICP_ERR_ARG;
  d->from = from;
  d->n = 0;
  d->from_is_copy = 0;
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

int lrr_icp_match(lrr_icp_dir *d, const lrr_icp_xform *xf, const uint32_t keys[4], const int64_t *order,
                  int64_t n_points, int method) {
  if (!d || !xf || n_points < 0) return LRR_ICP_ERR_ARG;
  lrr_icp_ctx *ctx = d->ctx;
  if (n_points > d->from->n) return fail(ctx, LRR_ICP_ERR_ARG, "match: n_points exceeds the from-cloud");
  if (!keys && !order && n_points > 0) return fail(ctx, LRR_ICP_ERR_ARG, "match: need keys or an explicit order");
  if (n_points > 0 && d->to->n <= 0) return fail(ctx, LRR_ICP_ERR_ARG, "match: empty to-cloud");
  for (int k = 0; k < 9; ++k) if (!std::isfinite(xf->A[k])) return fail(ctx, LRR_ICP_ERR_ARG, "match: non-finite transform");
  for (int k = 0; k < 3; ++k) if (!std::isfinite(xf->t[k]) || !std::isfinite(xf->c[k])) return fail(ctx, LRR_ICP_ERR_ARG, "match: non-finite transform");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  int rc = ensure_dir_capacity(d, n_points);
  if (rc) return rc;
  d->from_is_copy = 0;
  d->n = n_points;
  if (n_points == 0) return LRR_ICP_OK;
  QuerySrc qs;
  qs.rec = d->from->rec; qs.stride = kRecStride; qs.order = nullptr; qs.use_keys = 0;
  qs.nsrc = (int)d->from->n; qs.half = perm_half(d->from->n);
  std::memset(&qs.keys, 0, sizeof qs.keys);
  if (order) {
    if ((size_t)n_points > ctx->order_cap) {
      if (ctx->d_order) (void)hipFree(ctx->d_order);
      ctx->order_cap = 0;
      HIPCHK(ctx, hipMalloc(&ctx->d_order, sizeof(int) * (size_t)n_points));
      ctx->order_cap = (size_t)n_points;
    }
    std::vector<int> o32((size_t)n_points);
    for (int64_t i = 0; i < n_points; ++i) {
      if (order[i] < 0 || order[i] >= d->from->n) return fail(ctx, LRR_ICP_ERR_ARG, "match: order index out of range");
      o32[(size_t)i] = (int)order[i];
    }
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_order, o32.data(), sizeof(int) * (size_t)n_points, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // o32 is pageable and dies at scope exit
    qs.order = ctx->d_order;
  } else {
    qs.use_keys = 1;
    for (int r = 0; r < 4; ++r) qs.keys.k[r] = keys[r];
  }
  if (method == LRR_ICP_NN_AUTO) method = LRR_ICP_NN_GRID;
  // the match kernels also set every weight to 1.0 (cdcl .txx:645)
  return launch_match(ctx, d->to, d->from, qs, to_dev(xf), (int)n_points, method, d->m_from, d->m_to, d->w);
}

int lrr_icp_premap(lrr_icp_dir *d, const lrr_icp_xform *G) {
  if (!d || !G) return LRR_ICP_ERR_ARG;
  lrr_icp_ctx *ctx = d->ctx;
  if (d->n == 0) { d->from_is_copy = 1; return LRR_ICP_OK; }
  HIPCHK(ctx, hipSetDevice(ctx->device));
  PairSrc ps = pairs_of(d);
  // in place is safe: each thread reads its own 3 doubles before writing them
  hipLaunchKernelGGL(k_premap, dim3(blocks_for(d->n)), dim3(kBlock), 0, ctx->stream, ps, to_dev(G), (int)d->n, d->from_loc);
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
  int nb = grid_for(d->n);
  hipLaunchKernelGGL(k_residual, dim3(nb), dim3(kBlock), 0, ctx->stream, pairs_of(d), to_dev(xf), to_dev(norm), (int)d->n, d->w, d->r2, d->key_s, ctx->d_partials);
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

int lrr_icp_muse_scale(lrr_icp_dir *d, double *sigma) {
  if (!d || !sigma) return LRR_ICP_ERR_ARG;
  lrr_icp_ctx *ctx = d->ctx;
  *sigma = 0.0;
  int64_t n = d->n;
  if (n <= 0) return LRR_ICP_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  // rrel_muset_obj(0,false): trimmed MUSE over inlier fractions 0.25..0.95 step 0.05
  struct Cand { int64_t k; double ek, sd, div; };
  std::vector<Cand> cands;
  std::vector<SelJob> jobs;
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
    SelJob j{}; j.k0 = d->key_s; j.n0 = (int)n; j.k1 = nullptr; j.n1 = 0; j.k = k - 1; j.out = nullptr; j.cnt = nullptr; j.m_fixed = n;
    jobs.push_back(j);
  }
  std::vector<double> res(4 * jobs.size() + 4);
  int rc = run_select(ctx, jobs, res.data(), true);
  if (rc) return rc;
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
  *sigma = best_sk;
  return LRR_ICP_OK;
}

int lrr_icp_assign_weights(lrr_icp_dir *d, double sigma) {
  if (!d) return LRR_ICP_ERR_ARG;
  lrr_icp_ctx *ctx = d->ctx;
  if (d->n == 0) return LRR_ICP_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  hipLaunchKernelGGL(k_assign, dim3(blocks_for(d->n)), dim3(kBlock), 0, ctx->stream, d->r2, sigma, (int)d->n, d->w);
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
  int nb = grid_for(n);
  hipLaunchKernelGGL(k_from_sum, dim3(nb), dim3(kBlock), 0, ctx->stream, ps, (int)n, ctx->d_partials);
  HIPCHK(ctx, hipGetLastError());
  int rc = finish_reduce(ctx, nb, 3, r);
  if (rc) return rc;
  for (int k = 0; k < 3; ++k) out->cm[k] = r[k] / (double)n;
  HIPCHK(ctx, hipMemsetAsync(d->flag, 0, sizeof(unsigned) * d->flag_cap, ctx->stream));
  if (n > 0) {
    hipLaunchKernelGGL(k_mark, dim3(blocks_for(n)), dim3(kBlock), 0, ctx->stream, d->m_to, (int)n, d->flag);
    HIPCHK(ctx, hipGetLastError());
  }
  int nbt = grid_for(nt);
  hipLaunchKernelGGL(k_to_sum, dim3(nbt), dim3(kBlock), 0, ctx->stream, d->to->rec, d->flag, (int)nt, ctx->d_partials);
  HIPCHK(ctx, hipGetLastError());
  rc = finish_reduce(ctx, nbt, 4, r);
  if (rc) return rc;
  double nd = r[3];
  for (int k = 0; k < 3; ++k) out->cf[k] = r[k] / nd;
  int nb2 = std::max(nb, nbt);
  hipLaunchKernelGGL(k_radii, dim3(nb2), dim3(kBlock), 0, ctx->stream, ps, (int)n, d->to->rec, d->flag, (int)nt,
                     out->cm[0], out->cm[1], out->cm[2], out->cf[0], out->cf[1], out->cf[2], ctx->d_partials);
  HIPCHK(ctx, hipGetLastError());
  rc = finish_reduce(ctx, nb2, 2, r);
  if (rc) return rc;
  out->rm = r[0] / (double)n;
  out->rf = r[1] / nd;
  return LRR_ICP_OK;
}

int lrr_icp_clone_weights(lrr_icp_dir *d) {
  if (!d) return LRR_ICP_ERR_ARG;
  lrr_icp_ctx *ctx = d->ctx;
  if (d->n == 0) return LRR_ICP_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  hipLaunchKernelGGL(k_copy_w, dim3(blocks_for(d->n)), dim3(kBlock), 0, ctx->stream, d->w, d->wc, (int)d->n);
  HIPCHK(ctx, hipGetLastError());
  return LRR_ICP_OK;
}

int lrr_icp_weight_by_strength(lrr_icp_dir *a, lrr_icp_dir *b, double *grad_scale) {
  if (!a) return LRR_ICP_ERR_ARG;
  lrr_icp_ctx *ctx = a->ctx;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  double cnt_a = 0.0, cnt_b = 0.0;
  int rc;
  if (a->n > 0) {
    int nb = grid_for(a->n);
    hipLaunchKernelGGL(k_strength, dim3(nb), dim3(kBlock), 0, ctx->stream, pairs_of(a), (int)a->n, a->from->float_strength, a->w, a->key_g, ctx->d_partials);
    HIPCHK(ctx, hipGetLastError());
    if ((rc = finish_reduce(ctx, nb, 1, &cnt_a))) return rc;
  }
  if (b && b->n > 0) {
    int nb = grid_for(b->n);
    hipLaunchKernelGGL(k_strength, dim3(nb), dim3(kBlock), 0, ctx->stream, pairs_of(b), (int)b->n, b->from->float_strength, b->wc, b->key_g, ctx->d_partials);
    HIPCHK(ctx, hipGetLastError());
    if ((rc = finish_reduce(ctx, nb, 1, &cnt_b))) return rc;
  }
  long long m = (long long)(cnt_a + cnt_b);
  double gs = 0.0;
  if (m > 1) {
    // masked entries carry +inf keys and sort last, so rank k among the m finite keys is rank k overall
    std::vector<SelJob> jobs(1);
    jobs[0] = SelJob{};
    jobs[0].k0 = a->key_g; jobs[0].n0 = (int)a->n;
    jobs[0].k1 = b ? b->key_g : nullptr; jobs[0].n1 = b ? (int)b->n : 0;
    jobs[0].k = mad_rank(m); jobs[0].out = nullptr; jobs[0].cnt = nullptr; jobs[0].m_fixed = m;
    double res[4];
    if ((rc = run_select(ctx, jobs, res))) return rc;
    gs = mad_from_kth(m, res[0]);
  }
  double f = 1.0 / gs;
  if (a->n > 0) { hipLaunchKernelGGL(k_scale_w, dim3(blocks_for(a->n)), dim3(kBlock), 0, ctx->stream, a->w, (int)a->n, f); HIPCHK(ctx, hipGetLastError()); }
  if (b && b->n > 0) { hipLaunchKernelGGL(k_scale_w, dim3(blocks_for(b->n)), dim3(kBlock), 0, ctx->stream, b->wc, (int)b->n, f); HIPCHK(ctx, hipGetLastError()); }
  if (grad_scale) *grad_scale = gs;
  return LRR_ICP_OK;
}

int lrr_icp_weight_spatially(lrr_icp_dir *a, const lrr_icp_norm *norm_a, lrr_icp_dir *b, const lrr_icp_norm *norm_b, double *spatial_scale) {
  if (!a) return LRR_ICP_ERR_ARG;
  lrr_icp_ctx *ctx = a->ctx;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  if (a->n > 0) {
    hipLaunchKernelGGL(k_spatial, dim3(blocks_for(a->n)), dim3(kBlock), 0, ctx->stream, pairs_of(a), to_dev(norm_a), (int)a->n, 0, a->w, a->key_s);
    HIPCHK(ctx, hipGetLastError());
  }
  if (b && b->n > 0) {
    hipLaunchKernelGGL(k_spatial, dim3(blocks_for(b->n)), dim3(kBlock), 0, ctx->stream, pairs_of(b), to_dev(norm_b), (int)b->n, 1, b->wc, b->key_s);
    HIPCHK(ctx, hipGetLastError());
  }
  long long m = (long long)a->n + (b ? (long long)b->n : 0);
  double ss = 0.0;
  int rc;
  if (m > 1) {
    std::vector<SelJob> jobs(1);
    jobs[0] = SelJob{};
    jobs[0].k0 = a->key_s; jobs[0].n0 = (int)a->n;
    jobs[0].k1 = b ? b->key_s : nullptr; jobs[0].n1 = b ? (int)b->n : 0;
    jobs[0].k = mad_rank(m); jobs[0].out = nullptr; jobs[0].cnt = nullptr; jobs[0].m_fixed = m;
    double res[4];
    if ((rc = run_select(ctx, jobs, res))) return rc;
    ss = mad_from_kth(m, res[0]);
  }
  double f = 1.0 / ss;
  if (a->n > 0) { hipLaunchKernelGGL(k_scale_w, dim3(blocks_for(a->n)), dim3(kBlock), 0, ctx->stream, a->w, (int)a->n, f); HIPCHK(ctx, hipGetLastError()); }
  if (b && b->n > 0) { hipLaunchKernelGGL(k_scale_w, dim3(blocks_for(b->n)), dim3(kBlock), 0, ctx->stream, b->wc, (int)b->n, f); HIPCHK(ctx, hipGetLastError()); }
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
  // 90 fp64 accumulators per thread at dof 12: fewer, longer blocks
  int nb = (int)std::min<int64_t>(std::max<int64_t>(1, (d->n + kBlock - 1) / kBlock), 512);
  PairSrc ps = pairs_of(d);
  NormD nm = to_dev(norm);
  const double *w = (flags & LRR_ICP_ACC_CLONE_W) ? d->wc : d->w;
  int rigid = (flags & LRR_ICP_ACC_RIGID_INCREMENT) ? 1 : 0, rev = (flags & LRR_ICP_ACC_REVERSED) ? 1 : 0;
  bool premul = (flags & LRR_ICP_ACC_PREMUL_SKIP) != 0;
  if (dof == 12 && premul) hipLaunchKernelGGL((k_accumulate<12, true>), dim3(nb), dim3(kBlock), 0, ctx->stream, ps, nm, (int)d->n, w, jac_centre[0], jac_centre[1], jac_centre[2], rigid, rev, ctx->d_partials);
  else if (dof == 12) hipLaunchKernelGGL((k_accumulate<12, false>), dim3(nb), dim3(kBlock), 0, ctx->stream, ps, nm, (int)d->n, w, jac_centre[0], jac_centre[1], jac_centre[2], rigid, rev, ctx->d_partials);
  else if (premul) hipLaunchKernelGGL((k_accumulate<6, true>), dim3(nb), dim3(kBlock), 0, ctx->stream, ps, nm, (int)d->n, w, jac_centre[0], jac_centre[1], jac_centre[2], rigid, rev, ctx->d_partials);
  else hipLaunchKernelGGL((k_accumulate<6, false>), dim3(nb), dim3(kBlock), 0, ctx->stream, ps, nm, (int)d->n, w, jac_centre[0], jac_centre[1], jac_centre[2], rigid, rev, ctx->d_partials);
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

int lrr_icp_errors(lrr_icp_dir *d, const lrr_icp_xform *xf, double *sum_r2, double *sum_wr, double *sum_w) {
  if (!d || !xf) return LRR_ICP_ERR_ARG;
  lrr_icp_ctx *ctx = d->ctx;
  double r[3] = {0.0, 0.0, 0.0};
  if (d->n > 0) {
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int nb = grid_for(d->n);
    hipLaunchKernelGGL(k_errors, dim3(nb), dim3(kBlock), 0, ctx->stream, pairs_of(d), to_dev(xf), to_dev((const lrr_icp_norm *)nullptr), (int)d->n, d->w, ctx->d_partials);
    HIPCHK(ctx, hipGetLastError());
    int rc = finish_reduce(ctx, nb, 3, r);
    if (rc) return rc;
  }
  if (sum_r2) *sum_r2 = r[0];
  if (sum_wr) *sum_wr = r[1];
  if (sum_w) *sum_w = r[2];
  return LRR_ICP_OK;
}

static DirArgs dir_args(lrr_icp_dir *d) {
  DirArgs a;
  a.ps = pairs_of(d);
  a.n = (int)d->n;
  a.w = d->w; a.wc = d->wc; a.r2 = d->r2; a.key_s = d->key_s; a.key_g = d->key_g;
  a.flag = d->flag;
  a.float_product = d->from->float_strength;
  return a;
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
  int nbF = grid_for(F->n), nbB = grid_for(B->n), nt = (int)F->to->n, nbT = grid_for(nt);
  int nb3 = std::max(nbF, nbT);
  NormD ident = to_dev((const lrr_icp_norm *)nullptr);
  // F1
  if (o->normalize) HIPCHK(ctx, hipMemsetAsync(F->flag, 0, sizeof(unsigned) * F->flag_cap, st));
  hipLaunchKernelGGL(k_f_residual, dim3(nbF + nbB), dim3(kBlock), 0, st, aF, aB, nbF, nbB, to_dev(xfF), to_dev(xfB), ident, o->normalize, P1);
  HIPCHK(ctx, hipGetLastError());
  // scale modes; MUSE needs the sorted residuals: its tables and argmin stay on the host (first
  // estimate after set_transform only)
  WeightOpts wo;
  wo.rrl = o->rrl; wo.normalize = o->normalize; wo.strength_f = o->strength_f; wo.strength_b = o->strength_b;
  wo.mode_f = o->scale_mode_f; wo.mode_b = o->scale_mode_b; wo.inj_f = o->injected_f; wo.inj_b = o->injected_b;
  if (wo.mode_f == LRR_ICP_SCALE_MUSE) { int rc = lrr_icp_muse_scale(F, &wo.inj_f); if (rc) return rc; wo.mode_f = LRR_ICP_SCALE_INJECTED; }
  if (wo.mode_b == LRR_ICP_SCALE_MUSE) { int rc = lrr_icp_muse_scale(B, &wo.inj_b); if (rc) return rc; wo.mode_b = LRR_ICP_SCALE_INJECTED; }
  if (o->normalize) {
    hipLaunchKernelGGL(k_to_sum, dim3(nbT), dim3(kBlock), 0, st, F->to->rec, F->flag, nt, P2);
    HIPCHK(ctx, hipGetLastError());
    hipLaunchKernelGGL(k_f_radii, dim3(nb3), dim3(kBlock), 0, st, aF, nbF, P1, P2, nbT, F->to->rec, nt, P3, SC);
    HIPCHK(ctx, hipGetLastError());
  }
  hipLaunchKernelGGL(k_f_weights, dim3(nbF + nbB), dim3(kBlock), 0, st, aF, aB, nbF, nbB, wo, P1, P3, nb3, SC, P4);
  HIPCHK(ctx, hipGetLastError());
  // order statistics: ranks of the strength jobs come from the device-side counts
  SelJob *hj = ctx->h_jobs;
  int nj = 0;
  auto job = [&](const double *k0, int n0, const double *k1, int n1, bool counted, int a0, int a1, int b0, int b1) {
    SelJob j{};
    j.k0 = k0; j.n0 = n0; j.k1 = k1; j.n1 = n1; j.k = -1; j.out = SEL + 4 * nj;
    j.cnt = counted ? P4 : nullptr; j.cnt_a0 = a0; j.cnt_a1 = a1; j.cnt_b0 = b0; j.cnt_b1 = b1;
    j.m_fixed = (long long)n0 + n1;
    hj[nj++] = j;
  };
  if (o->rrl) {
    job(F->key_g, aF.n, nullptr, 0, true, 0, nbF, 0, 0);
    job(B->key_g, aB.n, nullptr, 0, true, nbF, nbF + nbB, 0, 0);
    job(F->key_s, aF.n, nullptr, 0, false, 0, 0, 0, 0);
    job(B->key_s, aB.n, nullptr, 0, false, 0, 0, 0, 0);
  } else {
    job(F->key_g, aF.n, B->key_g, aB.n, true, 0, nbF, nbF, nbF + nbB);
    job(F->key_s, aF.n, B->key_s, aB.n, false, 0, 0, 0, 0);
  }
  HIPCHK(ctx, hipMemcpyAsync(ctx->d_jobs, hj, sizeof(SelJob) * nj, hipMemcpyHostToDevice, st));
  { int rc = launch_select_kth(ctx, nj); if (rc) return rc; }
  // F6 + final reduction
  int nbAF = std::min(nbF, 256), nbAB = std::min(nbB, 256);
  const double *c = o->jac_centre;
  const int nt2 = o->dof * (o->dof + 1) / 2, K = nt2 + o->dof;
  if (o->dof == 12 && o->rrl) hipLaunchKernelGGL((k_f_accumulate<12, true>), dim3(nbAF + nbAB), dim3(kBlock), 0, st, aF, aB, nbAF, nbAB, wo, SC, SEL, c[0], c[1], c[2], o->rigid_increment, P6);
  else if (o->dof == 12) hipLaunchKernelGGL((k_f_accumulate<12, false>), dim3(nbAF + nbAB), dim3(kBlock), 0, st, aF, aB, nbAF, nbAB, wo, SC, SEL, c[0], c[1], c[2], o->rigid_increment, P6);
  else if (o->rrl) hipLaunchKernelGGL((k_f_accumulate<6, true>), dim3(nbAF + nbAB), dim3(kBlock), 0, st, aF, aB, nbAF, nbAB, wo, SC, SEL, c[0], c[1], c[2], o->rigid_increment, P6);
  else hipLaunchKernelGGL((k_f_accumulate<6, false>), dim3(nbAF + nbAB), dim3(kBlock), 0, st, aF, aB, nbAF, nbAB, wo, SC, SEL, c[0], c[1], c[2], o->rigid_increment, P6);
  HIPCHK(ctx, hipGetLastError());
  hipLaunchKernelGGL(k_reduce_partials, dim3(K), dim3(64), 0, st, P6, nbAF + nbAB, K, FIN);
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipMemcpyAsync(ctx->h_fused, OUT, sizeof(double) * (FIN_OFF + K), hipMemcpyDeviceToHost, st));
  HIPCHK(ctx, hipStreamSynchronize(st));
  const double *h = ctx->h_fused;
  for (int k = 0; k < 3; ++k) { out->norm.cm[k] = h[SC_CM + k]; out->norm.cf[k] = h[SC_CF + k]; }
  out->norm.rm = h[SC_RM]; out->norm.rf = h[SC_RF];
  out->sigma_f = h[SC_SIGF]; out->sigma_b = h[SC_SIGB];
  auto mad = [&](int slot) { const double *sl = h + SEL_OFF + 4 * slot; return sl[3] > 1.0 ? 1.4826 * (1 + 5.0 / (sl[3] - 1.0)) * sl[0] : 0.0; };
  if (o->rrl) { out->grad_scale[0] = mad(0); out->grad_scale[1] = mad(1); out->spatial_scale[0] = mad(2); out->spatial_scale[1] = mad(3); }
  else { out->grad_scale[0] = out->grad_scale[1] = mad(0); out->spatial_scale[0] = out->spatial_scale[1] = mad(1); }
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
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  std::vector<int> tmp((size_t)n);
  if (from_idx) {
    HIPCHK(ctx, hipMemcpy(tmp.data(), d->m_from, sizeof(int) * n, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < n; ++i) from_idx[i] = tmp[(size_t)i];
  }
  if (to_idx) {
    HIPCHK(ctx, hipMemcpy(tmp.data(), d->m_to, sizeof(int) * n, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < n; ++i) to_idx[i] = tmp[(size_t)i];
  }
  if (w) HIPCHK(ctx, hipMemcpy(w, d->w, sizeof(double) * n, hipMemcpyDeviceToHost));
  return LRR_ICP_OK;
}

int lrr_icp_nearest(lrr_icp_cloud *cloud, const double *query, int64_t nq, int method, int64_t *idx) {
  if (!cloud || nq < 0 || (nq > 0 && (!query || !idx))) return LRR_ICP_ERR_ARG;
  lrr_icp_ctx *ctx = cloud->ctx;
  if (nq == 0) return LRR_ICP_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  double *dq = nullptr;
  int *dto = nullptr;
  HIPCHK(ctx, hipMalloc(&dq, sizeof(double) * 3 * nq));
  if (hipMalloc(&dto, sizeof(int) * nq) != hipSuccess) { (void)hipFree(dq); return fail(ctx, LRR_ICP_ERR_HIP, "nearest: hipMalloc failed"); }
  int rc = LRR_ICP_OK;
  if (hipMemcpy(dq, query, sizeof(double) * 3 * nq, hipMemcpyHostToDevice) != hipSuccess) rc = fail(ctx, LRR_ICP_ERR_HIP, "nearest: upload failed");
  if (!rc) {
    QuerySrc qs;
    qs.rec = dq; qs.stride = 3; qs.order = nullptr; qs.use_keys = 0; qs.nsrc = (int)nq; qs.half = 1;
    std::memset(&qs.keys, 0, sizeof qs.keys);
    if (method == LRR_ICP_NN_AUTO) method = LRR_ICP_NN_GRID;
    rc = launch_match(ctx, cloud, nullptr, qs, identity_xf(), (int)nq, method, nullptr, dto, nullptr);
  }
  std::vector<int> tmp((size_t)nq);
  if (!rc && (hipStreamSynchronize(ctx->stream) != hipSuccess || hipMemcpy(tmp.data(), dto, sizeof(int) * nq, hipMemcpyDeviceToHost) != hipSuccess))
    rc = fail(ctx, LRR_ICP_ERR_HIP, "nearest: download failed");
  if (!rc) for (int64_t i = 0; i < nq; ++i) idx[i] = tmp[(size_t)i];
  (void)hipFree(dq); (void)hipFree(dto);
  return rc;
}

}  // extern "C"

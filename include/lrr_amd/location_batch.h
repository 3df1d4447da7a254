// location_batch.h — many query locations of itk::LocationRegistration per device submission.
//
// The reference registers its query locations one after the other (rrl/itkLocationRegistration.cxx:2134-2138):
// per location SetupEstimation (:1525-1707: FeaturesInROIs, set_features, set_transform) and then
// `while (!converged && iteration < 70) converged = estimation->one_iteration(0, 0, iteration)` (:2418),
// followed by ComputeDecisionMeasurements (:1816-1871) on the converged estimator.  One location is a few
// thousand matches: far too little for an MI355X.  rrl_location_batch keeps the two volumes resident and
// advances B locations in lockstep; every device phase of an iteration is ONE submission for all of them
// (C-ABI lrr_icp_batch_*).  The host part of an iteration is the same code the one-at-a-time class runs
// (symmetric_schedule, symmetric_apply_solution of icp_engine.h), so per-location results — match sets,
// iteration counts, transforms — equal those of rrl_estimation_symmetric_ICP_matching_all<3,12> driven by
// the caller's loop (tests/test_gpu_batch.py).
#ifndef LRR_AMD_LOCATION_BATCH_H_
#define LRR_AMD_LOCATION_BATCH_H_

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "lrr_amd/icp_engine.h"

namespace lrr_amd {

// one query location: what SetupEstimation derives from the keypoint match (rrl/itkLocationRegistration.cxx:1525-1643)
struct location_query {
  cdcl_trans<3, 12>::sptr transform;                       // initial transform (moving -> fixed); not modified
  vnl_vector_fixed<double, 3> moving_x0, moving_x1;        // ROI of the moving volume
  vnl_vector_fixed<double, 3> fixed_x0, fixed_x1;          // ROI of the fixed volume
  uint64_t seed;                                           // seed of the subset permutations of this location
  location_query() : moving_x0(0.0), moving_x1(0.0), fixed_x0(0.0), fixed_x1(0.0), seed(1) {}
};

// what the caller reads from the estimation object after its loop (rrl/itkLocationRegistration.cxx:2418-2510, 1816-1871)
struct location_result {
  bool enough_features;        // both ROI lists >= min_features (:1632); false: nothing else is set
  bool converged, oscillated, valid_transform;
  unsigned iterations;
  int64_t n_moving_inside, n_fixed_inside, matched_pairs;
  cdcl_trans<3, 12>::sptr forward, backward;   // transforms at the end of the loop (with their covariances);
                                               // backward carries the covariance of estimate_LS_backward(false)
  double initial_RMS, RMS_error;               // RMS_error() after iteration 0 (:2426) and at the end
  double weighted_error, weighted_error_backward;
  double sheet_angles, tube_angles, sheet_angles_backward, tube_angles_backward;  // radians
  double measurements[8];      // ComputeDecisionMeasurements (:1816-1871); NaN without diagnostics
  location_result()
      : enough_features(false), converged(false), oscillated(false), valid_transform(true), iterations(0), n_moving_inside(0),
        n_fixed_inside(0), matched_pairs(0), initial_RMS(-1.0), RMS_error(0.0), weighted_error(0.0), weighted_error_backward(0.0),
        sheet_angles(0.0), tube_angles(0.0), sheet_angles_backward(0.0), tube_angles_backward(0.0) {
    for (int k = 0; k < 8; ++k) measurements[k] = std::nan("");
  }
};

// validTransform (rrl/itkLocationRegistration.cxx:2914-2926): singular values of A within [0.1, 10]
inline bool valid_transform(cdcl_trans<3, 12>::sptr const &T) {
  double U[9], s[3], V[9];
  linalg::svd3(T->get_A().data_block(), U, s, V);
  for (int d = 0; d < 3; ++d) if (s[d] > 10.0 || s[d] < 0.1) return false;
  return true;
}

// ComputeTransferErrorCovariance (rrl/itkLocationRegistration.cxx:2793-2858): J C J^T sampled every `interval`
// mm over [start, end]; mean covariance, largest trace and largest eigenvalue over the samples (the dense
// products of the reference, ~2 us per sample: for tests and callers that want all three)
inline void sampled_transfer_error(cdcl_trans<3, 12>::sptr const &T, vnl_vector_fixed<double, 3> const &start,
                                   vnl_vector_fixed<double, 3> const &end, double interval,
                                   vnl_matrix_fixed<double, 3, 3> &covarianceJ, double &maxTrace, double &maxEval) {
  vnl_matrix_fixed<double, 12, 12> covariance = T->get_covariance();
  covarianceJ.fill(0.0);
  maxTrace = 0.0;
  maxEval = 0.0;
  unsigned count = 0;
  for (double k = start[2]; k <= end[2]; k += interval)
    for (double j = start[1]; j <= end[1]; j += interval)
      for (double i = start[0]; i <= end[0]; i += interval) {
        vnl_vector_fixed<double, 3> point;
        point[0] = i; point[1] = j; point[2] = k;
        vnl_matrix_fixed<double, 3, 12> Jth = T->jacobian_wrt_par(point);
        vnl_matrix_fixed<double, 3, 3> cur = Jth * covariance * Jth.transpose();
        covarianceJ += cur;
        double tr = cur(0, 0) + cur(1, 1) + cur(2, 2);
        if (tr > maxTrace) maxTrace = tr;
        double U[9], s[3], V[9];
        linalg::svd3(cur.data_block(), U, s, V);  // symmetric positive semi-definite: singular values = eigenvalues
        double ev = std::max(s[0], std::max(s[1], s[2]));
        if (ev > maxEval) maxEval = ev;
        ++count;
      }
  if (count) covarianceJ *= (1.0 / double(count));
}

// maxTrace of the same sampling — the only output the decision measurements use (:1832-1858) — without the
// dense products: with the affine Jacobian (row r = (x - c) at columns 3r..3r+2, 1 at column 9+r,
// cdcl_trans_affine.txx:141-161) trace(J C J^T) = l^T M l + 2 b.l + c0, l = x - c, M = sum_r C[3r.., 3r..],
// b = sum_r C[3r.., 9+r], c0 = sum_r C[9+r, 9+r]: 12 multiply-adds per sample instead of ~700
inline double sampled_transfer_max_trace(cdcl_trans<3, 12>::sptr const &T, vnl_vector_fixed<double, 3> const &start,
                                         vnl_vector_fixed<double, 3> const &end, double interval) {
  vnl_matrix_fixed<double, 12, 12> const &C = T->get_covariance();
  double M[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, bv[3] = {0, 0, 0}, c0 = 0.0;
  for (unsigned r = 0; r < 3; ++r) {
    for (unsigned a = 0; a < 3; ++a) {
      for (unsigned b = 0; b < 3; ++b) M[a][b] += C(3 * r + a, 3 * r + b);
      bv[a] += 0.5 * (C(3 * r + a, 9 + r) + C(9 + r, 3 * r + a));
    }
    c0 += C(9 + r, 9 + r);
  }
  const vnl_vector_fixed<double, 3> cen = T->center_moving_;
  double maxTrace = 0.0;
  for (double k = start[2]; k <= end[2]; k += interval)
    for (double j = start[1]; j <= end[1]; j += interval)
      for (double i = start[0]; i <= end[0]; i += interval) {
        const double l[3] = {i - cen[0], j - cen[1], k - cen[2]};
        double tr = c0;
        for (int a = 0; a < 3; ++a) tr += l[a] * ((M[a][0] * l[0] + M[a][1] * l[1]) + M[a][2] * l[2]) + 2.0 * bv[a] * l[a];
        if (tr > maxTrace) maxTrace = tr;
      }
  return maxTrace;
}

// The per-location host work of a lockstep iteration (solve, inverse, un-normalise: a couple of microseconds each,
// independent across locations) spread over a few sleeping worker threads: parallel_for(n, f) runs f(i) for every i
// exactly once, in chunks handed out through an atomic counter, the caller working along.  Workers block on a
// condition variable between calls (no spinning: the device runtime's own threads share the cores).
// LRR_BATCH_THREADS sets the number of workers (default: 9 on machines with 16 or more hardware threads, else up to 6, fewer
// on small machines; 0 = caller only).
class host_workers {
 public:
  host_workers() : stop_(false), generation_(0), n_(0), chunk_(1), next_(0), pending_(0) {
    const unsigned hw = std::thread::hardware_concurrency();
    unsigned want = hw >= 16 ? 9 : 6;  // (the device runtime's threads keep the rest)
    if (hw && hw / 2 < want + 1 && hw < 16) want = hw / 2 > 0 ? hw / 2 - 1 : 0;
    if (const char *e = std::getenv("LRR_BATCH_THREADS")) want = (unsigned)std::max(0, std::atoi(e));
    for (unsigned t = 0; t < want; ++t) threads_.emplace_back([this] { work(); });
  }
  ~host_workers() {
    { std::lock_guard<std::mutex> g(m_); stop_ = true; ++generation_; }
    cv_.notify_all();
    for (auto &t : threads_) t.join();
  }
  host_workers(const host_workers &) = delete;
  host_workers &operator=(const host_workers &) = delete;
  template <class F> void parallel_for(size_t n, F f) {
    if (threads_.empty() || n < 32) { for (size_t i = 0; i < n; ++i) f(i); return; }
    fn_ = [&f](size_t i) { f(i); };
    {
      std::lock_guard<std::mutex> g(m_);
      n_ = n;
      chunk_ = std::max<size_t>(4, n / (8 * (threads_.size() + 1)));
      next_.store(0);
      pending_ = threads_.size();
      ++generation_;
    }
    cv_.notify_all();
    drain();
    std::unique_lock<std::mutex> g(m_);
    done_.wait(g, [this] { return pending_ == 0; });
  }

 private:
  void drain() {
    for (;;) {
      const size_t b = next_.fetch_add(chunk_);
      if (b >= n_) return;
      const size_t e = std::min(n_, b + chunk_);
      for (size_t i = b; i < e; ++i) fn_(i);
    }
  }
  void work() {
    unsigned long seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> g(m_);
        cv_.wait(g, [&] { return generation_ != seen; });
        seen = generation_;
        if (stop_) return;
      }
      drain();
      std::lock_guard<std::mutex> g(m_);
      if (--pending_ == 0) done_.notify_one();
    }
  }
  std::vector<std::thread> threads_;
  std::mutex m_;
  std::condition_variable cv_, done_;
  bool stop_;
  unsigned long generation_;
  std::function<void(size_t)> fn_;
  size_t n_, chunk_;
  std::atomic<size_t> next_;
  size_t pending_;
};

template <class FeatureT = cdcl_feature_ICP<3> >
class rrl_location_batch {
 public:
  typedef typename FeatureT::sptr feature_sptr_type;
  typedef cdcl_trans<3, 12>::sptr trans_sptr_type;
  static const unsigned dof = 12;

  // uploads the two full volumes once (m_FeaturesMovingAll[0], m_FeaturesFixedAll[0]); they stay resident
  rrl_location_batch(vcl_vector<feature_sptr_type> const &moving_all, vcl_vector<feature_sptr_type> const &fixed_all)
      : batch_(nullptr), max_iterations_(70), min_features_(1000), diagnostics_(true), check_valid_(true),
        device_loop_(std::getenv("LRR_BATCH_HOST_LOOP") == nullptr),
        voxel_spacing_(1.0), sample_interval_(5.0) {
    moving_.upload(moving_all);
    fixed_.upload(fixed_all);
    check(lrr_icp_batch_create(device::ctx(), moving_.handle(), fixed_.handle(), &batch_), "batch create");
  }
  ~rrl_location_batch() { if (batch_) lrr_icp_batch_destroy(batch_); }
  rrl_location_batch(const rrl_location_batch &) = delete;
  rrl_location_batch &operator=(const rrl_location_batch &) = delete;

  void set_max_iterations(unsigned n) { max_iterations_ = n; }   // 70 (:2413)
  void set_min_features(unsigned n) { min_features_ = n; }       // 1000 (:1632)
  void set_diagnostics(bool on) { diagnostics_ = on; }
  void set_check_valid_transform(bool on) { check_valid_ = on; } // `if (!validTransform(...)) break;` (:2482)
  void set_device_loop(bool on) { device_loop_ = on; }           // false: the host half of every iteration on the host
  void set_sampling(double voxel_spacing, double interval) { voxel_spacing_ = voxel_spacing; sample_interval_ = interval; }
  // The caller hands the two 1 mm Voronoi index volumes to every location's estimator (set_voronoi_maps,
  // rrl/itkLocationRegistration.cxx:1641-1643).  Here once for the batch: every location is then matched with the maps'
  // lookup semantics (LRR_ICP_NN_VORONOI; the images are not read), unless lrr_amd::voronoi_lookup::enable(false) /
  // LRR_ICP_VORONOI=0 keeps the exact search — the same rule as icp_symmetric::set_voronoi_maps.
  template <class VoronoiPointer> void set_voronoi_maps(VoronoiPointer const &, VoronoiPointer const &) {
    if (voronoi_lookup::enabled()) check(lrr_icp_batch_set_nn_method(batch_, LRR_ICP_NN_VORONOI), "batch matcher");
    else { voronoi_lookup::note_exact(); check(lrr_icp_batch_set_nn_method(batch_, LRR_ICP_NN_GRID), "batch matcher"); }
  }
  void set_nn_method(int method) { check(lrr_icp_batch_set_nn_method(batch_, method), "batch matcher"); }
  lrr_icp_batch *handle() const { return batch_; }
  // wall-clock milliseconds of the phases of the last run() (roi, host, iterate, rebuild, angles, ...): what bench.py prices
  // the device-bound part of a batch with
  std::vector<std::pair<std::string, double> > const &last_phases() const { return last_phases_; }

  // registers every query; results[i] belongs to queries[i]
  void run(std::vector<location_query> const &queries, std::vector<location_result> &results) {
    const size_t B = queries.size();
    results.assign(B, location_result());
    if (B == 0) return;
    phase_timer tm(std::getenv("LRR_BATCH_TIMING") != nullptr);
    last_phases_.clear();
    std::vector<double> mbox(6 * B), fbox(6 * B);
    for (size_t b = 0; b < B; ++b)
      for (int d = 0; d < 3; ++d) {
        mbox[6 * b + d] = queries[b].moving_x0[d]; mbox[6 * b + 3 + d] = queries[b].moving_x1[d];
        fbox[6 * b + d] = queries[b].fixed_x0[d]; fbox[6 * b + 3 + d] = queries[b].fixed_x1[d];
      }
    std::vector<int64_t> nm(B), nf(B);
    check(lrr_icp_batch_set_rois(batch_, (int64_t)B, mbox.data(), fbox.data(), nm.data(), nf.data()), "batch ROI selection");
    tm.lap("roi");

    std::vector<state> st(B);
    std::vector<lrr_icp_batch_loc> loc(B);
    size_t running = 0;
    for (size_t b = 0; b < B; ++b) {
      state &s = st[b];
      location_result &r = results[b];
      r.n_moving_inside = nm[b];
      r.n_fixed_inside = nf[b];
      r.enough_features = !(nm[b] < (int64_t)min_features_ || nf[b] < (int64_t)min_features_);
      s.running = r.enough_features && nm[b] > 0 && nf[b] > 0;
      std::memset(&loc[b], 0, sizeof loc[b]);
      if (!s.running) continue;
      ++running;
      // SetupEstimation: set_features + set_transform (:1635-1637; rrl sym .txx:268-315); the transforms follow below
      s.N[0] = (unsigned)nm[b]; s.N[1] = (unsigned)nf[b];
      for (int d = 0; d < 2; ++d) { s.number_points[d] = std::min(1000u, s.N[d]); s.scale_estimated[d] = false; }
      s.rng[0] = queries[b].seed;
      s.rng[1] = queries[b].seed ^ 0xD1B54A32D192ED03ULL;
      s.sched.reset();
      s.error_after = 0.0;  // weighted_error() of a list without matches
      s.iteration = 0;
    }

    workers_.parallel_for(B, [&](size_t b) {  // (the inverse is a 3 x 3 SVD: microseconds each, independent)
      if (!st[b].running) return;
      st[b].F = queries[b].transform->clone();
      st[b].B = st[b].F->inverse();
    });

    lrr_icp_batch_opts opts;
    opts.normalize = 1; opts.strength_f = 1; opts.strength_b = 1; opts.swapped = 0;
    std::vector<lrr_icp_sym_out> est(B);
    std::vector<double> sums(8 * B);
    if (device_loop_ && running > 0) {
      // the caller's loop of every location on the device (lrr_icp_batch_iterate): the host half of estimate_LS and the
      // convergence test run as kernels; what comes back is the schedule's outcome and the normal equations of the
      // last estimate, from which the final transform and its covariance are rebuilt exactly as in the loop below
      std::vector<lrr_icp_batch_start> start(B);
      std::vector<lrr_icp_batch_final> fin(B);
      for (size_t b = 0; b < B; ++b) {
        std::memset(&start[b], 0, sizeof start[b]);
        if (!st[b].running) continue;
        start[b].run = 1;
        start[b].xf_fwd = xform_of(st[b].F);
        start[b].xf_bwd = xform_of(st[b].B);
        start[b].rng_fwd = st[b].rng[0];
        start[b].rng_bwd = st[b].rng[1];
      }
      tm.lap("host");
      check(lrr_icp_batch_iterate(batch_, start.data(), max_iterations_, check_valid_ ? 1 : 0, fin.data()), "batch iterate");
      tm.lap("iterate");
      workers_.parallel_for(B, [&](size_t b) {
        state &s = st[b];
        if (!s.running) return;
        const lrr_icp_batch_final &f = fin[b];
        location_result &r = results[b];
        s.running = false;
        s.iteration = (unsigned)f.iterations;
        // the lists hold the subsets of the last search; the schedule may have grown number_points_ after it
        s.number_points[0] = (unsigned)f.searched_fwd; s.number_points[1] = (unsigned)f.searched_bwd;
        s.rng[0] = f.rng_fwd; s.rng[1] = f.rng_bwd;
        s.error_after = f.weighted_error;
        s.rms = f.rms;
        if (f.iterations > 0) {
          s.scale_estimated[0] = s.scale_estimated[1] = true;
          lrr_icp_sym_out &e = est[b];
          std::memset(&e, 0, sizeof e);
          e.norm = f.norm;
          std::memcpy(e.AtA, f.AtA, sizeof e.AtA);
          std::memcpy(e.Atb, f.Atb, sizeof e.Atb);
          affine_forward_update(s.F, e);
          s.B = s.F->inverse();
        }
        r.matched_pairs = f.matched_pairs;
        r.initial_RMS = f.initial_rms;
        r.valid_transform = f.valid_transform != 0;
        r.converged = f.converged != 0;
        r.oscillated = f.oscillated != 0;
        r.iterations = (unsigned)f.iterations;
        r.weighted_error = f.weighted_error;
        r.RMS_error = f.rms;
        r.forward = s.F->clone();
      });
      running = 0;
      tm.lap("rebuild");
    }
    // the same loop with its host half on the host (set_device_loop(false) / LRR_BATCH_HOST_LOOP): all locations in
    // lockstep, one_iteration of every running location (rrl sym .txx:341-438)
    while (running > 0) {
      for (size_t b = 0; b < B; ++b) {
        state &s = st[b];
        loc[b].active = s.running ? 1 : 0;
        if (!s.running) continue;
        s.error_before = s.error_after;  // forward_->weighted_error() before the search: the value after the last estimate
        lrr_icp_perm_keys(&s.rng[0], loc[b].keys_fwd);
        lrr_icp_perm_keys(&s.rng[1], loc[b].keys_bwd);
        fill_loc(loc[b], s);
        results[b].matched_pairs += (int64_t)s.number_points[0] + s.number_points[1];
      }
      tm.lap("host");
      check(lrr_icp_batch_match(batch_, loc.data()), "batch find_closest_euclidean");
      tm.lap("match");
      check(lrr_icp_batch_estimate(batch_, loc.data(), &opts, est.data()), "batch estimate");
      tm.lap("estimate");
      // the host half of estimate_LS
      workers_.parallel_for(B, [&](size_t b) {
        state &s = st[b];
        if (!s.running) return;
        s.scale_estimated[0] = s.scale_estimated[1] = true;
        affine_forward_update(s.F, est[b]);
        s.B = s.F->inverse();
        fill_loc(loc[b], s);
      });
      tm.lap("solve");
      check(lrr_icp_batch_errors(batch_, loc.data(), sums.data()), "batch error metrics");
      tm.lap("errors");
      for (size_t b = 0; b < B; ++b) {
        state &s = st[b];
        if (!s.running) continue;
        const double *e = &sums[6 * b];
        s.error_after = e[2] != 0 ? e[1] / e[2] : e[1];  // weighted_error (rrl .txx:247-276)
        s.rms = e[0];                                     // the rrl class returns the raw sum (rrl .txx:216-238)
        const bool converged = s.sched.step(s.error_before, s.error_after, s.iteration, true, [&s](unsigned n) {
          bool a = grow(s, 0, n), c = grow(s, 1, n);
          return a && c;
        });
        if (s.iteration == 0) results[b].initial_RMS = s.rms;
        ++s.iteration;
        bool stop = converged || s.iteration >= max_iterations_;
        // `++iteration; if (!validTransform(...)) break;` (:2482): tested after every iteration, whether or not the loop
        // condition ends it too
        if (check_valid_ && !valid_transform(s.F)) { results[b].valid_transform = false; stop = true; }
        if (stop) {
          s.running = false;
          --running;
          location_result &r = results[b];
          r.converged = converged;
          r.oscillated = s.sched.oscillated();
          r.iterations = s.iteration;
          r.weighted_error = s.error_after;
          r.RMS_error = s.rms;
          r.forward = s.F->clone();
        }
      }
    }
    tm.lap("host");
    if (diagnostics_) decision_measurements(queries, st, loc, results, est, sums, tm);
    tm.report(B);
    last_phases_ = tm.acc;
    for (size_t b = 0; b < B; ++b)
      if (results[b].enough_features && st[b].B) results[b].backward = st[b].B;
  }

 private:
  // wall-clock per phase of run(), printed when LRR_BATCH_TIMING is set (tuning aid)
  struct phase_timer {
    bool on;
    std::chrono::steady_clock::time_point t;
    std::vector<std::pair<std::string, double> > acc;
    explicit phase_timer(bool enable) : on(enable), t(std::chrono::steady_clock::now()) {}
    void lap(const char *name) {
      auto now = std::chrono::steady_clock::now();
      double ms = std::chrono::duration<double, std::milli>(now - t).count();
      t = now;
      for (auto &a : acc) if (a.first == name) { a.second += ms; return; }
      acc.push_back(std::make_pair(std::string(name), ms));
    }
    void report(size_t B) {
      if (!on) return;
      std::fprintf(stderr, "[lrr batch] %zu locations:", B);
      for (auto &a : acc) std::fprintf(stderr, " %s %.2f ms", a.first.c_str(), a.second);
      std::fprintf(stderr, "\n");
    }
  };
  struct state {
    trans_sptr_type F, B;
    unsigned N[2], number_points[2];
    bool scale_estimated[2];
    uint64_t rng[2];
    symmetric_schedule sched;
    double error_before, error_after, rms;
    unsigned iteration;
    bool running;
    state() : error_before(0.0), error_after(0.0), rms(0.0), iteration(0), running(false) {
      N[0] = N[1] = number_points[0] = number_points[1] = 0;
      scale_estimated[0] = scale_estimated[1] = false;
      rng[0] = rng[1] = 0;
    }
  };
  // set_number_points (cdcl .txx:94-108)
  static bool grow(state &s, int d, unsigned n) {
    if (n >= s.N[d]) { s.number_points[d] = s.N[d]; return true; }
    s.number_points[d] = n;
    return false;
  }
  static lrr_icp_xform xform_of(trans_sptr_type const &T) {
    lrr_icp_xform x;
    vnl_matrix_fixed<double, 3, 3> &A = T->get_A();
    vnl_vector_fixed<double, 3> &t = T->get_translation();
    for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) x.A[3 * r + c] = A(r, c); x.t[r] = t[r]; x.c[r] = T->center_moving_[r]; }
    return x;
  }
  static void fill_loc(lrr_icp_batch_loc &L, state const &s) {
    L.xf_fwd = xform_of(s.F);
    L.xf_bwd = xform_of(s.B);
    L.n_fwd = s.number_points[0];
    L.n_bwd = s.number_points[1];
    // first estimate after set_transform: MUSE; afterwards the RMS under the previous weights (rrl .txx:620-641)
    L.scale_mode_f = s.scale_estimated[0] ? LRR_ICP_SCALE_RMS : LRR_ICP_SCALE_MUSE;
    L.scale_mode_b = s.scale_estimated[1] ? LRR_ICP_SCALE_RMS : LRR_ICP_SCALE_MUSE;
    L.injected_f = L.injected_b = 0.0;
    // the affine normalize() zeroes the centre before the accumulation (cdcl_trans_affine.txx:252)
    L.jac_centre[0] = L.jac_centre[1] = L.jac_centre[2] = 0.0;
  }

  // ComputeDecisionMeasurements (:1816-1871) of every location that ran: forward angles and error, then
  // estimate_LS_backward(false) (sets the backward covariance, re-weights both lists), then the backward ones
  void decision_measurements(std::vector<location_query> const &queries, std::vector<state> &st, std::vector<lrr_icp_batch_loc> &loc,
                             std::vector<location_result> &results, std::vector<lrr_icp_sym_out> &est, std::vector<double> &sums, phase_timer &tm) {
    const size_t B = st.size();
    bool any = false;
    for (size_t b = 0; b < B; ++b) {
      loc[b].active = (results[b].enough_features && st[b].iteration > 0) ? 1 : 0;
      if (loc[b].active) { any = true; fill_loc(loc[b], st[b]); }
    }
    if (!any) return;
    const double deg = 180.0 / 3.14159265358979323846;
    check(lrr_icp_batch_angles(batch_, loc.data(), sums.data()), "batch angles");
    tm.lap("angles");
    for (size_t b = 0; b < B; ++b)
      if (loc[b].active) {
        const double *a = &sums[8 * b];
        results[b].sheet_angles = a[0] / a[1];
        results[b].tube_angles = a[2] / a[3];
      }
    // estimate_LS_backward(false): the lists swap roles (rrl sym .txx:921-934); no new parameters
    lrr_icp_batch_opts opts;
    opts.normalize = 1; opts.strength_f = 1; opts.strength_b = 1; opts.swapped = 1;
    for (size_t b = 0; b < B; ++b)
      if (loc[b].active) { loc[b].scale_mode_f = loc[b].scale_mode_b = LRR_ICP_SCALE_RMS; }
    check(lrr_icp_batch_estimate(batch_, loc.data(), &opts, est.data()), "batch backward estimate");
    tm.lap("backward estimate");
    workers_.parallel_for(B, [&](size_t b) {
      if (!loc[b].active) return;
      symmetric_apply_solution<dof>(st[b].B, st[b].F, est[b], true, false);
      fill_loc(loc[b], st[b]);
    });
    tm.lap("apply");
    check(lrr_icp_batch_angles(batch_, loc.data(), sums.data()), "batch angles");
    std::vector<double> esums(6 * B);
    check(lrr_icp_batch_errors(batch_, loc.data(), esums.data()), "batch error metrics");
    tm.lap("angles + errors");
    workers_.parallel_for(B, [&](size_t b) {
      if (!loc[b].active) return;
      location_result &r = results[b];
      const double *a = &sums[8 * b + 4], *e = &esums[6 * b + 3];
      r.sheet_angles_backward = a[0] / a[1];
      r.tube_angles_backward = a[2] / a[3];
      r.weighted_error_backward = e[2] != 0 ? e[1] / e[2] : e[1];
      // transfer-error covariance sampled over the two ROIs (:1832-1841, 2793-2858); the forward one uses the
      // transform as the loop left it
      vnl_vector_fixed<double, 3> sp(voxel_spacing_);
      const double maxTrace = sampled_transfer_max_trace(r.forward, queries[b].moving_x0, queries[b].moving_x1 - sp, sample_interval_);
      const double maxTraceB = sampled_transfer_max_trace(st[b].B, queries[b].fixed_x0, queries[b].fixed_x1 - sp, sample_interval_);
      const double sa = r.sheet_angles * deg, ta = r.tube_angles * deg, sab = r.sheet_angles_backward * deg, tab = r.tube_angles_backward * deg;
      r.measurements[0] = std::max(r.weighted_error, r.weighted_error_backward);
      r.measurements[1] = std::max(maxTrace, maxTraceB);
      r.measurements[2] = std::max(sa, sab);
      r.measurements[3] = std::max(ta, tab);
      r.measurements[4] = std::min(r.weighted_error, r.weighted_error_backward);
      r.measurements[5] = std::min(maxTrace, maxTraceB);
      r.measurements[6] = std::min(sa, sab);
      r.measurements[7] = std::min(ta, tab);
    });
    tm.lap("measurements");
  }

  device_cloud<FeatureT> moving_, fixed_;
  lrr_icp_batch *batch_;
  host_workers workers_;
  unsigned max_iterations_, min_features_;
  bool diagnostics_, check_valid_, device_loop_;
  double voxel_spacing_, sample_interval_;
  std::vector<std::pair<std::string, double> > last_phases_;
};

}  // namespace lrr_amd
#endif

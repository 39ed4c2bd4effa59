// C ABI of libekfslam_hip.so (include/ekfslam_hip.h): host-side orchestration of the HIP kernels.
// No CPU fallback exists: every entry point needs a gfx950 device.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "ekf_device.h"
#include "ekf_host_plan.h"

namespace ekf {
void launch_solve(hipStream_t, const double*, const double*, const double*, const double*, double*, const double*,
                  double*, const int*, const StepIn*, SolveOut*, unsigned*, double*, const int*, unsigned*, const DeviceConfig&,
                  int, long, int, int);
bool step_is_split(int batch, int n_hi, int cus);
void launch_step_split(hipStream_t, int, double*, double*, double*, const double*, double*, const double*, double*,
                       const int*, const StepIn*, SolveOut*, unsigned*, double*, const int*, unsigned*, unsigned*, unsigned,
                       int, const DeviceConfig&, int, long, int, int, int);
void launch_step_split_tp(hipStream_t, int, double*, double*, double*, const double*, double*, const double*, double*,
                          const int*, const StepIn*, SolveOut*, unsigned*, double*, const int*, unsigned*, SolveOut*,
                          unsigned*, unsigned, int, const DeviceConfig&, int, long, int, int, int);
void launch_panels(hipStream_t, int, double*, double*, double*, const double*, double*, const int*,
                   const SolveOut*, const double*, int, long, int, int);
void launch_flush(hipStream_t, bool, double*, const double*, const double*, const double*, const int*,
                  const SolveOut*, int, long, int, int, int, int);
void launch_flush_rs(hipStream_t, bool, double*, const double*, const double*, const double*, const int*,
                     const SolveOut*, int, long, int, int, int, int, unsigned*, int, const int*, const CadOut*);
int flush_rs_queue_words();
void launch_predict_rc(hipStream_t, double*, const double*, double*, const int*, const SolveOut*, int, long,
                       int, int);
void launch_add_landmarks(hipStream_t, double*, double*, int, int, int, double, const double*);
void launch_mirror(hipStream_t, double*, const int*, int, long, int, int);
void launch_pack_small(hipStream_t, const double*, const double*, const unsigned*, int, int, double*);
void launch_pack_dense(hipStream_t, const double*, int, int, double*);
int small_state_limit(int batch);
int launch_small_stream(hipStream_t, double*, const double*, double*, const int*, const StepIn*, int, int, unsigned*,
                        const DeviceConfig&, int, long, int, double*, int, unsigned long long*, unsigned long long, bool);
void launch_associate(hipStream_t, const DetIn*, int*, int*, int*, double*, double*, double*, double*, StepIn*,
                      AssocOut*, unsigned*, const AssocConfig&, int, long, int, int, int);
void launch_fill_diag(hipStream_t, double*, int, int, const double*);
int dense_propagate(hipStream_t, double* P, double* tmp, const double* F, const double* Q, int n, int ld);
void launch_solve_cad(hipStream_t, const double*, const double*, double*, double*, const int*, const StepIn*, const CadPlan*, int,
                      CadOut*, unsigned*, const DeviceConfig&, int, long, const double*, int, double*, int, int, bool, const double*,
                      unsigned*, unsigned, const CadPre*);
void launch_chain_cad(hipStream_t, const double*, const double*, const double*, const double*, const CadOut*, const StepIn*,
                      const CadPlan*, int, const DeviceConfig&, int, long, double*, double*, double*, double*, unsigned*, unsigned,
                      unsigned*, int, unsigned, const CadPre*, CadPre*, const CadPlan*, bool);
void launch_mark(hipStream_t, unsigned*, unsigned);
void launch_gate(hipStream_t, unsigned*, unsigned, unsigned*, int);
int panels_cad_workgroups(int, int);
int chain_gather_workgroups(int, int);
int chain_sync_words();
void launch_snap_pose(hipStream_t, const double*, const int*, int, long, int, int, double*);
void launch_gather_cad(hipStream_t, const double*, const double*, const double*, const double*, const StepIn*, const CadPlan*, int,
                       int, const DeviceConfig&, int, long, double*);
long cadence_gbuf_doubles();
void launch_panels_cad(hipStream_t, double*, double*, double*, const double*, double*, const int*, const CadOut*,
                       SolveOut*, unsigned*, int, long, int, int, int, const double*, double*, unsigned*, unsigned, unsigned, unsigned*,
                       bool, unsigned, int, bool);
bool panels_cad_latency_regime(int, int);
}  // namespace ekf

using namespace ekf;

static thread_local std::string g_create_error;

// Stream pairs outlive their handles.  A handle drives two HIP streams (its own and the one the look-ahead sends the covariance
// pass to, beside the next cadence's solve).  HIP maps streams onto a few hardware queues as they are created; measured on this
// pool: the second handle of a process -- created after the first one's streams had been destroyed -- ran its pass and the solve
// beside it 10 % slower (N = 8000 x 1: 389 us against 355 us; 355 us again with lookahead = 0: the two streams no longer ran
// side by side), whatever memory it was given (tools/leg_order_probe.py, profiles/r04_dense_operands.txt part 2).  So the pair
// a destroyed handle leaves is parked here, per device, and the next handle on that device takes it over: the mapping the
// first pair got in a fresh process is the one every later handle runs on.  (Streams are idle when parked: free_all
// synchronises them.)
#include <mutex>
struct StreamPair { int device; hipStream_t stream, aux; };
static std::mutex g_pairs_mu;
static std::vector<StreamPair> g_pairs;
static bool take_stream_pair(int device, hipStream_t* stream, hipStream_t* aux) {
  std::lock_guard<std::mutex> lk(g_pairs_mu);
  for (size_t i = 0; i < g_pairs.size(); ++i)
    if (g_pairs[i].device == device) {
      *stream = g_pairs[i].stream;
      *aux = g_pairs[i].aux;
      g_pairs.erase(g_pairs.begin() + (long)i);
      return true;
    }
  return false;
}
static void park_stream_pair(int device, hipStream_t stream, hipStream_t aux) {
  std::lock_guard<std::mutex> lk(g_pairs_mu);
  g_pairs.push_back({device, stream, aux});
}

constexpr int RING = 16;
constexpr int RING_GROUP = 4;             // slots per completion event: an event record between two launches costs the
                                          // stream a barrier packet, ~2 us per online step of a small filter
constexpr int PACK_SMALL_N = 131;        // states up to 64 landmarks are downloaded by k_pack_small (137 KB of pinned memory)

struct ekf_handle : ekf::HostPlan {
  DeviceConfig dcfg{};
  hipStream_t stream = nullptr;
  double *dP = nullptr, *dV = nullptr, *dW = nullptr, *dscratch = nullptr;
  double* ddacc2[2] = {nullptr, nullptr};  // pending pose-block noise, double-buffered like the mean
  int dcur = 0;
  double* dmu2[2] = {nullptr, nullptr};   // the mean is double-buffered: a step reads [cur], writes [cur^1]
  int cur = 0;
  int* dn = nullptr;
  unsigned* dflags = nullptr;
  SolveOut* dso = nullptr;
  double* dfac = nullptr;         // pending factors restricted to the gathered indices (k_solve -> k_panels)
  StepIn *d_ring = nullptr, *h_ring = nullptr;
  hipEvent_t ring_ev[RING / RING_GROUP]{};   // one event per group of slots (see ring_take)
  bool ring_used[RING / RING_GROUP]{};
  bool ring_open[RING / RING_GROUP]{};
  int ring_pos = 0;
  StepIn* d_stream = nullptr;
  size_t stream_cap = 0;
  int* dfloor = nullptr;          // per trajectory floor of the active bound, applied by k_solve (see push_floor)
  double *dF = nullptr, *dQ = nullptr, *dTmp = nullptr;   // dense path, allocated on first use
  double* dPlin = nullptr;        // dense path with P in column panels: its row-major staging copy
  // device-side association (allocated on first use)
  int *dtagmap = nullptr, *dneff = nullptr;
  DetIn *d_det = nullptr, *h_det = nullptr;
  StepIn* d_assoc_step = nullptr;
  AssocOut* d_assoc_out = nullptr;
  AssocConfig acfg{};
  hipEvent_t t0 = nullptr, t1 = nullptr;
  bool profile = false;
  int profile_stride = 1;         // every how many launches of the pass carry an event pair while profiling ("profile_stride")
  long prof_seen = 0;
  std::vector<hipEvent_t> prof_pool;
  size_t prof_used = 0;
  // "profile_kernels" = 1: the cadence's other launches carry event pairs too (a diagnostic run: every record costs its stream
  // ~6 us); class of pair i: 0 the covariance pass, 1 the solve launch, 2 the chain / look-ahead gather launch, 3 the panel launch
  int opt_profile_kernels = 0;
  std::vector<int> prof_cls;
  unsigned* dqueue = nullptr;     // work-queue heads of k_flush_rs (zeroed before every launch)
  // k_flush_rs, equal static shares (a few long trajectories): the piece table.  Two copies on the device and in pinned
  // host memory, used alternately: a rebuilt table is uploaded stream-ordered, without a host synchronisation, while the
  // pass that read the previous one may still be running.
  int* dshares2[2] = {nullptr, nullptr};
  int* hshares2[2] = {nullptr, nullptr};
  hipEvent_t shares_ev[2] = {nullptr, nullptr};   // the upload out of hshares2[i] has been executed
  bool shares_ev_used[2] = {false, false};
  int shares_cur = 0;
  int shares_key[5] = {0, 0, 0, 0, 0};   // (batch, slabs, last strip, workgroups, order) the current table was built for
  int shares_ok = 0;              // pieces of its longest share (0: no table for this key -- the queue modes are used)
  unsigned* dready = nullptr;     // per trajectory: sequence number of the last solve that completed (k_step_split)
  SolveOut* dmbox = nullptr;      // per trajectory: that solve's header and records, written through (mailbox_publish)
  unsigned step_seq = 0;          // sequence number of the last single-launch step
  // per trajectory: head + per-landmark records of a cadence (allocated on first use).  Two copies, used alternately by
  // consecutive cadences (`cpar`): in a chained run the next cadence's solve writes its records while this cadence's panel
  // launch still reads these
  CadOut* dcad2[2] = {nullptr, nullptr};
  int cpar = 0;
  // Chained solves (round 6; "chain"): the solves of a run follow one another on the handle's stream -- the next cadence's
  // block comes from this cadence's records (k_chain_cad) -- while panel launch and covariance pass of every cadence run
  // on the second stream.  dprow3: rows 0..2 of every P after a cadence, left by its panel launch (batch x 3 x ld, two copies
  // like dcad2); dgmu: the mean at the next cadence's positions (batch x 128).
  double* dprow3[2] = {nullptr, nullptr};
  double* dgmu = nullptr;
  double *dxg = nullptr, *dbg = nullptr;   // the chain launch's gathered rows (batch x 84 x 88 each): written by its gather workgroups
  unsigned* dsync = nullptr;      // device-scope counters of the chained run's hand-overs (ekf_cadence.hip: SYNC_*)
  unsigned gather_count = 0;      // gather workgroups launched so far (what the next chain workgroups wait for)
  unsigned sigma = 0;             // chained transitions so far (the value the run's counters SYNC_SOLVE / SYNC_PASS carry)
  // a cadence's inputs formed one cadence ahead (CadPre): two copies, by the cadence's serial number (pre_serial: whose inputs a
  // copy holds; serials count every cadence of the handle)
  CadPre* dpre[2] = {nullptr, nullptr};
  long pre_serial[2] = {-1, -1};
  long cad_serial = 0;
  int opt_pre_positions = 1;
  // covariance (MB, whole bank) from which the next solve runs beside the pass in a chained run.  0: always -- with the counters'
  // hand-overs chaining wins at every size tried (N = 12 .. 1000, banks of 1 .. 32: +25 .. +43 %); the round-3 look-ahead, whose
  // hand-overs are events (~25 us per cadence), keeps its 48 MB
  int opt_beside_min_mb = 0;
  int opt_lookahead_min_mb = 48;
  hipEvent_t ev_pass = nullptr;   // recorded on the second stream when a chain of cadences ends (join_aux): the only event of the chained order
  bool chain_run = false;         // the run in flight records the transforms (every solve is k_solve_cad<true>)
  bool aux_pass = false;          // a covariance pass is in flight on the second stream (ev_pass recorded behind it)
  int opt_chain = 1;
  // 0 (default) = a one-lane gate launch in front of every chained panel launch: ~5 us of the second stream, which has them to
  // spare (no measurable cost: 62.2 against 61.5 k at N = 2000 x 1, profiles/r06_chained_solves.txt), and no workgroup of a large
  // launch ever spins; 1 = small panel launches (each workgroup a CU to itself) are their own gate (panel_head_wait)
  int opt_panel_own_gate = 0;
  // 1 = where a fused cadence's covariance pass follows its panel launch at once, in the row-slab form, the panel launch writes V
  // only and the pass forms its W fragments from V and the records' S^-1 (half of the panel launch's stores); bit-identical
  int opt_w_from_v = 1;
  long w_from_v_passes = 0;       // statistics
  int opt_panel_shape = 0;        // diagnostics: 0 = the panel launch's shape by its size; 1 k_panels_cad_ks, 2 k_panels_cad<1>, 3 k_panels_cad<4> whatever the size
  int opt_panel_tform = 1;        // 1 = a chained cadence's panel launch in the latency regime takes the triangular-solve form (k_panels_cad_tf)
  int opt_run_end_flush = 0;      // 1 = ekf_stream_run applies what its last cadence left pending, so that the next call starts fused
  long chained = 0;               // statistics: cadences whose block came from k_chain_cad
  // The mirrored column entries of a cadence's panel launch, gathered by extra workgroups of its solve launch and laid down
  // as rows (batch x 83 x ld doubles, allocated on first use; not for banks where that would exceed 1 GiB).  `colbuf_live`:
  // the solve in flight has filled it (a solve launched beside a covariance pass -- look-ahead -- cannot: P_base is in motion)
  double* dcolbuf = nullptr;
  bool colbuf_live = false;
  int opt_col_gather = 1;         // 1 = gather them beside the solve, 0 = the panel launch gathers everything itself
  long cadences = 0, cadence_traj_steps = 0;   // statistics: fused cadences launched, trajectory-steps they completed
  // The packed cadences of the ekf_stream_run in flight (ekf_host_plan.h: plan_cadences): one CadPlan per (cadence,
  // trajectory), planned on the host for the whole run and uploaded once, stream-ordered, out of pinned memory.  Two copies,
  // used alternately: the host may plan the next run while the upload of this one has not executed yet.
  RunPlan run_plan;
  CadPlan* dplan2[2] = {nullptr, nullptr};
  CadPlan* hplan2[2] = {nullptr, nullptr};
  size_t plan_cap2[2] = {0, 0};
  hipEvent_t plan_ev[2] = {nullptr, nullptr};     // the upload out of hplan2[i] has been executed
  bool plan_ev_used[2] = {false, false};
  int plan_cur = 0;
  // look-ahead (small launches): the solve of the next cadence runs on the handle's stream beside the covariance pass of
  // this one, which goes to a second stream between two events; see ekf_stream_run
  hipStream_t aux = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  double* dgbuf = nullptr;        // per trajectory: the next cadence's block, gathered while this one's ranks are pending
  long lookaheads = 0;
  long assoc_fallbacks = 0;       // statistics: windows a binding took through the host association (ekf_debug_note_assoc_fallback)
  long small_launches = 0;        // statistics: launches of the small-state path (k_small_stream)
  long fused_fetches = 0;         // statistics: ekf_step_fetch calls answered by the step's own launch
  long dense_packs = 0;           // statistics: large downloads written by k_pack_dense (pinned destination)
  int opt_pack_dense = 1;
  int fetch_b = -1;               // ekf_step_fetch: the trajectory whose state the next small-state launch leaves in h_pack
  bool fetched = false;
  unsigned long long fetch_seq = 0;   // ... and the sequence number that launch releases behind it (polled by the host)
  int opt_fetch_spin = 1;
  int opt_fetch_verify = 1;       // 1 (default) = ekf_step_fetch checks the payload's XOR checksum before it trusts a polled hand-over (~1 us of the 10 - 15 us the polling saves); 0 = the trailer's second sequence number only
  long fetch_retries = 0;         // statistics: hand-overs whose integrity trailer did not match (answered after a stream sync)
  int opt_zero_copy_inputs = 1;   // small-state online steps read their records from the pinned ring (no staged copy)
  int last_kernel = -1, last_nkt = 0, last_streaming = 0;   // what the last covariance pass launched
  int last_wv = 0;                // ... and whether it formed its W fragments from V ("w_from_v")
  int last_shares = 0;            // ... and whether it ran on equal static shares (k_flush_rs, a few long trajectories)
  std::vector<unsigned> flags_host;
  unsigned* h_flags = nullptr;    // pinned: the sticky flags are read back with a stream-ordered copy
  double* h_pack = nullptr;       // pinned: where k_pack_small leaves a small state (n x n covariance, mean, flags)
  // Set when an enqueueing call failed half way (e.g. a launch of the look-ahead failed after the next cadence's solve had
  // already run): the device state of every trajectory is undefined until it is uploaded again; see check_internal
  std::vector<unsigned char> host_bad;
  std::string err;
};

static int fail(ekf_handle* h, int code, const std::string& msg) {
  if (h) h->err = msg; else g_create_error = msg;
  return code;
}

static int flush_pending(ekf_handle* h);
static int flush_pending(ekf_handle* h, hipStream_t st, const CadOut* wv = nullptr);
static int materialize(ekf_handle* h, int b);

#define HIP_TRY(h, expr)                                                                   \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess)                                                                  \
      return fail(h, EKF_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));      \
  } while (0)

// Copies between a host matrix (row-major, `host_pitch` doubles per row) and the block [r0, r0 + rows) x [c0, c0 + cols) of
// trajectory b's covariance in its device layout (ekf_device.h: row-major up to ld = 4096, column panels of 4096 doubles
// beyond): one 2-D copy per column panel the block touches.  `other` = nullptr: the host side; else a device matrix in
// plain row-major (the dense product's staging), copied device to device.
static hipError_t copy_cov(ekf_handle* h, int b, double* host, int host_pitch, int r0, int c0, int rows, int cols, bool to_device,
                           bool device_to_device = false) {
  for (int p = c0 / PPW; p <= (c0 + cols - 1) / PPW; ++p) {
    const int cs = std::max(c0, p * PPW), ce = std::min(c0 + cols, (p + 1) * PPW);
    double* dev = h->dP + (size_t)b * h->pstride + p_index(h->ld, r0, cs);
    double* hst = host + (cs - c0);
    const size_t dpitch = sizeof(double) * (size_t)p_lds(h->ld), hpitch = sizeof(double) * (size_t)host_pitch;
    const size_t width = sizeof(double) * (size_t)(ce - cs);
    const hipMemcpyKind kind = device_to_device ? hipMemcpyDeviceToDevice : (to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost);
    const hipError_t e = to_device ? hipMemcpy2DAsync(dev, dpitch, hst, hpitch, width, (size_t)rows, kind, h->stream)
                                   : hipMemcpy2DAsync(hst, hpitch, dev, dpitch, width, (size_t)rows, kind, h->stream);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

extern "C" int ekf_config_default(ekf_config* cfg) {
  if (!cfg) return EKF_ERR_ARG;
  cfg->motion_sigma = 0.1;
  cfg->meas_sigma = 0.7;
  cfg->arc_threshold = 1e-2;
  cfg->landmark_init_var = 10000.0;
  cfg->enable_measurement_model = 1;
  cfg->enable_circular_interpolation = 1;
  cfg->disable_motion_model = 0;
  cfg->reserved = 0;
  return EKF_OK;
}

extern "C" int ekf_device_count(int* count) {
  if (!count) return EKF_ERR_ARG;
  int c = 0;
  if (hipGetDeviceCount(&c) != hipSuccess) c = 0;
  *count = c;
  return EKF_OK;
}

// Pinned host memory for the arrays a binding hands to its caller (see include/ekfslam_hip.h).
extern "C" void* ekf_host_alloc(size_t bytes) {
  void* p = nullptr;
  if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocPortable) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  return p;
}
extern "C" void ekf_host_free(void* p) {
  if (p) (void)hipHostFree(p);
}

extern "C" const char* ekf_last_error(ekf_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

static void free_all(ekf_handle* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  void* ptrs[] = {h->dP, h->dmu2[0], h->dmu2[1], h->dV, h->dW, h->ddacc2[0], h->ddacc2[1], h->dscratch, h->dn, h->dflags, h->dso, h->dfac,
                  h->d_ring, h->d_stream, h->dF, h->dQ, h->dTmp, h->dPlin, h->dtagmap, h->dneff, h->d_det, h->d_assoc_step, h->dfloor, h->dqueue, h->dready, h->dmbox,
                  h->d_assoc_out, h->dcad2[0], h->dcad2[1], h->dprow3[0], h->dprow3[1], h->dgmu, h->dxg, h->dbg, h->dsync, h->dpre[0], h->dpre[1], h->dshares2[0], h->dshares2[1], h->dgbuf, h->dplan2[0], h->dplan2[1], h->dcolbuf};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  if (h->h_ring) (void)hipHostFree(h->h_ring);
  if (h->h_det) (void)hipHostFree(h->h_det);
  for (int i = 0; i < 2; ++i) {
    if (h->hshares2[i]) (void)hipHostFree(h->hshares2[i]);
    if (h->shares_ev[i]) (void)hipEventDestroy(h->shares_ev[i]);
    if (h->hplan2[i]) (void)hipHostFree(h->hplan2[i]);
    if (h->plan_ev[i]) (void)hipEventDestroy(h->plan_ev[i]);
  }
  if (h->h_flags) (void)hipHostFree(h->h_flags);
  if (h->h_pack) (void)hipHostFree(h->h_pack);
  for (auto& e : h->ring_ev) if (e) (void)hipEventDestroy(e);
  for (auto& e : h->prof_pool) (void)hipEventDestroy(e);
  if (h->t0) (void)hipEventDestroy(h->t0);
  if (h->t1) (void)hipEventDestroy(h->t1);
  if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
  if (h->ev_join) (void)hipEventDestroy(h->ev_join);
  if (h->ev_pass) (void)hipEventDestroy(h->ev_pass);
  if (h->aux) (void)hipStreamSynchronize(h->aux);
  if (h->stream && h->aux) {
    park_stream_pair(h->device, h->stream, h->aux);    // (see g_pairs: the next handle on this device takes the pair over)
  } else {
    if (h->aux) (void)hipStreamDestroy(h->aux);
    if (h->stream) (void)hipStreamDestroy(h->stream);
  }
  delete h;
}

extern "C" int ekf_create(int device, int n_max, int batch, const ekf_config* cfg, ekf_handle** out) {
  if (!out) return fail(nullptr, EKF_ERR_ARG, "ekf_create: out is NULL");
  *out = nullptr;
  if (n_max < 3 || (n_max & 1) == 0) return fail(nullptr, EKF_ERR_ARG, "ekf_create: n_max must be 3 + 2*N");
  if (batch < 1) return fail(nullptr, EKF_ERR_ARG, "ekf_create: batch must be >= 1");
  // The kernels address one trajectory's covariance with unsigned 32-bit byte offsets (buffer instructions: k_flush_rs,
  // k_solve's staging, k_gemm_f64's resources): what is allocated for it must stay below 4 GiB.  Checked before the device is
  // looked for, so that the limit can be tested without one.
  {
    const int rows = (n_max + 63) / 64 * 64;             // (= ld beyond 4096, where the covariance is kept in column panels)
    if (n_max > EKF_N_MAX_LIMIT || (unsigned long long)p_alloc(rows, rows) * 8ull >= (1ull << 32))
      return fail(nullptr, EKF_ERR_ARG,
                  "ekf_create: n_max = " + std::to_string(n_max) + " exceeds EKF_N_MAX_LIMIT = " +
                      std::to_string(EKF_N_MAX_LIMIT) + " (one covariance must stay below 4 GiB: 32-bit byte offsets)");
  }
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0)
    return fail(nullptr, EKF_ERR_HIP, std::string("ekf_create: no HIP device (") + hipGetErrorString(e) +
                                          "); this library has no CPU fallback");
  if (device < 0 || device >= count) return fail(nullptr, EKF_ERR_ARG, "ekf_create: device index out of range");
  hipDeviceProp_t prop;
  if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess)
    return fail(nullptr, EKF_ERR_HIP, std::string("hipGetDeviceProperties: ") + hipGetErrorString(e));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(nullptr, EKF_ERR_HIP, std::string("ekf_create: device is ") + prop.gcnArchName +
                                          ", kernels are built for gfx950 only");
  ekf_handle* h = new ekf_handle();
  h->device = device;
  h->cu_count = prop.multiProcessorCount;
  h->n_max = n_max;
  // Row stride of P: a multiple of 64 doubles (every 64-column strip of the pass lies inside the row, 512-B aligned)
  // and, up to 4096, a power of two: with rows exactly 32 KB (or 16, 8 ... KB) apart the sixteen 512-byte row
  // segments of a tile fall into the same DRAM pages instead of sixteen different ones -- measured on the pass at
  // N=2000 x 32: 775 us with ld = 4032, 729 us with ld = 4096; N=1500: 767 -> 696 us; larger strides (ld = 8192) are
  // slower than the plain round-up, so beyond 4096 the stride is n_max rounded up to 64 (profiles/r02_ld_sweep.txt).
  h->rows = (n_max + 63) / 64 * 64;                    // rows allocated per trajectory
  h->ld = h->rows;
  if (n_max <= 4096) {
    int p2 = 64;
    while (p2 < n_max) p2 *= 2;
    h->ld = p2;
  }
  h->batch = batch;
  h->pstride = p_alloc(h->rows, h->ld);                // (column panels of 4096 doubles beyond ld = 4096: ekf_device.h)
  if (cfg) h->cfg = *cfg; else ekf_config_default(&h->cfg);
  if (const char* e = std::getenv("EKFSLAM_HIP_SMALL_STATE")) h->opt_small_state = std::atoi(e) != 0;   // (tests pin the general kernels at small sizes)
  if (const char* e = std::getenv("EKFSLAM_HIP_FETCH_SPIN")) h->opt_fetch_spin = std::atoi(e) != 0;
  const double s = h->cfg.motion_sigma, q = h->cfg.meas_sigma;
  h->dcfg.rd[0] = s * s;                        // src/replay_no_ros.py:421
  h->dcfg.rd[1] = s * s;
  h->dcfg.rd[2] = (s / 2) * (s / 2);
  h->dcfg.qd[0] = q * q;                        // :438
  h->dcfg.qd[1] = q * q;
  h->dcfg.arc_threshold = h->cfg.arc_threshold;
  h->dcfg.enable_measurement_model = h->cfg.enable_measurement_model;
  h->dcfg.enable_circular_interpolation = h->cfg.enable_circular_interpolation;
  h->dcfg.disable_motion_model = h->cfg.disable_motion_model;
  h->acfg.gate2 = 1.5 * 1.5;                      // src/replay_no_ros.py:289
  h->acfg.init_var = h->cfg.landmark_init_var;
  h->acfg.n_ignore = 0;
  h->acfg.active_bound = 1;
  h->n.assign(batch, 3);
  h->neff.assign(batch, 3);
  h->neff_enq.assign(batch, 3);
  h->floor_host.assign(batch, 3);
  h->host_bad.assign(batch, 0);

#define CREATE_TRY(expr)                                                                  \
  do {                                                                                    \
    hipError_t e2_ = (expr);                                                              \
    if (e2_ != hipSuccess) {                                                              \
      g_create_error = std::string(#expr) + ": " + hipGetErrorString(e2_);                \
      free_all(h);                                                                        \
      return EKF_ERR_HIP;                                                                 \
    }                                                                                     \
  } while (0)
  CREATE_TRY(hipSetDevice(device));
  if (!take_stream_pair(device, &h->stream, &h->aux)) {
    CREATE_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    // The second stream gets the LOWEST priority: the runtime multiplexes streams onto a few hardware queues per priority level,
    // so a priority of its own keeps it off the hardware queue of the handle's own stream however many streams the process has
    // created (two streams on one queue run one behind the other: the chained order stays correct -- every wait is for an
    // earlier-enqueued launch -- but nothing overlaps); and what runs there (panel launches, covariance passes) is the work
    // that may wait.  No measurable effect on a fresh process (profiles/r06_chained_solves.txt).
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) least = 0;
    if (const char* e = std::getenv("EKFSLAM_HIP_AUX_PRIORITY")) least = std::atoi(e);
    if (hipStreamCreateWithPriority(&h->aux, hipStreamNonBlocking, least) != hipSuccess) {
      (void)hipGetLastError();
      CREATE_TRY(hipStreamCreateWithFlags(&h->aux, hipStreamNonBlocking));
    }
  }
  CREATE_TRY(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
  CREATE_TRY(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
  CREATE_TRY(hipEventCreateWithFlags(&h->ev_pass, hipEventDisableTiming));
  const size_t ldz = (size_t)h->ld;
  CREATE_TRY(hipMalloc(&h->dP, sizeof(double) * (size_t)h->pstride * batch));
  CREATE_TRY(hipMalloc(&h->dmu2[0], sizeof(double) * ldz * batch));
  CREATE_TRY(hipMalloc(&h->dmu2[1], sizeof(double) * ldz * batch));
  CREATE_TRY(hipMalloc(&h->dV, sizeof(double) * ldz * KTOT * batch));
  CREATE_TRY(hipMalloc(&h->dW, sizeof(double) * ldz * KTOT * batch));
  CREATE_TRY(hipMalloc(&h->ddacc2[0], sizeof(double) * 4 * batch));
  CREATE_TRY(hipMalloc(&h->ddacc2[1], sizeof(double) * 4 * batch));
  CREATE_TRY(hipMalloc(&h->dscratch, sizeof(double) * ldz * 2));
  CREATE_TRY(hipMalloc(&h->dn, sizeof(int) * batch));
  CREATE_TRY(hipMalloc(&h->dflags, sizeof(unsigned) * batch));
  CREATE_TRY(hipMalloc(&h->dfloor, sizeof(int) * batch));
  CREATE_TRY(hipMalloc(&h->dqueue, sizeof(unsigned) * flush_rs_queue_words()));
  CREATE_TRY(hipMalloc(&h->dready, sizeof(unsigned) * batch));
  CREATE_TRY(hipMalloc(&h->dmbox, sizeof(SolveOut) * batch));
  CREATE_TRY(hipMemsetAsync(h->dmbox, 0, sizeof(SolveOut) * batch, h->stream));
  CREATE_TRY(hipMemsetAsync(h->dready, 0, sizeof(unsigned) * batch, h->stream));
  CREATE_TRY(hipMalloc(&h->dso, sizeof(SolveOut) * batch));
  CREATE_TRY(hipMalloc(&h->dfac, sizeof(double) * FACS * batch));
  CREATE_TRY(hipMalloc(&h->d_ring, sizeof(StepIn) * batch * RING));
  CREATE_TRY(hipHostMalloc(&h->h_ring, sizeof(StepIn) * batch * RING, hipHostMallocDefault));
  CREATE_TRY(hipHostMalloc(&h->h_flags, sizeof(unsigned) * batch, hipHostMallocDefault));
  for (auto& ev : h->ring_ev) CREATE_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  CREATE_TRY(hipEventCreate(&h->t0));
  CREATE_TRY(hipEventCreate(&h->t1));
  CREATE_TRY(hipMemsetAsync(h->dP, 0, sizeof(double) * (size_t)h->pstride * batch, h->stream));
  CREATE_TRY(hipMemsetAsync(h->dmu2[0], 0, sizeof(double) * ldz * batch, h->stream));
  CREATE_TRY(hipMemsetAsync(h->dmu2[1], 0, sizeof(double) * ldz * batch, h->stream));
  CREATE_TRY(hipMemsetAsync(h->dV, 0, sizeof(double) * ldz * KTOT * batch, h->stream));
  CREATE_TRY(hipMemsetAsync(h->dW, 0, sizeof(double) * ldz * KTOT * batch, h->stream));
  CREATE_TRY(hipMemsetAsync(h->ddacc2[0], 0, sizeof(double) * 4 * batch, h->stream));
  CREATE_TRY(hipMemsetAsync(h->ddacc2[1], 0, sizeof(double) * 4 * batch, h->stream));
  CREATE_TRY(hipMemsetAsync(h->dflags, 0, sizeof(unsigned) * batch, h->stream));
  CREATE_TRY(hipMemsetAsync(h->dso, 0, sizeof(SolveOut) * batch, h->stream));
  CREATE_TRY(hipMemsetAsync(h->dfac, 0, sizeof(double) * FACS * batch, h->stream));
  // reference initial state (src/replay_no_ros.py:69-70): mu = 0, P = MOTION_MODEL_VARIANCE * I3
  {
    std::vector<double> p3(3 * 3, 0.0);
    p3[0] = p3[4] = p3[8] = h->cfg.motion_sigma;
    for (int b = 0; b < batch; ++b)
      CREATE_TRY(copy_cov(h, b, p3.data(), 3, 0, 0, 3, 3, true));
    CREATE_TRY(hipMemcpyAsync(h->dn, h->n.data(), sizeof(int) * batch, hipMemcpyHostToDevice, h->stream));
    CREATE_TRY(hipMemcpyAsync(h->dfloor, h->floor_host.data(), sizeof(int) * batch, hipMemcpyHostToDevice, h->stream));
    CREATE_TRY(hipStreamSynchronize(h->stream));
  }
#undef CREATE_TRY
  *out = h;
  return EKF_OK;
}

extern "C" int ekf_destroy(ekf_handle* h) {
  if (!h) return EKF_ERR_ARG;
  free_all(h);
  return EKF_OK;
}

// After a device-side association the state may have grown on the device: read the sizes back.
static int refresh_sizes(ekf_handle* h) {
  if (!h->sizes_dirty) return EKF_OK;
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, hipMemcpyAsync(h->n.data(), h->dn, sizeof(int) * h->batch, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipMemcpyAsync(h->neff.data(), h->dneff, sizeof(int) * h->batch, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  h->neff_enq = h->neff;
  h->sizes_dirty = false;
  return EKF_OK;
}

static int check_b(ekf_handle* h, int b, const char* fn) {
  if (!h) return EKF_ERR_ARG;
  if (int rc = refresh_sizes(h)) return rc;
  if (b < 0 || b >= h->batch) return fail(h, EKF_ERR_ARG, std::string(fn) + ": trajectory index out of range");
  return EKF_OK;
}

// A bounded wait of a single-launch step that ran into its limit leaves EKF_FLAG_INTERNAL on the trajectory: the
// timed-out workgroups wrote nothing while others of the same step may have, so the trajectory's state is UNDEFINED from
// there on, until it is uploaded again (ekf_upload_state* clears the flag).  The same holds -- for every trajectory of
// the handle -- after an enqueueing call failed half way (host_bad).  Every call that hands results to the host reports
// it: EKF_ERR_STATE.  b < 0: any trajectory.  Synchronises the handle's stream (the flags come back with a stream-ordered
// copy into pinned memory, on that stream).
static int check_internal(ekf_handle* h, int b, const char* fn) {
  HIP_TRY(h, hipMemcpyAsync(h->h_flags, h->dflags, sizeof(unsigned) * h->batch, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  for (int t = (b < 0 ? 0 : b); t < (b < 0 ? h->batch : b + 1); ++t) {
    if (h->host_bad[t])
      return fail(h, EKF_ERR_STATE,
                  std::string(fn) + ": an earlier call on this handle failed after part of its work had been enqueued; the "
                      "state of trajectory " + std::to_string(t) + " is undefined: upload it again (ekf_upload_state / "
                      "ekf_upload_state_diag)");
    if (h->h_flags[t] & EKF_FLAG_INTERNAL)
      return fail(h, EKF_ERR_STATE,
                  std::string(fn) + ": trajectory " + std::to_string(t) +
                      " carries EKF_FLAG_INTERNAL (a bounded wait inside a single-launch step timed out; the state is "
                      "undefined): upload the state again (ekf_upload_state / ekf_upload_state_diag) "
                      "and consider ekf_set_option(\"fused_step\", 0)");
  }
  return EKF_OK;
}
// (an upload replaces mean and covariance of trajectory b entirely: the trajectory is good again)
static int clear_internal(ekf_handle* h, int b) {
  HIP_TRY(h, hipMemcpyAsync(h->h_flags, h->dflags + b, sizeof(unsigned), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  unsigned f = h->h_flags[0];
  if (f & EKF_FLAG_INTERNAL) {
    f &= ~EKF_FLAG_INTERNAL;
    HIP_TRY(h, hipMemcpy(h->dflags + b, &f, sizeof(unsigned), hipMemcpyHostToDevice));
  }
  h->host_bad[b] = 0;
  return EKF_OK;
}
// Stepping calls refuse a handle whose state is undefined after a half-enqueued failure (no synchronisation).
static int check_host_bad(ekf_handle* h, const char* fn) {
  for (int t = 0; t < h->batch; ++t)
    if (h->host_bad[t])
      return fail(h, EKF_ERR_STATE, std::string(fn) + ": an earlier call on this handle failed after part of its work had "
                                        "been enqueued; upload every trajectory again before stepping");
  return EKF_OK;
}

static int set_size(ekf_handle* h, int b, int n) {
  h->n[b] = n;
  HIP_TRY(h, hipMemcpyAsync(h->dn + b, &h->n[b], sizeof(int), hipMemcpyHostToDevice, h->stream));
  return EKF_OK;
}

// The active bound a step record carries (StepIn.neff) comes from the observations the host saw when the record was
// made.  k_solve raises it to this per-trajectory floor = the bound of the state when the enqueueing call starts
// (or n where the shortcut is switched off), so that a stream uploaded before the state changed -- a dense
// upload, ekf_predict_dense, other steps, a replay of the same stream -- still covers everything correlated.
// exact: the floor must equal the current bound (streams); otherwise it only must not exceed it (single steps
// carry the current bound themselves).
static int push_floor(ekf_handle* h, bool exact) {
  if (h->sizes_dirty) return EKF_OK;                   // device-side association keeps the bound on the device
  bool need = false;
  for (int b = 0; b < h->batch; ++b) {
    const int want = h->opt_active_bound ? h->neff[b] : h->n[b];
    if ((exact || !h->opt_active_bound) ? h->floor_host[b] != want : h->floor_host[b] > want) need = true;
  }
  if (!need) return EKF_OK;
  for (int b = 0; b < h->batch; ++b) h->floor_host[b] = h->opt_active_bound ? h->neff[b] : h->n[b];
  HIP_TRY(h, hipMemcpyAsync(h->dfloor, h->floor_host.data(), sizeof(int) * h->batch, hipMemcpyHostToDevice, h->stream));
  return EKF_OK;
}

extern "C" int ekf_upload_state(ekf_handle* h, int b, const double* mu, const double* P, int n) {
  if (int rc = check_b(h, b, "ekf_upload_state")) return rc;
  if (!mu || !P) return fail(h, EKF_ERR_ARG, "ekf_upload_state: NULL array");
  if (n < 3 || (n & 1) == 0 || n > h->n_max) return fail(h, EKF_ERR_ARG, "ekf_upload_state: n must be 3+2N and <= n_max");
  HIP_TRY(h, hipSetDevice(h->device));
  if (int rc = flush_pending(h)) return rc;
  if (int rc = clear_internal(h, b)) return rc;
  // the upper triangle is what the device keeps (and all it ever reads): blocks of rows, each from its first diagonal
  // column on -- 56 % of the matrix at n = 4003 (1.3 instead of 2.3 ms over PCIe)
  constexpr int UP_ROWS = 512;
  for (int r0 = 0; r0 < n; r0 += UP_ROWS)
    HIP_TRY(h, copy_cov(h, b, const_cast<double*>(P) + (size_t)r0 * n + r0, n, r0, r0, std::min(UP_ROWS, n - r0), n - r0, true));
  HIP_TRY(h, hipMemcpyAsync(h->dmu2[h->cur] + (size_t)b * h->ld, mu, sizeof(double) * n, hipMemcpyHostToDevice, h->stream));
  if (int rc = set_size(h, b, n)) return rc;
  h->neff[b] = n;                                      // arbitrary dense covariance: everything is active
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return EKF_OK;
}

extern "C" int ekf_upload_state_diag(ekf_handle* h, int b, const double* mu, const double* diagP, int n) {
  if (int rc = check_b(h, b, "ekf_upload_state_diag")) return rc;
  if (!mu || !diagP) return fail(h, EKF_ERR_ARG, "ekf_upload_state_diag: NULL array");
  if (n < 3 || (n & 1) == 0 || n > h->n_max) return fail(h, EKF_ERR_ARG, "ekf_upload_state_diag: n must be 3+2N and <= n_max");
  HIP_TRY(h, hipSetDevice(h->device));
  if (int rc = flush_pending(h)) return rc;
  if (int rc = clear_internal(h, b)) return rc;
  HIP_TRY(h, hipMemcpyAsync(h->dscratch, diagP, sizeof(double) * n, hipMemcpyHostToDevice, h->stream));
  launch_fill_diag(h->stream, h->dP + (size_t)b * h->pstride, h->ld, n, h->dscratch);
  HIP_TRY(h, hipGetLastError());
  h->neff[b] = 3;                                      // diagonal covariance: nothing is correlated yet
  HIP_TRY(h, hipMemcpyAsync(h->dmu2[h->cur] + (size_t)b * h->ld, mu, sizeof(double) * n, hipMemcpyHostToDevice, h->stream));
  if (int rc = set_size(h, b, n)) return rc;
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return EKF_OK;
}

// The pinned buffer small states come back through (k_pack_small, k_small_stream's host_out): n x n covariance, mean, flags;
// its last word is the sequence number ekf_step_fetch polls.
// (covariance, mean, flags word; behind them the integrity trailer of ekf_step_fetch -- sequence number, XOR checksum --
//  and, at a fixed place at the end, the sequence word the host polls)
constexpr size_t PACK_WORDS = (size_t)PACK_SMALL_N * PACK_SMALL_N + PACK_SMALL_N + 4;
static int pack_buffer(ekf_handle* h) {
  if (h->h_pack) return EKF_OK;
  HIP_TRY(h, hipHostMalloc(&h->h_pack, sizeof(double) * PACK_WORDS, hipHostMallocCoherent));
  std::memset(h->h_pack, 0, sizeof(double) * PACK_WORDS);
  return EKF_OK;
}

// The step kernels keep only the upper triangle of P_base current.  Before the host (or the dense product)
// looks at trajectory b: apply the pending ranks, then mirror the upper triangle into the lower one.
static int materialize(ekf_handle* h, int b) {
  if (int rc = flush_pending(h)) return rc;
  if (h->sizes_dirty)
    if (int rc = refresh_sizes(h)) return rc;
  launch_mirror(h->stream, h->dP + (size_t)b * h->pstride, h->dn + b, h->ld, h->pstride, 1, h->n[b]);
  HIP_TRY(h, hipGetLastError());
  return EKF_OK;
}

extern "C" int ekf_download_state(ekf_handle* h, int b, double* mu, double* P, int n) {
  if (int rc = check_b(h, b, "ekf_download_state")) return rc;
  if (n != h->n[b]) return fail(h, EKF_ERR_ARG, "ekf_download_state: n does not match the state size");
  HIP_TRY(h, hipSetDevice(h->device));
  if (P && mu && n <= PACK_SMALL_N) {
    // a small state: flush, then ONE kernel writes covariance (mirrored from the stored triangle), mean and flags into
    // pinned host memory; one synchronisation for everything (see k_pack_small)
    if (h->host_bad[b]) return check_internal(h, b, "ekf_download_state");
    if (int rc = flush_pending(h)) return rc;
    if (int rc = pack_buffer(h)) return rc;
    launch_pack_small(h->stream, h->dP + (size_t)b * h->pstride, h->dmu2[h->cur] + (size_t)b * h->ld, h->dflags + b, h->ld, n, h->h_pack);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if ((unsigned)h->h_pack[(size_t)n * n + n] & EKF_FLAG_INTERNAL) return check_internal(h, b, "ekf_download_state");
    std::memcpy(P, h->h_pack, sizeof(double) * (size_t)n * n);
    std::memcpy(mu, h->h_pack + (size_t)n * n, sizeof(double) * n);
    return EKF_OK;
  }
  if (int rc = check_internal(h, b, "ekf_download_state")) return rc;
  if (P) {
    // a destination in pinned host memory (what the Python binding hands out for large covariances: ekf_host_alloc) is
    // written by a kernel -- mirrored on the way, no SDMA copy (k_pack_dense); anything else by the mirror pass + rectangle copy
    hipPointerAttribute_t attr{};
    double* dst = nullptr;
    // (up to 40 MB -- N = 1100: there the kernel is as fast as the copy engine at its best, 38 - 47 GB/s, and does not have
    //  the copy's bad days; beyond, the engine's larger PCIe payloads win: 54.6 against 49.8 GB/s at N = 2000.  tools/download_paths.py)
    const bool pack = h->opt_pack_dense == 2 || (h->opt_pack_dense == 1 && (size_t)n * n * sizeof(double) <= (40u << 20));
    if (pack && hipPointerGetAttributes(&attr, P) == hipSuccess && attr.type == hipMemoryTypeHost)
      dst = static_cast<double*>(attr.devicePointer);
    else
      (void)hipGetLastError();                     // (an ordinary pointer is "invalid value" to the query)
    if (dst) {
      if (int rc = flush_pending(h)) return rc;    // the covariance is P_base + pending ranks, upper triangle
      launch_pack_dense(h->stream, h->dP + (size_t)b * h->pstride, h->ld, n, dst);
      HIP_TRY(h, hipGetLastError());
      h->dense_packs += 1;
    } else {
      if (int rc = materialize(h, b)) return rc;
      HIP_TRY(h, copy_cov(h, b, P, n, 0, 0, n, n, false));
    }
  }
  if (mu)
    HIP_TRY(h, hipMemcpyAsync(mu, h->dmu2[h->cur] + (size_t)b * h->ld, sizeof(double) * n, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return EKF_OK;
}

extern "C" int ekf_download_block(ekf_handle* h, int b, int r0, int c0, int rows, int cols, double* out) {
  if (int rc = check_b(h, b, "ekf_download_block")) return rc;
  const int n = h->n[b];
  if (!out || rows <= 0 || cols <= 0 || r0 < 0 || c0 < 0 || r0 + rows > n || c0 + cols > n)
    return fail(h, EKF_ERR_ARG, "ekf_download_block: block outside the state");
  HIP_TRY(h, hipSetDevice(h->device));
  if (int rc = check_internal(h, b, "ekf_download_block")) return rc;
  if (int rc = materialize(h, b)) return rc;
  HIP_TRY(h, copy_cov(h, b, out, cols, r0, c0, rows, cols, false));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return EKF_OK;
}

extern "C" int ekf_download_mean(ekf_handle* h, int b, double* mu, int n) {
  return ekf_download_state(h, b, mu, nullptr, n);
}

extern "C" int ekf_state_size(ekf_handle* h, int b, int* n) {
  if (int rc = check_b(h, b, "ekf_state_size")) return rc;
  if (!n) return fail(h, EKF_ERR_ARG, "ekf_state_size: NULL");
  *n = h->n[b];
  return EKF_OK;
}

extern "C" int ekf_add_landmarks(ekf_handle* h, int b, int first_index, const double* xy, int k) {
  if (int rc = check_b(h, b, "ekf_add_landmarks")) return rc;
  if (k <= 0) return EKF_OK;
  if (!xy) return fail(h, EKF_ERR_ARG, "ekf_add_landmarks: NULL xy");
  const int n_old = h->n[b], n_new = n_old + 2 * k;
  if (first_index != (n_old - 3) / 2)
    return fail(h, EKF_ERR_ARG, "ekf_add_landmarks: first_index must continue the landmark count (replay_no_ros.py:294-295)");
  if (n_new > h->n_max) return fail(h, EKF_ERR_ARG, "ekf_add_landmarks: state would exceed n_max");
  if (2 * k > 2 * h->ld) return fail(h, EKF_ERR_ARG, "ekf_add_landmarks: too many landmarks in one call");
  HIP_TRY(h, hipSetDevice(h->device));
  if (int rc = flush_pending(h)) return rc;
  HIP_TRY(h, hipStreamSynchronize(h->stream));   // xy is staged through a single scratch buffer
  HIP_TRY(h, hipMemcpyAsync(h->dscratch, xy, sizeof(double) * 2 * k, hipMemcpyHostToDevice, h->stream));
  launch_add_landmarks(h->stream, h->dP + (size_t)b * h->pstride, h->dmu2[h->cur] + (size_t)b * h->ld, h->ld, n_old, n_new,
                       h->cfg.landmark_init_var, h->dscratch);
  HIP_TRY(h, hipGetLastError());
  if (int rc = set_size(h, b, n_new)) return rc;
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return EKF_OK;
}

// ---- step machinery -------------------------------------------------------------------------
static int prof_event(ekf_handle* h, hipEvent_t* ev, int cls = 0) {
  if ((h->prof_used & 1) == 0) {
    if (h->prof_cls.size() <= h->prof_used / 2) h->prof_cls.resize(h->prof_used / 2 + 1);
    h->prof_cls[h->prof_used / 2] = cls;
  }
  if (h->prof_used == h->prof_pool.size()) {
    hipEvent_t e;
    HIP_TRY(h, hipEventCreate(&e));
    h->prof_pool.push_back(e);
  }
  *ev = h->prof_pool[h->prof_used++];
  return EKF_OK;
}
// (diagnostic) an event pair around the launches between prof_open and prof_close on stream `st`, class `cls`
struct ProfBracket { hipEvent_t e1 = nullptr; hipStream_t st = nullptr; };
static int prof_open(ekf_handle* h, int cls, hipStream_t st, ProfBracket* pb) {
  pb->e1 = nullptr;
  if (!h->profile || !h->opt_profile_kernels) return EKF_OK;
  hipEvent_t e0;
  if (int rc = prof_event(h, &e0, cls)) return rc;
  if (int rc = prof_event(h, &pb->e1, cls)) return rc;
  pb->st = st;
  HIP_TRY(h, hipEventRecord(e0, st));
  return EKF_OK;
}
static int prof_close(ekf_handle* h, ProfBracket* pb) {
  if (pb->e1) HIP_TRY(h, hipEventRecord(pb->e1, pb->st));
  return EKF_OK;
}

// Apply the pending low-rank update to P_base:  P_base += W V + diag(dacc)  (one pass over P), on stream `st` (the
// handle's own unless the look-ahead of ekf_stream_run sends it to the second one).
static int flush_pending(ekf_handle* h, hipStream_t st, const CadOut* wv) {
  if (h->pending_k == 0) return EKF_OK;
  if (!st) st = h->stream;
  const PassPlan p = plan_pass(h);
  const int* shares = nullptr;
  if (p.kernel == 2 && p.long_few) {
    // build_pass_shares depends on the size only through the number of slabs and the last strip: with the active bound
    // on and a growing map the bound changes at almost every pass, the table only when it crosses a strip
    const int nrb = (p.e_hi + 127) / 128, s_last = (p.e_hi - 1) >> 6;
    const int key[5] = {h->batch, nrb, s_last, p.rs_workgroups, h->opt_share_order};
    if (std::memcmp(key, h->shares_key, sizeof key) != 0) {
      const size_t words = (size_t)h->cu_count * pass_share_pieces() * 4;
      const int nb = h->shares_cur ^ 1;
      if (!h->dshares2[nb]) {
        HIP_TRY(h, hipMalloc(&h->dshares2[nb], sizeof(int) * words));
        HIP_TRY(h, hipHostMalloc(&h->hshares2[nb], sizeof(int) * words, hipHostMallocDefault));
        std::memset(h->hshares2[nb], 0, sizeof(int) * words);
        HIP_TRY(h, hipEventCreateWithFlags(&h->shares_ev[nb], hipEventDisableTiming));
      }
      // (the pinned copy is free once its previous upload has been executed: two tables back, long ago)
      if (h->shares_ev_used[nb]) HIP_TRY(h, hipEventSynchronize(h->shares_ev[nb]));
      h->shares_ok = build_pass_shares(h->batch, p.e_hi, p.rs_workgroups, h->hshares2[nb]);
      if (h->shares_ok > 0 && h->opt_share_order) order_pass_shares(p.rs_workgroups, pass_share_pieces(), h->hshares2[nb], words);
      // stream-ordered: the launch below, on the same stream, reads the table after the copy; the pass that read the
      // other table -- possibly still running on the handle's other stream -- is not touched
      HIP_TRY(h, hipMemcpyAsync(h->dshares2[nb], h->hshares2[nb], sizeof(int) * words, hipMemcpyHostToDevice, st));
      HIP_TRY(h, hipEventRecord(h->shares_ev[nb], st));
      h->shares_ev_used[nb] = true;
      h->shares_cur = nb;
      std::memcpy(h->shares_key, key, sizeof key);
    }
    if (h->shares_ok > 0) shares = h->dshares2[h->shares_cur];
  }
  hipEvent_t e0 = nullptr, e1 = nullptr;
  // (profiling: every `profile_stride`-th launch of the pass is bracketed by an event pair -- a record costs its stream ~6 us)
  const bool timed = h->profile && (h->prof_seen++ % h->profile_stride) == 0;
  if (timed) {
    if (int rc = prof_event(h, &e0)) return rc;
    if (int rc = prof_event(h, &e1)) return rc;
    HIP_TRY(h, hipEventRecord(e0, st));
  }
  h->last_kernel = p.kernel;
  h->last_wv = (p.kernel == 2 && wv) ? 1 : 0;
  h->last_nkt = p.nkt;
  h->last_streaming = p.streaming ? 1 : 0;
  h->last_shares = shares ? h->shares_ok : 0;
  if (p.kernel == 2) {                                 // (the step before left the queue heads at zero)
    launch_flush_rs(st, p.streaming, h->dP, h->dV, h->dW, h->ddacc2[h->dcur], h->dn, h->dso, h->ld, h->pstride,
                    h->batch, p.e_hi, p.nkt, p.rs_workgroups, h->dqueue, h->opt_pass_chunk, shares, p.kernel == 2 ? wv : nullptr);
  } else {
    launch_flush(st, p.streaming, h->dP, h->dV, h->dW, h->ddacc2[h->dcur], h->dn, h->dso, h->ld, h->pstride, h->batch,
                 p.e_hi, p.nkt, flush_rows_per_block(h, p.streaming, p.e_hi));
  }
  if (timed) HIP_TRY(h, hipEventRecord(e1, st));
  HIP_TRY(h, hipGetLastError());
  h->pending_k = 0;                                    // (with no rank pending k_solve takes the pending noise as zero: no clearing)
  h->pending_steps = 0;
  return EKF_OK;
}
static int flush_pending(ekf_handle* h) { return flush_pending(h, nullptr); }

// The small-state path (ekf_small.hip): a filter bank whose covariances fit the LDS of a CU runs `nsteps` steps per trajectory
// inside one workgroup, P resident in LDS; nothing is ever pending on it.
static bool small_path(const ekf_handle* h) {
  return h->opt_small_state && h->n_max <= small_state_limit(h->batch) && h->pending_k == 0;
}
static int enqueue_small(ekf_handle* h, const StepIn* d_in, int nsteps) {
  const int n_hi = h->sizes_dirty ? h->n_max : *std::max_element(h->n.begin(), h->n.end());
  const int out_b = h->fetch_b;                        // (ekf_step_fetch, last pass of its step: see there)
  h->fetch_b = -1;
  if (launch_small_stream(h->stream, h->dP, h->dmu2[h->cur], h->dmu2[h->cur ^ 1], h->dn, d_in, h->batch, nsteps, h->dflags,
                          h->dcfg, h->ld, h->pstride, n_hi, out_b >= 0 ? h->h_pack : nullptr, out_b,
                          out_b >= 0 ? reinterpret_cast<unsigned long long*>(h->h_pack + PACK_WORDS - 1) : nullptr,
                          out_b >= 0 ? ++h->fetch_seq : 0ull, h->batch > 3 * h->cu_count) != 0)
    return fail(h, EKF_ERR_HIP, "small-state launch: hipFuncSetAttribute failed");
  HIP_TRY(h, hipGetLastError());
  h->fetched = out_b >= 0;
  h->cur ^= 1;
  h->small_launches += 1;
  return EKF_OK;
}

// Enqueue one device pass with inputs already at d_in (StepIn[batch]); m_hi = max m over the batch.
static int enqueue_pass(ekf_handle* h, const StepIn* d_in, int m_hi) {
  if (small_path(h)) return enqueue_small(h, d_in, 1);
  const int n_hi = h->sizes_dirty ? h->n_max : *std::max_element(h->n.begin(), h->n.end());
  const int mcap = cap_for(m_hi);
  const int ktp = ranks_for(mcap);
  const double* mu_in = h->dmu2[h->cur];
  double* mu_out = h->dmu2[h->cur ^ 1];
  const double* dacc_in = h->ddacc2[h->dcur];
  double* dacc_out = h->ddacc2[h->dcur ^ 1];
  if (m_hi == 0 && h->pending_k == 0) {
    // prediction only, nothing pending: rows/cols 0,1 of P_base directly, O(n)
    launch_solve(h->stream, h->dP, h->dV, h->dW, dacc_in, dacc_out, mu_in, mu_out, h->dn, d_in, h->dso, h->dflags,
                 h->dfac, h->dfloor, h->dqueue, h->dcfg, h->ld, h->pstride, h->batch, 0);
    launch_predict_rc(h->stream, h->dP, mu_in, mu_out, h->dn, h->dso, h->ld, h->pstride, h->batch, n_hi);
    // k_predict_rc applied the noise itself (and nothing reads the pending-noise buffers while no rank is pending)
    HIP_TRY(h, hipGetLastError());
    h->cur ^= 1;
    return EKF_OK;
  }
  // The kernels WRITE the rank slots of `mcap` landmarks behind the pending ones (zeros where a trajectory observes fewer) plus
  // the k-tile pad; the step is CHARGED the ranks of the busiest trajectory only (round 5: 2 per landmark, as the packed
  // cadences do) -- the next step starts right behind them and overwrites the zeros.  m = 5: 7 steps per pass (until round 4:
  // 5, the count rounded up to 8 landmarks), m = 12: 3 (2).
  if (((h->pending_k + ktp + 3) & ~3) > KTOT)
    if (int rc = flush_pending(h)) return rc;
  dacc_in = h->ddacc2[h->dcur];
  dacc_out = h->ddacc2[h->dcur ^ 1];
  if (h->opt_fused_step && step_is_split(h->batch, n_hi, h->cu_count)) {
    // few workgroups (the latency regime): the whole step as one launch, the panels gathered beside the solve
    launch_step_split(h->stream, mcap, h->dP, h->dV, h->dW, dacc_in, dacc_out, mu_in, mu_out, h->dn, d_in, h->dso,
                      h->dflags, h->dfac, h->dfloor, h->dqueue, h->dready, ++h->step_seq, h->opt_fused_step == 1, h->dcfg, h->ld,
                      h->pstride,
                      h->batch, n_hi, h->pending_k);
  } else if (h->opt_fused_step && mcap <= 8 && (long)((n_hi + 63) / 64) * h->batch > 512 &&
             (long)(1 + (n_hi + 255) / 256) * h->batch <= 2L * h->cu_count) {
    // the throughput shape of the panels with room left on the chip for one more workgroup per trajectory: still one
    // launch -- workgroup 0 of a trajectory solves, the others gather their panels meanwhile (k_panels<.., SPLIT>)
    launch_step_split_tp(h->stream, mcap, h->dP, h->dV, h->dW, dacc_in, dacc_out, mu_in, mu_out, h->dn, d_in, h->dso,
                         h->dflags, h->dfac, h->dfloor, h->dqueue, h->dmbox, h->dready, ++h->step_seq, h->opt_fused_step == 1,
                         h->dcfg, h->ld, h->pstride, h->batch, n_hi, h->pending_k);
  } else {
    launch_solve(h->stream, h->dP, h->dV, h->dW, dacc_in, dacc_out, mu_in, mu_out, h->dn, d_in, h->dso, h->dflags,
                 h->dfac, h->dfloor, h->dqueue, h->dcfg, h->ld, h->pstride, h->batch, h->pending_k);
    launch_panels(h->stream, mcap, h->dP, h->dV, h->dW, mu_in, mu_out, h->dn, h->dso, h->dfac, h->ld, h->pstride,
                  h->batch, n_hi);
  }
  HIP_TRY(h, hipGetLastError());
  h->dcur ^= 1;
  h->cur ^= 1;
  h->pending_k += 2 * m_hi;
  h->pending_steps += 1;
  // cadence of the covariance pass: a fixed number of steps if asked for, otherwise as many steps as fit
  // `rank_limit` pending ranks (default 80) -- 5 steps at m = 8, 10 at m = 4, 40 at m = 1
  const bool due = h->opt_flush_every > 0 ? h->pending_steps >= h->opt_flush_every
                                          : h->pending_k + std::max(2 * m_hi, 2) > h->opt_rank_limit;   // (a step like this one would not fit)
  if (due || h->pending_k + 2 > KTOT)
    if (int rc = flush_pending(h)) return rc;
  return EKF_OK;
}

// The plan of a run's cadences (h->run_plan) to the device: stream-ordered out of pinned memory, no host synchronisation.
static int upload_run_plan(ekf_handle* h) {
  const size_t count = h->run_plan.entries.size();
  const int nb = h->plan_cur ^ 1;
  if (h->plan_cap2[nb] < count) {
    const size_t cap = std::max(count, (size_t)64 * h->batch);
    if (h->plan_ev_used[nb]) HIP_TRY(h, hipEventSynchronize(h->plan_ev[nb]));
    HIP_TRY(h, hipStreamSynchronize(h->stream));       // (kernels of an earlier run may still read the old device copy)
    if (h->dplan2[nb]) HIP_TRY(h, hipFree(h->dplan2[nb]));
    if (h->hplan2[nb]) HIP_TRY(h, hipHostFree(h->hplan2[nb]));
    h->dplan2[nb] = nullptr;
    h->hplan2[nb] = nullptr;
    h->plan_cap2[nb] = 0;
    HIP_TRY(h, hipMalloc(&h->dplan2[nb], sizeof(CadPlan) * cap));
    HIP_TRY(h, hipHostMalloc(&h->hplan2[nb], sizeof(CadPlan) * cap, hipHostMallocDefault));
    if (!h->plan_ev[nb]) HIP_TRY(h, hipEventCreateWithFlags(&h->plan_ev[nb], hipEventDisableTiming));
    h->plan_cap2[nb] = cap;
    h->plan_ev_used[nb] = false;
  }
  // (the pinned copy is free once its previous upload has been executed: two runs back)
  if (h->plan_ev_used[nb]) HIP_TRY(h, hipEventSynchronize(h->plan_ev[nb]));
  std::memcpy(h->hplan2[nb], h->run_plan.entries.data(), sizeof(CadPlan) * count);
  HIP_TRY(h, hipMemcpyAsync(h->dplan2[nb], h->hplan2[nb], sizeof(CadPlan) * count, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipEventRecord(h->plan_ev[nb], h->stream));
  h->plan_ev_used[nb] = true;
  h->plan_cur = nb;
  return EKF_OK;
}

// Cadence c of the run in flight (h->run_plan, uploaded): one solve launch, one panel launch; the covariance pass follows
// when it is due -- behind every cadence but the run's last, and behind that one when its slots are used up.
//
// Small launches, where the covariance pass leaves CUs free (the column-strip kernel on at least ~48 MB of covariance; the
// row-slab pass on static shares: a few long trajectories), do not run solve -> panel -> pass -> solve one behind the
// other: the only true dependency between two cadences of a trajectory is the sequential landmark recurrence
// (src/replay_no_ros.py:436-480: landmark j + 1 is linearised at the mean landmark j produced).
//   * CHAINED SOLVES (round 6, "chain" = 1): every solve of the run also records the pose block behind its cadence
//     (k_solve_cad<true>: CadOut::posefin); k_chain_cad forms the next cadence's block and mean from the cadence's records and
//     from P_base as it stood BEFORE the cadence, so the handle's stream runs  solve_c -> chain_{c+1} -> solve_{c+1} -> ...  while
//     the second stream runs  gate -> panel_c -> pass_c -> mark -> gate -> ...  Hand-overs are device-scope counters (ekf_cadence.hip:
//     SYNC_*): the gate waits for solve_c (announced by chain_{c+1}'s start); panel_c ends when chain_{c+1}'s gather workgroups
//     have read what the pass rewrites and solve_{c+1} has been placed; chain_{c+1}'s gathers wait for the mark behind pass_{c-1}.
//     A transition is ENQUEUED chain, solve, gate, panel, pass, mark: every wait is for an earlier-enqueued launch.  Two copies of
//     the records, of the pose rows (dprow3) and of the inputs formed ahead (CadPre), used alternately.
//   * LOOK-AHEAD (round 3, "chain" = 0): panel_c -> k_gather_cad (the next block from P_base and the ranks still pending) ->
//     { pass_c on the second stream | solve_{c+1} } -> join.
// `presolved` says that this cadence's solve has already been enqueued one of these ways; *next_presolved that the next
// one's now is.
static bool beside_the_pass(const ekf_handle* h, const PassPlan& plan) {
  // (worth it where the pass is the column-strip kernel -- the row-slab pass fills every CU by itself -- and long enough
  //  to pay for the gather and the two cross-stream hand-overs, ~25 us together: from ~48 MB of covariance.  N = 2000 x 1:
  //  38.7 k -> 45.1 k steps/s, x 2: 57.6 k -> 61.9 k, x 4: 89.5 k -> 92.1 k; N = 500 x 1 and N = 20 x 1 lose 4 - 9 %;
  //  N = 8000 x 1 on static shares, the pass on 255 workgroups: 9.35 - 9.58 k -> 9.82 - 10.2 k)
  // ... or the row-slab pass on static shares that leaves the solves their CUs (a few long trajectories: N = 8000 x 1)
  // ... and for banks of up to 40 trajectories: every solve workgroup has to find a CU beside the pass, and the gather grows with
  // the bank (17 us at 32 trajectories, 71 us at 256) -- N = 500 x 32 +5 %, N = 300 x 48 -5 %, N = 200 x 128 -21 %,
  // N = 100 x 256 -36 % with the look-ahead (bench.py --option lookahead=0; round 4)
  const bool small_pass = plan.kernel == 0 && h->batch <= 40 &&
                          (double)h->batch * 8.0 * plan.e_hi * plan.e_hi >= 1.0e6 * (h->opt_chain ? h->opt_beside_min_mb : h->opt_lookahead_min_mb);
  const bool shares_pass = plan.kernel == 2 && (plan.beside || (plan.long_few && h->batch < 8 && h->opt_pass_workgroups > 0 &&
                                                                  h->opt_pass_workgroups + h->batch <= h->cu_count));
  return small_pass || shares_pass;
}

// whatever is still running on the second stream is waited for by the handle's own stream
static int join_aux(ekf_handle* h) {
  if (!h->aux_pass) return EKF_OK;
  HIP_TRY(h, hipEventRecord(h->ev_pass, h->aux));      // (behind the last chained pass: one event per chain of cadences)
  HIP_TRY(h, hipStreamWaitEvent(h->stream, h->ev_pass, 0));
  h->aux_pass = false;
  return EKF_OK;
}

static int enqueue_cadence(ekf_handle* h, int c, bool presolved, bool* next_presolved) {
  *next_presolved = false;
  const RunPlan& rp = h->run_plan;
  const int n_hi = *std::max_element(h->n.begin(), h->n.end());
  const CadPlan* dpl = h->dplan2[h->plan_cur] + (size_t)c * h->batch;
  for (int i = 0; i < 2; ++i)
    if (!h->dcad2[i]) HIP_TRY(h, hipMalloc(&h->dcad2[i], sizeof(CadOut) * h->batch));
  CadOut* dcad = h->dcad2[h->cpar];
  double* prow_out = h->chain_run ? h->dprow3[h->cpar] : nullptr;
  const long serial = h->cad_serial++;                 // this cadence's number (CadPre copies are looked up by it)
  for (int b = 0; b < h->batch; ++b) h->neff_enq[b] = rp.entries[(size_t)c * h->batch + b].neff;
  const double* mu_in = h->dmu2[h->cur];
  double* mu_out = h->dmu2[h->cur ^ 1];
  if (!presolved) {
    if (int rc = join_aux(h)) return rc;
    // (the chain runs on one CU per trajectory: the rest of the chip gathers the panel launch's mirrored column entries
    //  meanwhile -- where there is a rest, and something to gather)
    const size_t cb_bytes = sizeof(double) * (size_t)h->batch * CAD_CU * h->ld;
    double* colbuf = nullptr;
    // (while its items -- trajectories x strips of 64 state indices -- are at most four rounds of the idle CUs' waves: N = 2000:
    //  up to ~80 trajectories; x 64 +1 - 2 %, x 128 -2 % on scattered landmarks, profiles/r05_scattered_indices.txt)
    const long col_items = (long)h->batch * ((n_hi + 63) / 64), col_waves = 8L * (h->cu_count - h->batch);
    if (h->opt_col_gather && cb_bytes <= ((size_t)1 << 30) && col_waves > 0 && col_items <= 4 * col_waves && rp.slots_hi[c] > 0) {
      // (an optional optimisation: where its buffer cannot be had the panel launch gathers everything itself, bit-identically)
      if (!h->dcolbuf && hipMalloc(&h->dcolbuf, cb_bytes) != hipSuccess) {
        (void)hipGetLastError();
        h->dcolbuf = nullptr;
        h->opt_col_gather = 0;
      }
      colbuf = h->dcolbuf;
    }
    const int col_wgs = h->cu_count - h->batch;
    ProfBracket pb;
    if (int rc = prof_open(h, 1, h->stream, &pb)) return rc;
    launch_solve_cad(h->stream, h->dP, mu_in, mu_out, h->ddacc2[h->dcur ^ 1], h->dn, h->d_stream, dpl, h->batch, dcad,
                     h->dflags, h->dcfg, h->ld, h->pstride, nullptr, 0, colbuf, n_hi, col_wgs, h->chain_run, nullptr, nullptr, 0u, nullptr);
    if (int rc = prof_close(h, &pb)) return rc;
    h->colbuf_live = colbuf != nullptr;
  }
  const int ranks = 2 * rp.slots_hi[c], nrp = (ranks + 3) & ~3;   // every trajectory writes the busiest one's ranks (zeros beyond its own)
  // what follows the panel launch is decided before it is launched (chained: it goes to the second stream)
  const bool more = c + 1 < rp.ncad;
  const int pend_after = h->pending_k + ranks, steps_after = h->pending_steps + rp.steps_hi[c];
  const bool due = pend_after > 0 && (more || pend_after + 2 > std::min(KTOT, h->opt_rank_limit) ||
                                      (h->opt_flush_every > 0 && steps_after >= h->opt_flush_every));
  bool beside = false;
  if (due && more && h->opt_lookahead) {
    const int pk = h->pending_k;
    h->pending_k = pend_after;                         // (plan_pass reads the handle)
    beside = beside_the_pass(h, plan_pass(h));
    h->pending_k = pk;
  }
  const bool chain_next = beside && h->chain_run;
  // ("w_from_v") the pass follows this panel launch at once, nothing else is pending, both take the forms that know how
  bool wv = false;
  if (h->opt_w_from_v && due && !beside && h->pending_k == 0 && ranks > 0 &&      // (... and the panel launch takes a replay shape)
      (h->opt_panel_shape >= 2 || (h->opt_panel_shape == 0 && !panels_cad_latency_regime(h->batch, n_hi)))) {
    h->pending_k = pend_after;
    wv = plan_pass(h).kernel == 2;
    h->pending_k = 0;
  }
  hipStream_t pst = h->stream;                         // the panel launch's stream
  unsigned* psync = nullptr;
  unsigned head_sigma = 0u, tail_target = 0u;
  const CadPlan* dpl2 = dpl + h->batch;
  int rc = EKF_OK;
  if (chain_next) {
    // ---- chained.  The handle's stream: chain_{c+1} -> solve_{c+1}; the second stream: [gate ->] panel_c -> pass_c -> mark.
    // No event: a hand-over through the command processor costs the waiting stream 7 us behind a record and ~19 us across
    // streams (profiles/r06_chained_solves.txt); the counters cost a load.  ENQUEUED IN THIS ORDER: every device-side wait is
    // for a launch that is already in its queue (chain_{c+1}'s gathers: the previous transition's mark; panel_c: chain_{c+1}'s
    // start and gathers, solve_{c+1}'s start), so nothing can hang where the runtime maps both streams onto one hardware queue --
    // and the host calls nothing that may block in between (every buffer exists before the run).
    h->sigma += 1u;
    const int gw = chain_gather_workgroups(h->batch, h->cu_count);
    h->gather_count += (unsigned)(h->batch * gw);
    // (dmu2[cur] is the mean cadence c reads -- its landmark entries are the mean before the cadence --, dmu2[cur ^ 1] the one
    //  its solve left the pose in; dcad2[cpar] cadence c's records, dprow3[cpar ^ 1] the pose rows BEFORE it)
    ProfBracket pbc, pbs;
    if (int rc2 = prof_open(h, 2, h->stream, &pbc)) return rc2;
    // the next cadence's inputs if an earlier chain launch formed them; the one after it: formed by this launch
    const CadPre* pre_in = (h->opt_pre_positions && h->pre_serial[(serial + 1) & 1] == serial + 1) ? h->dpre[(serial + 1) & 1] : nullptr;
    CadPre* pre_out = (h->opt_pre_positions && c + 2 < rp.ncad) ? h->dpre[(serial + 2) & 1] : nullptr;
    launch_chain_cad(h->stream, h->dP, h->dprow3[h->cpar ^ 1], h->dmu2[h->cur], h->dmu2[h->cur ^ 1], dcad, h->d_stream,
                     dpl2, h->batch, h->dcfg, h->ld, h->pstride, h->dgbuf, h->dgmu, h->dxg, h->dbg, h->dsync, h->gather_count,
                     h->dflags, gw, h->sigma, pre_in, pre_out, pre_out ? dpl2 + h->batch : nullptr, h->aux_pass);
    if (pre_out) h->pre_serial[(serial + 2) & 1] = serial + 2;
    if (int rc2 = prof_close(h, &pbc)) return rc2;
    if (int rc2 = prof_open(h, 1, h->stream, &pbs)) return rc2;
    launch_solve_cad(h->stream, h->dP, h->dmu2[h->cur ^ 1], h->dmu2[h->cur], h->ddacc2[h->dcur], h->dn, h->d_stream, dpl2,
                     h->batch, h->dcad2[h->cpar ^ 1], h->dflags, h->dcfg, h->ld, h->pstride, h->dgbuf, 1, nullptr, n_hi, 0, true, h->dgmu,
                     h->dsync, h->sigma, pre_in);
    if (int rc2 = prof_close(h, &pbs)) return rc2;
    // From here on the next cadence's solve overwrites the pose mean and the pending-noise buffer: a failure below cannot be
    // undone.  Whatever happens the streams are joined, and a failure marks every trajectory undefined (EKF_ERR_STATE from
    // then on, until it is uploaded again).
    if (hipGetLastError() != hipSuccess) rc = fail(h, EKF_ERR_HIP, "chained solves: launch of the next cadence's solve failed");
    // (a one-lane gate in front of the panel launch; with "panel_own_gate" a small panel launch -- each of its workgroups and each
    //  solve workgroup a CU to itself -- waits for its solve itself)
    if (h->opt_panel_own_gate && panels_cad_workgroups(h->batch, n_hi) + 2 * h->batch <= h->cu_count / 2) head_sigma = h->sigma;
    else launch_gate(h->aux, h->dsync, h->sigma, h->dflags, h->batch);
    pst = h->aux;
    psync = h->dsync;
    tail_target = h->gather_count;
  } else if (int rc2 = join_aux(h)) {
    return rc2;
  }
  {
    ProfBracket pb;
    if (int rc2 = prof_open(h, 3, pst, &pb)) return rc2;
    launch_panels_cad(pst, h->dP, h->dV, h->dW, mu_in, mu_out, h->dn, dcad, h->dso, h->dqueue, h->ld,
                      h->pstride, h->batch, n_hi, nrp, h->colbuf_live ? h->dcolbuf : nullptr, prow_out, psync, head_sigma, tail_target, h->dflags,
                      chain_next && h->opt_panel_tform && !h->colbuf_live && panels_cad_latency_regime(h->batch, n_hi),
                      chain_next ? h->sigma : 0u, h->opt_panel_shape, wv);
    if (int rc2 = prof_close(h, &pb)) return rc2;
  }
  h->colbuf_live = false;
  if (hipGetLastError() != hipSuccess && rc == EKF_OK) rc = fail(h, EKF_ERR_HIP, "fused cadence: launch of the panel kernel failed");
  h->dcur ^= 1;
  h->cur ^= 1;
  h->cpar ^= 1;
  h->pending_k += ranks;
  h->pending_steps += rp.steps_hi[c];
  h->cadences += 1;
  h->cadence_traj_steps += rp.steps_sum[c];
  if (chain_next) {
    // pass_c behind the panel launch on the second stream, then the mark the next chain launch's gather workgroups wait for
    if (rc == EKF_OK) rc = flush_pending(h, h->aux);
    launch_mark(h->aux, h->dsync, h->sigma);
    if (hipGetLastError() != hipSuccess && rc == EKF_OK) rc = fail(h, EKF_ERR_HIP, "chained solves: launch of the mark failed");
    h->aux_pass = true;
    if (rc != EKF_OK) {
      (void)hipStreamSynchronize(h->aux);
      (void)hipStreamSynchronize(h->stream);
      h->aux_pass = false;
      std::fill(h->host_bad.begin(), h->host_bad.end(), (unsigned char)1);
      return rc;
    }
    h->chained += 1;
    h->lookaheads += 1;
    *next_presolved = true;
    return EKF_OK;
  }
  if (rc != EKF_OK) return rc;
  if (h->pending_k == 0) {                             // nothing observed anywhere in the bank: the panel launch has applied the noise
    h->pending_steps = 0;
    return EKF_OK;
  }
  if (!due) return EKF_OK;
  if (!beside) {
    if (wv) h->w_from_v_passes += 1;
    return flush_pending(h, nullptr, wv ? dcad : nullptr);
  }
  if (!h->dgbuf) HIP_TRY(h, hipMalloc(&h->dgbuf, sizeof(double) * cadence_gbuf_doubles() * h->batch));
  // ---- look-ahead: gather (stream) -> { pass (second stream) | solve of the next cadence (stream) } -> join ----
  const int kb = (h->pending_k + 3) & ~3;
  {
    ProfBracket pb;
    if (int rc2 = prof_open(h, 2, h->stream, &pb)) return rc2;
    launch_gather_cad(h->stream, h->dP, h->dV, h->dW, h->ddacc2[h->dcur], h->d_stream, dpl2, h->batch, kb, h->dcfg, h->ld,
                      h->pstride, h->dgbuf);
    if (int rc2 = prof_close(h, &pb)) return rc2;
  }
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, hipEventRecord(h->ev_fork, h->stream));
  HIP_TRY(h, hipStreamWaitEvent(h->aux, h->ev_fork, 0));
  // (the solve first: it is ready to go the moment the gather ends, the pass has an event to wait for -- the one
  //  workgroup per trajectory finds its CU before the pass fills the chip)
  {
    ProfBracket pb;
    if (int rc2 = prof_open(h, 1, h->stream, &pb)) return rc2;
    launch_solve_cad(h->stream, h->dP, h->dmu2[h->cur], h->dmu2[h->cur ^ 1], h->ddacc2[h->dcur ^ 1], h->dn, h->d_stream, dpl2,
                     h->batch, h->dcad2[h->cpar], h->dflags, h->dcfg, h->ld, h->pstride, h->dgbuf, (kb + 7) / 8, nullptr, n_hi, 0,
                     h->chain_run, nullptr, nullptr, 0u, nullptr);
    if (int rc2 = prof_close(h, &pb)) return rc2;
  }
  h->colbuf_live = false;                              // (beside the pass P_base is in motion: that cadence's panel launch gathers itself)
  // From here on the next cadence's solve has overwritten the pose mean and the pending-noise buffer: a failure
  // below cannot be undone.  Whatever happens the two streams are joined again, and a failure marks every trajectory
  // undefined (EKF_ERR_STATE from then on, until it is uploaded again).
  if (hipGetLastError() != hipSuccess) rc = fail(h, EKF_ERR_HIP, "look-ahead: launch of the next cadence's solve failed");
  if (rc == EKF_OK) rc = flush_pending(h, h->aux);
  const hipError_t ej = hipEventRecord(h->ev_join, h->aux);
  const hipError_t ew = ej == hipSuccess ? hipStreamWaitEvent(h->stream, h->ev_join, 0) : ej;   // whatever follows on the handle's stream follows the pass
  if (rc == EKF_OK && ew != hipSuccess) rc = fail(h, EKF_ERR_HIP, std::string("look-ahead: joining the streams failed: ") + hipGetErrorString(ew));
  if (rc != EKF_OK) {
    (void)hipStreamSynchronize(h->aux);
    std::fill(h->host_bad.begin(), h->host_bad.end(), (unsigned char)1);
    return rc;
  }
  h->lookaheads += 1;
  *next_presolved = true;
  return EKF_OK;
}

// The input rings (h_ring / d_ring, h_det / d_det): RING slots used in order.  A slot may be refilled by the host once the
// work that read it has run; that is tracked per GROUP of slots -- the event of a group is recorded behind the launch that
// read its last slot and waited for when the ring comes round to its first slot again.
static int ring_take(ekf_handle* h, int* slot) {
  const int s = h->ring_pos, g = s / RING_GROUP;
  h->ring_pos = (s + 1) % RING;
  if (s % RING_GROUP == 0) {
    if (h->ring_open[g]) HIP_TRY(h, hipStreamSynchronize(h->stream));   // (a failed call left the group without its event)
    else if (h->ring_used[g]) HIP_TRY(h, hipEventSynchronize(h->ring_ev[g]));
  }
  h->ring_open[g] = true;
  *slot = s;
  return EKF_OK;
}
static int ring_done(ekf_handle* h, int slot) {
  if (slot % RING_GROUP != RING_GROUP - 1) return EKF_OK;
  const int g = slot / RING_GROUP;
  HIP_TRY(h, hipEventRecord(h->ring_ev[g], h->stream));
  h->ring_used[g] = true;
  h->ring_open[g] = false;
  return EKF_OK;
}

static int do_step(ekf_handle* h, int base_flags, const double* lin, const double* ang, const int* idx,
                   const double* range, const double* bearing, const int* m, int stride, int fetch_b = -1) {
  if (!h) return EKF_ERR_ARG;
  if (int rc = check_host_bad(h, "ekf_step")) return rc;
  if (int rc = refresh_sizes(h)) return rc;
  const bool upd = (base_flags & FLAG_UPDATE) != 0, pred = (base_flags & FLAG_PREDICT) != 0;
  if (pred && (!lin || !ang)) return fail(h, EKF_ERR_ARG, "NULL lin/ang");
  if (upd && (!m || stride < 0)) return fail(h, EKF_ERR_ARG, "NULL m / bad stride");
  int m_hi = 0;
  if (upd && h->cfg.enable_measurement_model)
    for (int b = 0; b < h->batch; ++b) {
      if (m[b] < 0 || m[b] > stride) return fail(h, EKF_ERR_ARG, "m[b] must be in [0, stride]");
      m_hi = std::max(m_hi, m[b]);
    }
  if (m_hi > 0 && (!idx || !range || !bearing)) return fail(h, EKF_ERR_ARG, "NULL observation arrays");
  if (m_hi > 0) {                                      // everything is checked before any handle state changes
    std::vector<unsigned char> seen;
    for (int b = 0; b < h->batch; ++b)
      if (const char* why = validate_obs(h, b, idx + (long)b * stride, m[b], seen)) return fail(h, EKF_ERR_ARG, why);
  }
  HIP_TRY(h, hipSetDevice(h->device));
  if (int rc = push_floor(h, false)) return rc;
  const int passes = std::max(1, (m_hi + MMAX - 1) / MMAX);
  for (int p = 0; p < passes; ++p) {
    int slot;
    if (int rc = ring_take(h, &slot)) return rc;
    StepIn* hs = h->h_ring + (size_t)slot * h->batch;
    StepIn* ds = h->d_ring + (size_t)slot * h->batch;
    int flags = (upd ? FLAG_UPDATE : 0) | ((pred && p == 0) ? FLAG_PREDICT : 0);
    int m_pass_hi = 0;
    for (int b = 0; b < h->batch; ++b) {
      const int mb = (upd && h->cfg.enable_measurement_model) ? m[b] : 0;
      const long off = (long)b * stride;
      fill_step(hs[b], h->n[b], h->neff[b], pred ? lin[b] : 0.0, pred ? ang[b] : 0.0, flags,
                idx ? idx + off : nullptr, range ? range + off : nullptr, bearing ? bearing + off : nullptr, mb, p);
      m_pass_hi = std::max(m_pass_hi, hs[b].m);
      h->neff_enq[b] = h->opt_active_bound ? h->neff[b] : h->n[b];
    }
    h->fetch_b = p == passes - 1 ? fetch_b : -1;
    if (small_path(h) && h->opt_zero_copy_inputs) {
      // small-state path: the one workgroup per trajectory fetches its 352-byte record straight from the pinned ring (one
      // coalesced read over PCIe, ~1.5 us) -- a staged host-to-device copy in front of the kernel costs 5 - 10 us of latency
      // per step, which at these sizes is a third of the step
      if (int rc = enqueue_pass(h, hs, m_pass_hi)) return rc;
      if (int rc = ring_done(h, slot)) return rc;                // (the slot is free once the kernel has run)
      continue;
    }
    HIP_TRY(h, hipMemcpyAsync(ds, hs, sizeof(StepIn) * h->batch, hipMemcpyHostToDevice, h->stream));
    if (int rc = ring_done(h, slot)) return rc;
    if (int rc = enqueue_pass(h, ds, m_pass_hi)) return rc;
  }
  h->fetch_b = -1;
  return EKF_OK;
}

// ---- device-side association ------------------------------------------------------------------
static int assoc_init(ekf_handle* h) {
  if (h->dtagmap) return EKF_OK;
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, hipMalloc(&h->dtagmap, sizeof(int) * TAGMAX * h->batch));
  HIP_TRY(h, hipMemsetAsync(h->dtagmap, 0xFF, sizeof(int) * TAGMAX * h->batch, h->stream));   // -1
  HIP_TRY(h, hipMalloc(&h->dneff, sizeof(int) * h->batch));
  HIP_TRY(h, hipMalloc(&h->d_det, sizeof(DetIn) * h->batch * RING));
  HIP_TRY(h, hipHostMalloc(&h->h_det, sizeof(DetIn) * h->batch * RING, hipHostMallocDefault));
  HIP_TRY(h, hipMalloc(&h->d_assoc_step, sizeof(StepIn) * h->batch * 2));   // two update passes: the first 16 landmarks, the rest
  HIP_TRY(h, hipMalloc(&h->d_assoc_out, sizeof(AssocOut) * h->batch));
  HIP_TRY(h, hipMemsetAsync(h->d_assoc_out, 0, sizeof(AssocOut) * h->batch, h->stream));
  return EKF_OK;
}

extern "C" int ekf_set_association(ekf_handle* h, double gate_range, const int* ignore_tags, int n_ignore) {
  if (!h) return EKF_ERR_ARG;
  if (n_ignore < 0 || n_ignore > IGNMAX || (n_ignore > 0 && !ignore_tags))
    return fail(h, EKF_ERR_ARG, "ekf_set_association: at most 16 ignored tags");
  h->acfg.gate2 = gate_range * gate_range;
  h->acfg.n_ignore = n_ignore;
  for (int i = 0; i < n_ignore; ++i) h->acfg.ignore[i] = ignore_tags[i];
  return EKF_OK;
}

extern "C" int ekf_step_detections(ekf_handle* h, const double* lin, const double* ang, const int* count,
                                   const int* tag_id, const double* pose_t, const double* pose_err, int stride) {
  if (!h) return EKF_ERR_ARG;
  if (!lin || !ang || !count || stride < 0) return fail(h, EKF_ERR_ARG, "ekf_step_detections: NULL array");
  if (int rc = check_host_bad(h, "ekf_step_detections")) return rc;
  if (int rc = assoc_init(h)) return rc;
  HIP_TRY(h, hipSetDevice(h->device));
  if (int rc = push_floor(h, false)) return rc;
  // an upper bound of the landmarks observed this window (distinct tag ids) selects the kernel instantiation
  int m_hi = 0;
  int slot;
  if (int rc = ring_take(h, &slot)) return rc;
  DetIn* hs = h->h_det + (size_t)slot * h->batch;
  DetIn* ds = h->d_det + (size_t)slot * h->batch;
  for (int b = 0; b < h->batch; ++b) {
    const int c = count[b];
    if (c < 0 || c > stride || c > DMAX) return fail(h, EKF_ERR_ARG, "ekf_step_detections: count must be <= min(stride, EKF_DMAX)");
    if (c > 0 && (!tag_id || !pose_t || !pose_err)) return fail(h, EKF_ERR_ARG, "ekf_step_detections: NULL detection arrays");
    DetIn& d = hs[b];
    d.lin = lin[b];
    d.ang = ang[b];
    d.count = c;
    d.pad = 0;
    int distinct = 0;
    for (int i = 0; i < c; ++i) {
      const long e = (long)b * stride + i;
      d.tag_id[i] = tag_id[e];
      d.pose_err[i] = pose_err[e];
      d.pose_t[i][0] = pose_t[3 * e];
      d.pose_t[i][1] = pose_t[3 * e + 1];
      d.pose_t[i][2] = pose_t[3 * e + 2];
      bool seen = false;
      for (int k = 0; k < i; ++k) seen |= (d.tag_id[k] == d.tag_id[i]);
      distinct += seen ? 0 : 1;
    }
    m_hi = std::max(m_hi, std::min(distinct, AMAX));
  }
  if (!h->cfg.enable_measurement_model) m_hi = 0;
  // the host's view of the active bound must be on the device before the first device-side window
  if (!h->sizes_dirty)
    HIP_TRY(h, hipMemcpyAsync(h->dneff, h->neff.data(), sizeof(int) * h->batch, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipMemcpyAsync(ds, hs, sizeof(DetIn) * h->batch, hipMemcpyHostToDevice, h->stream));
  if (int rc = ring_done(h, slot)) return rc;
  const int mcap = cap_for(std::min(m_hi, MMAX));
  if (((h->pending_k + ranks_for(mcap) + 3) & ~3) > KTOT)   // (what the step's kernels will write: see enqueue_pass)
    if (int rc = flush_pending(h)) return rc;
  h->acfg.active_bound = h->opt_active_bound;
  launch_associate(h->stream, ds, h->dtagmap, h->dn, h->dneff, h->dmu2[h->cur], h->dP, h->dV, h->dW, h->d_assoc_step,
                   h->d_assoc_out, h->dflags, h->acfg, h->ld, h->pstride, h->n_max, h->pending_k, h->batch);
  HIP_TRY(h, hipGetLastError());
  h->sizes_dirty = true;
  // m_hi == 0 only when no trajectory has a detection (or the measurement model is off): then, with nothing
  // pending, the O(n) prediction-only kernel applies; any detection selects the generic path, which is also
  // right when the gate leaves nothing (its ranks are zero)
  // (more than EKF_MMAX distinct tags in some trajectory's window: a second pass with the rest -- an update without a
  //  prediction; trajectories that had fewer find m = 0 there)
  if (int rc = enqueue_pass(h, h->d_assoc_step, std::min(m_hi, MMAX))) return rc;
  if (m_hi > MMAX) return enqueue_pass(h, h->d_assoc_step + h->batch, m_hi - MMAX);
  return EKF_OK;
}

extern "C" int ekf_download_tags(ekf_handle* h, int b, int* m, int* idx, int* tag_id, double* xw, double* yw,
                                 double* err, double* range, double* bearing) {
  if (int rc = check_b(h, b, "ekf_download_tags")) return rc;
  if (!h->d_assoc_out) return fail(h, EKF_ERR_STATE, "ekf_download_tags: no device-side association has run");
  HIP_TRY(h, hipSetDevice(h->device));
  if (int rc = check_internal(h, b, "ekf_download_tags")) return rc;
  AssocOut a;
  HIP_TRY(h, hipMemcpyAsync(&a, h->d_assoc_out + b, sizeof(a), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  if (m) *m = a.m;
  for (int i = 0; i < AMAX; ++i) {
    if (idx) idx[i] = a.idx[i];
    if (tag_id) tag_id[i] = a.tag_id[i];
    if (xw) xw[i] = a.xw[i];
    if (yw) yw[i] = a.yw[i];
    if (err) err[i] = a.err[i];
    if (range) range[i] = a.range[i];
    if (bearing) bearing[i] = a.bearing[i];
  }
  return EKF_OK;
}

extern "C" int ekf_download_tag_index(ekf_handle* h, int b, int* tag_of_index, int capacity, int* n_landmarks) {
  if (int rc = check_b(h, b, "ekf_download_tag_index")) return rc;
  if (!n_landmarks) return fail(h, EKF_ERR_ARG, "ekf_download_tag_index: NULL");
  if (int rc = assoc_init(h)) return rc;
  HIP_TRY(h, hipSetDevice(h->device));
  if (int rc = check_internal(h, b, "ekf_download_tag_index")) return rc;
  std::vector<int> tm(TAGMAX);
  HIP_TRY(h, hipMemcpyAsync(tm.data(), h->dtagmap + (size_t)b * TAGMAX, sizeof(int) * TAGMAX, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  int count = 0;
  for (int id = 0; id < TAGMAX; ++id)
    if (tm[id] >= 0) {
      count = std::max(count, tm[id] + 1);
      if (tag_of_index && tm[id] < capacity) tag_of_index[tm[id]] = id;
    }
  *n_landmarks = count;
  return EKF_OK;
}

extern "C" int ekf_upload_tag_index(ekf_handle* h, int b, const int* tag_of_index, int n_landmarks) {
  if (int rc = check_b(h, b, "ekf_upload_tag_index")) return rc;
  if (n_landmarks < 0 || (n_landmarks > 0 && !tag_of_index)) return fail(h, EKF_ERR_ARG, "ekf_upload_tag_index: bad arguments");
  if (int rc = assoc_init(h)) return rc;
  HIP_TRY(h, hipSetDevice(h->device));
  std::vector<int> tm(TAGMAX, -1);
  for (int i = 0; i < n_landmarks; ++i) {
    if (tag_of_index[i] < 0 || tag_of_index[i] >= TAGMAX) return fail(h, EKF_ERR_ARG, "ekf_upload_tag_index: tag id outside [0, 1024)");
    tm[tag_of_index[i]] = i;
  }
  HIP_TRY(h, hipMemcpyAsync(h->dtagmap + (size_t)b * TAGMAX, tm.data(), sizeof(int) * TAGMAX, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return EKF_OK;
}

extern "C" int ekf_predict(ekf_handle* h, const double* lin, const double* ang) {
  return do_step(h, FLAG_PREDICT, lin, ang, nullptr, nullptr, nullptr, nullptr, 0);
}

extern "C" int ekf_update(ekf_handle* h, const int* idx, const double* range, const double* bearing,
                          const int* m, int stride) {
  return do_step(h, FLAG_UPDATE, nullptr, nullptr, idx, range, bearing, m, stride);
}

extern "C" int ekf_step(ekf_handle* h, const double* lin, const double* ang, const int* idx,
                        const double* range, const double* bearing, const int* m, int stride) {
  return do_step(h, FLAG_PREDICT | FLAG_UPDATE, lin, ang, idx, range, bearing, m, stride);
}

// ekf_step + ekf_download_state(b) in one call: what one iteration of the reference's loop is (EKF_pose_estimation returns
// mean and covariance every call, src/replay_no_ros.py:229-237, :482).  On the small-state path the step's own launch
// leaves trajectory b's state in pinned host memory (k_small_stream's host_out): one launch and one synchronisation per call.
extern "C" int ekf_step_fetch(ekf_handle* h, const double* lin, const double* ang, const int* idx, const double* range,
                              const double* bearing, const int* m, int stride, int b, double* mu, double* P, int n) {
  if (int rc = check_b(h, b, "ekf_step_fetch")) return rc;
  if (!mu || !P) return fail(h, EKF_ERR_ARG, "ekf_step_fetch: NULL output array");
  if (n != h->n[b]) return fail(h, EKF_ERR_ARG, "ekf_step_fetch: n does not match the state size");
  h->fetched = false;
  int want = -1;
  if (n <= PACK_SMALL_N && small_path(h) && h->opt_zero_copy_inputs) {
    HIP_TRY(h, hipSetDevice(h->device));
    if (int rc = pack_buffer(h)) return rc;
    want = b;
  }
  const int rc_step = do_step(h, FLAG_PREDICT | FLAG_UPDATE, lin, ang, idx, range, bearing, m, stride, want);
  h->fetch_b = -1;
  if (rc_step) return rc_step;
  if (!h->fetched) return ekf_download_state(h, b, mu, P, n);
  h->fetched = false;
  h->fused_fetches += 1;
  if (h->opt_fetch_spin) {
    // poll the sequence word the kernel releases behind its stores (pinned, coherent); every few microseconds make sure the
    // stream is still busy -- a launch that failed would otherwise be waited for forever
    const unsigned long long* word = reinterpret_cast<const unsigned long long*>(h->h_pack + PACK_WORDS - 1);
    for (long spin = 1;; ++spin) {
      if (__atomic_load_n(word, __ATOMIC_ACQUIRE) == h->fetch_seq) break;
      if ((spin & 4095) == 0) {
        const hipError_t q = hipStreamQuery(h->stream);
        if (q == hipErrorNotReady) continue;
        if (q != hipSuccess) return fail(h, EKF_ERR_HIP, std::string("ekf_step_fetch: ") + hipGetErrorString(q));
        if (__atomic_load_n(word, __ATOMIC_ACQUIRE) == h->fetch_seq) break;
        return fail(h, EKF_ERR_HIP, "ekf_step_fetch: the stream drained without the step's state having been written");
      }
      __builtin_ia32_pause();
    }
    // Integrity: the host has seen the flag word while the launch is still running; that the payload is complete
    // rests on the ordering of the kernel's posted writes (see k_small_stream).  Tripwire: the trailer's copy of the sequence
    // number, written by another wave; with "fetch_verify" also the XOR checksum over the whole payload.  A mismatch falls
    // back to waiting for the launch -- after which every write is visible -- and is counted.
    const unsigned long long* trailer = reinterpret_cast<const unsigned long long*>(h->h_pack) + (size_t)n * n + n + 1;
    bool good = __atomic_load_n(trailer, __ATOMIC_ACQUIRE) == h->fetch_seq;
    if (good && h->opt_fetch_verify) {
      const unsigned long long* w = reinterpret_cast<const unsigned long long*>(h->h_pack);
      unsigned long long x = 0ull;
      for (size_t i = 0; i < (size_t)n * n + n + 1; ++i) x ^= w[i];
      good = x == trailer[1];
    }
    if (!good) {
      h->fetch_retries += 1;
      HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
  } else {
    HIP_TRY(h, hipStreamSynchronize(h->stream));
  }
  if ((unsigned)h->h_pack[(size_t)n * n + n] & EKF_FLAG_INTERNAL) return check_internal(h, b, "ekf_step_fetch");
  std::memcpy(P, h->h_pack, sizeof(double) * (size_t)n * n);
  std::memcpy(mu, h->h_pack + (size_t)n * n, sizeof(double) * n);
  return EKF_OK;
}

extern "C" int ekf_stream_upload(ekf_handle* h, int steps, const double* lin, const double* ang, const int* idx,
                                 const double* range, const double* bearing, const int* m, int stride) {
  if (!h) return EKF_ERR_ARG;
  if (int rc = refresh_sizes(h)) return rc;
  if (steps <= 0) return fail(h, EKF_ERR_ARG, "ekf_stream_upload: steps must be > 0");
  if (!lin || !ang || !m || stride < 0) return fail(h, EKF_ERR_ARG, "ekf_stream_upload: NULL array");
  if (stride > MMAX) return fail(h, EKF_ERR_ARG, "ekf_stream_upload: stride must be <= EKF_MMAX");
  const size_t count = (size_t)steps * h->batch;
  // validate the whole stream before anything of the previous one is replaced
  std::vector<unsigned char> seen;
  for (int k = 0; k < steps; ++k)
    for (int b = 0; b < h->batch; ++b) {
      const size_t e = (size_t)k * h->batch + b;
      const int mb = h->cfg.enable_measurement_model ? m[e] : 0;
      if (m[e] < 0 || m[e] > stride) return fail(h, EKF_ERR_ARG, "ekf_stream_upload: m out of range");
      if (mb > 0 && (!idx || !range || !bearing)) return fail(h, EKF_ERR_ARG, "ekf_stream_upload: NULL observation arrays");
      if (mb > 0)
        if (const char* why = validate_obs(h, b, idx + e * stride, mb, seen)) return fail(h, EKF_ERR_ARG, why);
    }
  std::vector<StepIn> host(count);
  h->stream_mhi.assign(steps, 0);
  h->stream_m.assign(count, 0);
  h->stream_own.assign(count, 3);
  h->stream_maxlm.assign(h->batch, 0);
  h->stream_steps = 0;
  // The records carry the bound that follows from the stream's own observations; what the state already
  // correlates when the stream is RUN (possibly later, possibly more than once) is added there (push_floor).
  std::vector<int> own(h->batch, 3);
  for (int k = 0; k < steps; ++k)
    for (int b = 0; b < h->batch; ++b) {
      const size_t e = (size_t)k * h->batch + b;
      const int mb = h->cfg.enable_measurement_model ? m[e] : 0;
      fill_step(host[e], h->n[b], own[b], lin[e], ang[e], FLAG_PREDICT | FLAG_UPDATE, idx ? idx + e * stride : nullptr,
                range ? range + e * stride : nullptr, bearing ? bearing + e * stride : nullptr, mb, 0);
      h->stream_mhi[k] = std::max(h->stream_mhi[k], mb);
      h->stream_m[e] = (unsigned char)host[e].m;
      h->stream_own[e] = own[b];
      for (int i = 0; i < mb; ++i) h->stream_maxlm[b] = std::max(h->stream_maxlm[b], host[e].idx[i] + 1);
    }
  HIP_TRY(h, hipSetDevice(h->device));
  if (h->stream_cap < count) {
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (h->d_stream) HIP_TRY(h, hipFree(h->d_stream));
    h->d_stream = nullptr;
    h->stream_cap = 0;
    HIP_TRY(h, hipMalloc(&h->d_stream, sizeof(StepIn) * count));
    h->stream_cap = count;
  }
  HIP_TRY(h, hipMemcpyAsync(h->d_stream, host.data(), sizeof(StepIn) * count, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));      // inputs are resident in HBM from here on
  h->stream_steps = steps;
  return EKF_OK;
}

extern "C" int ekf_stream_run(ekf_handle* h, int first, int count) {
  if (!h) return EKF_ERR_ARG;
  if (int rc = check_host_bad(h, "ekf_stream_run")) return rc;
  if (int rc = refresh_sizes(h)) return rc;
  if (first < 0 || count < 0 || first + count > h->stream_steps)
    return fail(h, EKF_ERR_STATE, "ekf_stream_run: range outside the uploaded stream");
  for (int b = 0; b < h->batch; ++b)                   // the state may have been replaced since the upload
    if (h->stream_maxlm[b] > (h->n[b] - 3) / 2)
      return fail(h, EKF_ERR_STATE, "ekf_stream_run: the uploaded stream observes landmarks the current state does not have");
  HIP_TRY(h, hipSetDevice(h->device));
  if (int rc = push_floor(h, true)) return rc;
  if (small_path(h) && count > 0) {
    // the whole range as ONE launch (in pieces of 4096 steps: a bounded kernel), P resident in LDS across all its steps
    for (int k = first; k < first + count; k += 4096)
      if (int rc = enqueue_small(h, h->d_stream + (size_t)k * h->batch, std::min(4096, first + count - k))) return rc;
    for (int b = 0; b < h->batch; ++b)
      h->neff[b] = std::max(h->neff[b], std::min(h->n[b], h->stream_own[(size_t)(first + count - 1) * h->batch + b]));
    h->neff_enq = h->neff;
    return EKF_OK;
  }
  for (int k = first; k < first + count;) {
    // Where nothing is pending the rest of the range runs as packed cadences (ekf_cadence.hip): per cadence one solve launch
    // and one panel launch for a trajectory's next 40 landmark updates and every prediction in between, each trajectory on
    // its own cursor, a covariance pass between two cadences.  While ranks are pending (steps enqueued before this call) the
    // per-step kernels append to them until their pass is due.
    if (cadences_possible(h)) {
      // (planned and uploaded in pieces of at most 8192 steps: 32 B per cadence and trajectory; a piece ends with every
      //  trajectory at its last step, and what its last cadence left pending is flushed so that the next piece starts fused)
      const int piece_end = std::min(first + count, k + 8192);
      plan_cadences(h, k, piece_end, h->run_plan);
      if (int rc = upload_run_plan(h)) return rc;
      // chained solves: decided per piece -- every solve of it then records its cadence's transform; whether a cadence is
      // chained to the next is decided where the pass between them is planned (enqueue_cadence).  (Banks of up to 40, as the
      // look-ahead: every solve and chain workgroup has to find a CU beside the pass.)
      h->chain_run = h->opt_chain && h->opt_lookahead && h->run_plan.ncad >= 2 && h->batch <= 40;
      if (h->chain_run) {
        // ... and only where a pass of this bank can leave CUs to a solve beside it at all (with every state index active and a
        // full cadence pending: the headline's 32 x N = 2000 never does -- its solves stay the plain instantiation)
        const std::vector<int> enq = h->neff_enq;
        const int pk = h->pending_k;
        h->neff_enq = h->n;
        h->pending_k = KTOT;
        h->chain_run = beside_the_pass(h, plan_pass(h));
        h->neff_enq = enq;
        h->pending_k = pk;
      }
      if (h->chain_run) {
        const int n_hi = *std::max_element(h->n.begin(), h->n.end());
        for (int i = 0; i < 2; ++i)
          if (!h->dprow3[i]) HIP_TRY(h, hipMalloc(&h->dprow3[i], sizeof(double) * 3 * (size_t)h->ld * h->batch));
        if (!h->dxg) {
          const size_t gw = sizeof(double) * (size_t)h->batch * 84 * 88;
          HIP_TRY(h, hipMalloc(&h->dxg, gw));
          HIP_TRY(h, hipMalloc(&h->dbg, gw));
          HIP_TRY(h, hipMalloc(&h->dsync, sizeof(unsigned) * chain_sync_words()));
          HIP_TRY(h, hipMemsetAsync(h->dsync, 0, sizeof(unsigned) * chain_sync_words(), h->stream));
          HIP_TRY(h, hipStreamSynchronize(h->stream));   // (the second stream's launches read the counters too)
        }
        // (everything a chained cadence touches exists before its first launch: between the enqueue of a launch that waits on a
        //  device-side counter and the enqueue of the launch that advances it the host must not block -- an allocation may)
        if (!h->dgbuf) HIP_TRY(h, hipMalloc(&h->dgbuf, sizeof(double) * cadence_gbuf_doubles() * h->batch));
        for (int i = 0; i < 2; ++i) {
          if (!h->dcad2[i]) HIP_TRY(h, hipMalloc(&h->dcad2[i], sizeof(CadOut) * h->batch));
          if (!h->dpre[i]) HIP_TRY(h, hipMalloc(&h->dpre[i], sizeof(CadPre) * h->batch));
        }
        if (!h->dgmu) HIP_TRY(h, hipMalloc(&h->dgmu, sizeof(double) * (128 * h->batch + 32)));   // (+ 32 words: the stamps of a -DCHAIN_STAMPS build)
        // the pose rows "before the first cadence": where the previous cadence's panel launch would have left them
        launch_snap_pose(h->stream, h->dP, h->dn, h->ld, h->pstride, h->batch, n_hi, h->dprow3[h->cpar ^ 1]);
        HIP_TRY(h, hipGetLastError());
      }
      bool presolved = false;                          // the next cadence's solve has been enqueued already (chained / look-ahead)
      for (int c = 0; c < h->run_plan.ncad; ++c)
        if (int rc = enqueue_cadence(h, c, presolved, &presolved)) return rc;
      if (int rc = join_aux(h)) return rc;
      k = piece_end;
      if (k < first + count)
        if (int rc = flush_pending(h)) return rc;
      continue;
    }
    for (int b = 0; b < h->batch; ++b)
      h->neff_enq[b] = std::min(h->n[b], std::max(h->floor_host[b], h->stream_own[(size_t)k * h->batch + b]));
    if (int rc = enqueue_pass(h, h->d_stream + (size_t)k * h->batch, h->stream_mhi[k])) return rc;
    ++k;
  }
  if (count > 0)
    for (int b = 0; b < h->batch; ++b)
      h->neff[b] = std::max(h->neff[b], std::min(h->n[b], h->stream_own[(size_t)(first + count - 1) * h->batch + b]));
  // A cadence only forms where nothing is pending (its panel is gathered from P_base alone).  A run's last cadence leaves its
  // ranks pending unless its slots are used up -- right for one long run and for online steps behind it, but a caller that
  // drives the stream in SHORT pieces would alternate between one short cadence and the per-step kernels: "run_end_flush" = 1
  // applies them here (one covariance pass per call) and every piece runs fused.
  if (h->opt_run_end_flush && h->opt_fused_cadence)
    if (int rc = flush_pending(h)) return rc;
  return EKF_OK;
}

extern "C" int ekf_run_stream(ekf_handle* h, int steps, const double* lin, const double* ang, const int* idx,
                              const double* range, const double* bearing, const int* m, int stride) {
  if (!h) return EKF_ERR_ARG;
  if (steps <= 0) return EKF_OK;
  if (int rc = ekf_stream_upload(h, steps, lin, ang, idx, range, bearing, m, stride)) return rc;
  return ekf_stream_run(h, 0, steps);
}

extern "C" int ekf_predict_dense(ekf_handle* h, int b, const double* F, const double* Q) {
  if (int rc = check_b(h, b, "ekf_predict_dense")) return rc;
  if (!F || !Q) return fail(h, EKF_ERR_ARG, "ekf_predict_dense: NULL matrix");
  const int n = h->n[b];
  const size_t bytes = sizeof(double) * (size_t)h->rows * h->ld;
  HIP_TRY(h, hipSetDevice(h->device));
  if (int rc = materialize(h, b)) return rc;           // the product needs the full matrix
  if (!h->dF) {
    HIP_TRY(h, hipMalloc(&h->dF, bytes));
    HIP_TRY(h, hipMalloc(&h->dQ, bytes));
    HIP_TRY(h, hipMalloc(&h->dTmp, bytes));
  }
  HIP_TRY(h, hipMemsetAsync(h->dF, 0, bytes, h->stream));
  HIP_TRY(h, hipMemcpy2DAsync(h->dF, sizeof(double) * h->ld, F, sizeof(double) * n, sizeof(double) * n, n,
                              hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipMemcpy2DAsync(h->dQ, sizeof(double) * h->ld, Q, sizeof(double) * n, sizeof(double) * n, n,
                              hipMemcpyHostToDevice, h->stream));
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (h->profile) {
    if (int rc = prof_event(h, &e0)) return rc;
    if (int rc = prof_event(h, &e1)) return rc;
    HIP_TRY(h, hipEventRecord(e0, h->stream));
  }
  h->neff[b] = n;                                      // a general F correlates everything
  // the product works on plain row-major matrices: a covariance kept in column panels (ld > 4096) goes through a
  // row-major copy (device to device, one 2-D copy per panel each way; 4 n^3 flop dwarf it)
  double* Pdense = h->dP + (size_t)b * h->pstride;
  if (p_panels(h->ld) > 1) {
    if (!h->dPlin) HIP_TRY(h, hipMalloc(&h->dPlin, bytes));
    HIP_TRY(h, copy_cov(h, b, h->dPlin, h->ld, 0, 0, n, n, false, true));
    Pdense = h->dPlin;
  }
  if (dense_propagate(h->stream, Pdense, h->dTmp, h->dF, h->dQ, n, h->ld) != 0)
    return fail(h, EKF_ERR_HIP, "ekf_predict_dense: launch failed");
  if (p_panels(h->ld) > 1) HIP_TRY(h, copy_cov(h, b, h->dPlin, h->ld, 0, 0, n, n, true, true));
  if (h->profile) HIP_TRY(h, hipEventRecord(e1, h->stream));
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return EKF_OK;
}

extern "C" int ekf_flush(ekf_handle* h) {
  if (!h) return EKF_ERR_ARG;
  HIP_TRY(h, hipSetDevice(h->device));
  return flush_pending(h);
}

extern "C" int ekf_sync(ekf_handle* h) {
  if (!h) return EKF_ERR_ARG;
  HIP_TRY(h, hipSetDevice(h->device));
  return check_internal(h, -1, "ekf_sync");          // (synchronises)
}

extern "C" int ekf_status_flags(ekf_handle* h, int b, unsigned* flags) {
  if (int rc = check_b(h, b, "ekf_status_flags")) return rc;
  if (!flags) return fail(h, EKF_ERR_ARG, "ekf_status_flags: NULL");
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, hipMemcpyAsync(flags, h->dflags + b, sizeof(unsigned), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return EKF_OK;
}

extern "C" int ekf_timer_begin(ekf_handle* h) {
  if (!h) return EKF_ERR_ARG;
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, hipEventRecord(h->t0, h->stream));
  return EKF_OK;
}

extern "C" int ekf_timer_end(ekf_handle* h, double* elapsed_ms) {
  if (!h || !elapsed_ms) return EKF_ERR_ARG;
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, hipEventRecord(h->t1, h->stream));
  HIP_TRY(h, hipEventSynchronize(h->t1));
  float ms = 0.f;
  HIP_TRY(h, hipEventElapsedTime(&ms, h->t0, h->t1));
  *elapsed_ms = ms;
  return EKF_OK;
}

extern "C" int ekf_profile_enable(ekf_handle* h, int on) {
  if (!h) return EKF_ERR_ARG;
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  h->profile = on != 0;
  h->prof_used = 0;
  h->prof_seen = 0;
  // (the events a run will use exist before it starts: creating one inside the run costs the host tens of microseconds)
  if (h->profile)
    while (h->prof_pool.size() < (h->opt_profile_kernels ? 1024u : 128u)) {
      hipEvent_t e;
      HIP_TRY(h, hipEventCreate(&e));
      h->prof_pool.push_back(e);
    }
  return EKF_OK;
}

extern "C" int ekf_profile_read(ekf_handle* h, double* pass_ms_total, long long* pass_launches) {
  if (!h || !pass_ms_total || !pass_launches) return EKF_ERR_ARG;
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  if (h->aux) HIP_TRY(h, hipStreamSynchronize(h->aux));
  double total = 0.0;
  long long count = 0;
  for (size_t i = 0; i + 1 < h->prof_used; i += 2) {
    if (h->prof_cls[i / 2] != 0) continue;
    float ms = 0.f;
    HIP_TRY(h, hipEventElapsedTime(&ms, h->prof_pool[i], h->prof_pool[i + 1]));
    total += ms;
    count += 1;
  }
  *pass_ms_total = total;
  *pass_launches = count;
  h->prof_used = 0;
  return EKF_OK;
}

// ("profile_kernels" = 1) the same for the cadence's other launches: cls 1 the solve launch, 2 the chain / look-ahead gather
// launch, 3 the panel launch (0: the covariance pass); does not reset -- read these BEFORE ekf_profile_read
extern "C" int ekf_profile_read_class(ekf_handle* h, int cls, double* ms_total, long long* launches) {
  if (!h || !ms_total || !launches) return EKF_ERR_ARG;
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  if (h->aux) HIP_TRY(h, hipStreamSynchronize(h->aux));
  double total = 0.0;
  long long count = 0;
  for (size_t i = 0; i + 1 < h->prof_used; i += 2) {
    if (h->prof_cls[i / 2] != cls) continue;
    float ms = 0.f;
    HIP_TRY(h, hipEventElapsedTime(&ms, h->prof_pool[i], h->prof_pool[i + 1]));
    total += ms;
    count += 1;
  }
  *ms_total = total;
  *launches = count;
  return EKF_OK;
}

extern "C" long long ekf_profile_passes(ekf_handle* h) { return h ? (long long)h->prof_seen : -1; }

// (diagnostics section of the header: the words behind the queue heads, where a -DRS_STAMPS build of
//  k_flush_rs leaves its time stamps)
extern "C" int ekf_debug_read(ekf_handle* h, void* dst, long bytes) {
  if (!h || !dst) return EKF_ERR_ARG;
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  const long have = (long)sizeof(unsigned) * flush_rs_queue_words();
  HIP_TRY(h, hipMemcpy(dst, h->dqueue, (size_t)std::min(bytes, have), hipMemcpyDeviceToHost));
  return EKF_OK;
}

// (diagnostics section of the header; no device needed) the equal static shares of the row-slab pass for a
// few long trajectories: out = workgroups x 16 pieces x (trajectory, slab, first strip, strips); returns the pieces of
// the longest share (0: no table) -- tests/test_cpu_host.py checks that every strip of every slab is covered exactly once
extern "C" int ekf_debug_pass_shares(int batch, int n_hi, int workgroups, int* out) {
  if (batch < 1 || n_hi < 3 || workgroups < 1 || !out) return -1;
  return build_pass_shares(batch, n_hi, workgroups, out);
}

// (diagnostics section of the header) pieces of the longest share if the last covariance pass ran on equal
// static shares, else 0
extern "C" int ekf_debug_last_pass_shares(ekf_handle* h) { return h ? h->last_shares : -1; }

// (diagnostics section of the header) how many fused cadences ekf_stream_run has launched and how many steps
// they covered: tests assert that the path they mean to check is the one that ran
extern "C" int ekf_debug_cadences(ekf_handle* h, long* cadences, long* steps) {
  if (!h) return EKF_ERR_ARG;
  if (cadences) *cadences = h->cadences;
  if (steps) *steps = h->cadence_traj_steps / std::max(h->batch, 1);
  return EKF_OK;
}
// (diagnostics section of the header) how many of them had their solve run beside the previous covariance pass
extern "C" long ekf_debug_lookaheads(ekf_handle* h) { return h ? h->lookaheads : -1; }
// (diagnostics section of the header) host fallbacks of the device-side association, as the binding reported them
extern "C" long ekf_debug_assoc_fallbacks(ekf_handle* h) { return h ? h->assoc_fallbacks : -1; }
extern "C" void ekf_debug_note_assoc_fallback(ekf_handle* h) { if (h) h->assoc_fallbacks += 1; }
// (diagnostics section of the header) ... and how many of those had their block formed by k_chain_cad (chained solves)
extern "C" long ekf_debug_chained(ekf_handle* h) { return h ? h->chained : -1; }
// ... and covariance passes that formed their W fragments from V and the records ("w_from_v")
extern "C" long ekf_debug_w_from_v(ekf_handle* h) { return h ? h->w_from_v_passes : -1; }
extern "C" int ekf_debug_last_pass_wv(ekf_handle* h) { return h ? h->last_wv : -1; }
// (diagnostics section of the header) launches of the small-state path so far
extern "C" long ekf_debug_small_launches(ekf_handle* h) { return h ? h->small_launches : -1; }
extern "C" long ekf_debug_fused_fetches(ekf_handle* h) { return h ? h->fused_fetches : -1; }
extern "C" long ekf_debug_dense_packs(ekf_handle* h) { return h ? h->dense_packs : -1; }
extern "C" long ekf_debug_fetch_retries(ekf_handle* h) { return h ? h->fetch_retries : -1; }

// (diagnostics section of the header) the fused cadence's record of trajectory b (head + per-landmark records)
extern "C" long ekf_debug_cad(ekf_handle* h, int b, void* dst, long bytes) {
  if (!h || b < 0 || b >= h->batch || !h->dcad2[h->cpar ^ 1]) return -1;
  if (hipSetDevice(h->device) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess) return -1;
  const long have = (long)sizeof(CadOut);
  if (dst && bytes > 0 && hipMemcpy(dst, h->dcad2[h->cpar ^ 1] + b, (size_t)std::min(bytes, have), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  return have;
}

// (diagnostics section of the header) raw device buffers of trajectory b, exactly as they stand -- no flush,
// no mirror, no status check: which = 0 P_base (device layout, ekf_device.h: rows x ld up to ld = 4096, column panels beyond), 1 V (80 x ld), 2 W (80 x ld, MFMA-tiled), 3 the mean buffer
// the NEXT step reads, 4 the other mean buffer.  Returns the number of doubles the buffer holds (copies min(count, that)).
extern "C" long ekf_debug_snapshot(ekf_handle* h, int b, int which, double* dst, long count) {
  if (!h || b < 0 || b >= h->batch || which < 0 || which > 6) return -1;
  if (hipSetDevice(h->device) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess) return -1;
  const double* src = nullptr;
  long have = 0;
  switch (which) {
    case 0: src = h->dP + (size_t)b * h->pstride; have = h->pstride; break;
    case 1: src = h->dV + (size_t)b * KTOT * h->ld; have = (long)KTOT * h->ld; break;
    case 2: src = h->dW + (size_t)b * KTOT * h->ld; have = (long)KTOT * h->ld; break;
    case 3: src = h->dmu2[h->cur] + (size_t)b * h->ld; have = h->ld; break;
    case 6: src = h->dgbuf ? h->dgbuf + (size_t)b * 84 * 88 : nullptr; have = h->dgbuf ? 84L * 88 : 0; break;   // (chained solves: the last chained block of trajectory b, 84 x 88)
    case 5: src = h->dgmu; have = h->dgmu ? 128L * h->batch + 32 : 0; break;   // (chained solves: the means at the positions, all trajectories; then a -DCHAIN_STAMPS build's stamps)
    default: src = h->dmu2[h->cur ^ 1] + (size_t)b * h->ld; have = h->ld; break;
  }
  if (dst && count > 0 &&
      hipMemcpy(dst, src, sizeof(double) * (size_t)std::min(count, have), hipMemcpyDeviceToHost) != hipSuccess)
    return -1;
  return have;
}

// (diagnostics section of the header; no device needed) the units of the row-slab pass's work queues in
// hand-out order for a batch of `batch` trajectories of `nrb` slabs: what tests/test_cpu_host.py checks for coverage
extern "C" int ekf_debug_pass_units(int batch, int nrb, int nch, int mode, int* out, int cap) {
  if (batch < 1 || nrb < 1 || nch < 1 || mode < 0 || mode > 3 || (cap > 0 && !out)) return -1;
  return debug_pass_units(batch, nrb, nch, mode, out, cap);
}

extern "C" int ekf_last_pass(ekf_handle* h, int* kernel, int* k_tiles, int* streaming) {
  if (!h) return EKF_ERR_ARG;
  if (kernel) *kernel = h->last_kernel;
  if (k_tiles) *k_tiles = h->last_nkt;
  if (streaming) *streaming = h->last_streaming;
  return EKF_OK;
}

extern "C" int ekf_set_option(ekf_handle* h, const char* name, int value) {
  if (!h || !name) return EKF_ERR_ARG;
  if (std::strcmp(name, "pass_kernel") == 0) {
    if (value != -1 && value != 0 && value != 2)         // (1 was the producer/consumer form, removed in round 3: never faster)
      return fail(h, EKF_ERR_ARG, "pass_kernel: -1 (auto), 0 (column strips) or 2 (row slabs)");
    h->opt_pass_kernel = value;
    return EKF_OK;
  }
  if (std::strcmp(name, "rank_limit") == 0) {
    if (value < 2 || value > KTOT) return fail(h, EKF_ERR_ARG, "rank_limit out of range");
    h->opt_rank_limit = value;
    return EKF_OK;
  }
  if (std::strcmp(name, "pass_rows_per_block") == 0) {
    if (value < 0 || value > 4096) return fail(h, EKF_ERR_ARG, "pass_rows_per_block out of range");
    h->opt_rows_per_block = value;
    return EKF_OK;
  }
  if (std::strcmp(name, "pass_chunk") == 0) {
    if (value < 0 || value > 4096) return fail(h, EKF_ERR_ARG, "pass_chunk out of range");
    h->opt_pass_chunk = value;
    return EKF_OK;
  }
  if (std::strcmp(name, "small_state") == 0) {
    if (value != 0 && value != 1) return fail(h, EKF_ERR_ARG, "small_state must be 0 or 1");
    HIP_TRY(h, hipSetDevice(h->device));
    if (int rc = flush_pending(h)) return rc;          // (the small-state path runs only with nothing pending)
    h->opt_small_state = value;
    return EKF_OK;
  }
  if (std::strcmp(name, "zero_copy_inputs") == 0) {
    if (value != 0 && value != 1) return fail(h, EKF_ERR_ARG, "zero_copy_inputs must be 0 or 1");
    h->opt_zero_copy_inputs = value;
    return EKF_OK;
  }
  if (!std::strcmp(name, "pack_dense")) {
    if (value < 0 || value > 2) return fail(h, EKF_ERR_ARG, "pack_dense must be 0, 1 or 2");
    h->opt_pack_dense = value;
    return EKF_OK;
  }
  if (!std::strcmp(name, "col_gather")) {
    if (value != 0 && value != 1) return fail(h, EKF_ERR_ARG, "col_gather must be 0 or 1");
    h->opt_col_gather = value;
    return EKF_OK;
  }
  if (!std::strcmp(name, "fetch_verify")) {
    if (value != 0 && value != 1) return fail(h, EKF_ERR_ARG, "fetch_verify must be 0 or 1");
    h->opt_fetch_verify = value;
    return EKF_OK;
  }
  if (!std::strcmp(name, "fetch_spin")) {
    if (value != 0 && value != 1) return fail(h, EKF_ERR_ARG, "fetch_spin must be 0 or 1");
    h->opt_fetch_spin = value;
    return EKF_OK;
  }
  if (std::strcmp(name, "pass_share_order") == 0) {
    if (value != 0 && value != 1) return fail(h, EKF_ERR_ARG, "pass_share_order must be 0 or 1");
    h->opt_share_order = value;                        // (part of the cached table's key: it is rebuilt)
    return EKF_OK;
  }
  if (std::strcmp(name, "fused_step") == 0) {
    // (2 = diagnostic: the solve never publishes its completion, so that every panel workgroup's bounded wait must
    //  time out and raise EKF_FLAG_INTERNAL -- the results of such a step are garbage)
    if (value < 0 || value > 2) return fail(h, EKF_ERR_ARG, "fused_step must be 0, 1 or 2 (diagnostic)");
    h->opt_fused_step = value;
    return EKF_OK;
  }
  if (std::strcmp(name, "lookahead") == 0) {
    if (value != 0 && value != 1) return fail(h, EKF_ERR_ARG, "lookahead must be 0 or 1");
    h->opt_lookahead = value;
    return EKF_OK;
  }
  if (std::strcmp(name, "lookahead_min_mb") == 0) {
    if (value < 0 || value > 100000) return fail(h, EKF_ERR_ARG, "lookahead_min_mb out of range");
    h->opt_lookahead_min_mb = value;
    return EKF_OK;
  }
  if (std::strcmp(name, "beside_min_mb") == 0) {
    if (value < 0 || value > 100000) return fail(h, EKF_ERR_ARG, "beside_min_mb out of range");
    h->opt_beside_min_mb = value;
    return EKF_OK;
  }
  if (std::strcmp(name, "pre_positions") == 0) {
    if (value != 0 && value != 1) return fail(h, EKF_ERR_ARG, "pre_positions must be 0 or 1");
    h->opt_pre_positions = value;
    return EKF_OK;
  }
  if (std::strcmp(name, "panel_own_gate") == 0) {
    if (value != 0 && value != 1) return fail(h, EKF_ERR_ARG, "panel_own_gate must be 0 or 1");
    h->opt_panel_own_gate = value;
    return EKF_OK;
  }
  if (std::strcmp(name, "w_from_v") == 0) {
    if (value != 0 && value != 1) return fail(h, EKF_ERR_ARG, "w_from_v must be 0 or 1");
    h->opt_w_from_v = value;
    return EKF_OK;
  }
  if (std::strcmp(name, "panel_shape") == 0) {
    if (value < 0 || value > 3) return fail(h, EKF_ERR_ARG, "panel_shape must be 0 .. 3");
    h->opt_panel_shape = value;
    return EKF_OK;
  }
  if (std::strcmp(name, "panel_tform") == 0) {
    if (value != 0 && value != 1) return fail(h, EKF_ERR_ARG, "panel_tform must be 0 or 1");
    h->opt_panel_tform = value;
    return EKF_OK;
  }
  if (std::strcmp(name, "run_end_flush") == 0) {
    if (value != 0 && value != 1) return fail(h, EKF_ERR_ARG, "run_end_flush must be 0 or 1");
    h->opt_run_end_flush = value;
    return EKF_OK;
  }
  if (std::strcmp(name, "profile_kernels") == 0) {
    if (value != 0 && value != 1) return fail(h, EKF_ERR_ARG, "profile_kernels must be 0 or 1");
    h->opt_profile_kernels = value;
    return EKF_OK;
  }
  if (std::strcmp(name, "profile_stride") == 0) {
    if (value < 1 || value > 1024) return fail(h, EKF_ERR_ARG, "profile_stride must be in [1, 1024]");
    h->profile_stride = value;
    return EKF_OK;
  }
  if (std::strcmp(name, "chain") == 0) {
    if (value != 0 && value != 1) return fail(h, EKF_ERR_ARG, "chain must be 0 or 1");
    h->opt_chain = value;
    return EKF_OK;
  }
  if (std::strcmp(name, "fused_cadence") == 0) {
    if (value != 0 && value != 1) return fail(h, EKF_ERR_ARG, "fused_cadence must be 0 or 1");
    h->opt_fused_cadence = value;
    return EKF_OK;
  }
  if (std::strcmp(name, "pass_workgroups") == 0) {
    if (value < 0 || value > 4096) return fail(h, EKF_ERR_ARG, "pass_workgroups out of range");
    h->opt_pass_workgroups = value;
    return EKF_OK;
  }
  if (std::strcmp(name, "active_bound") == 0) {
    if (value != 0 && value != 1) return fail(h, EKF_ERR_ARG, "active_bound must be 0 or 1");
    HIP_TRY(h, hipSetDevice(h->device));
    if (int rc = flush_pending(h)) return rc;
    h->opt_active_bound = value;
    return EKF_OK;
  }
  if (std::strcmp(name, "flush_every") == 0) {
    if (value < 0 || value > 64) return fail(h, EKF_ERR_ARG, "flush_every must be in [0, 64] (0 = auto)");
    h->opt_flush_every = value;
    return EKF_OK;
  }
  if (std::strcmp(name, "pass_streaming") == 0) {
    if (value < -1 || value > 1) return fail(h, EKF_ERR_ARG, "pass_streaming must be -1 (auto), 0 or 1");
    h->opt_streaming = value;
    return EKF_OK;
  }
  return fail(h, EKF_ERR_ARG, std::string("unknown option ") + name);
}

/*
 * ekfslam_hip.h -- C ABI of libekfslam_hip.so: the MI355X (gfx950) EKF-SLAM predict/update core.
 *
 * The reference (AHHHZ975/SLAM-Duckietown) has no FFI layer: its hot path is one Python
 * function, EKF_pose_estimation (src/replay_no_ros.py:269-482), and src/ekf_bindings.py is an
 * empty placeholder for "EKF bindings".  This header is what that placeholder would bind with
 * ctypes.  Each entry point names the reference lines it replaces.  Plain pointers and sizes
 * only; every function returns an int status (0 = EKF_OK) unless stated otherwise.
 *
 * State layout at this boundary is the reference's own (src/replay_no_ros.py:69-70, :341-360):
 *   mu : (n,)   float64   [x, y, theta, l0x, l0y, ..., l(N-1)x, l(N-1)y],  n = 3 + 2N
 *   P  : (n,n)  float64   row-major (NumPy C order)
 * On the device a covariance is kept as its upper triangle, row-major with a padded row stride up to n_max = 4096 and in
 * column panels of 4096 doubles beyond (every row segment of a tile 32 KB from the next, whatever the size of the state);
 * uploads and downloads translate -- the layout never shows at this boundary.
 * Device buffers are owned by the handle; host arrays are borrowed for the duration of a call.
 * A handle is not thread-safe: one handle per device per host thread.
 */
#ifndef EKFSLAM_HIP_H
#define EKFSLAM_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EKF_OK 0
#define EKF_ERR_ARG (-1)      /* bad argument (message in ekf_last_error) */
#define EKF_ERR_HIP (-2)      /* a HIP runtime call failed, or no gfx950 device */
#define EKF_ERR_STATE (-3)    /* call not valid in the current state */

#define EKF_MMAX 16           /* max landmarks per single device update pass (longer lists are split) */
#define EKF_N_MAX_LIMIT 21823 /* largest n_max = 3 + 2N ekf_create accepts (N = 10910): the kernels address one
                                 covariance with unsigned 32-bit byte offsets, so what is allocated for it -- rows
                                 (n_max rounded up to 64) x column panels of 4096 doubles, 8 bytes each -- must
                                 stay below 4 GiB; larger values fail with EKF_ERR_ARG */

/* sticky per-trajectory flags, ekf_status_flags() */
#define EKF_FLAG_NONFINITE 1u /* a non-finite mean entry was produced (q = 0 at :466-469, singular S at :473) */
#define EKF_FLAG_INTERNAL 4u  /* a bounded wait inside a single-launch step timed out (the GPU did not run the solve
                                 workgroup of the launch beside its panel workgroups for tens of ms).  A timed-out
                                 wave writes nothing while other waves of the same step may have written: the
                                 trajectory's state is UNDEFINED from there on, and ekf_sync and every ekf_download_*
                                 (state, mean, block, tags, tag index) return EKF_ERR_STATE while the flag is set.  The
                                 same status is returned, for every trajectory of the handle, after an enqueueing call
                                 failed half way.  Recovery: upload the trajectory again (ekf_upload_state* clears the
                                 condition); ekf_set_option("fused_step", 0) selects the two-launch step, which has no wait. */
#define EKF_FLAG_ASSOC 2u     /* device-side association dropped a detection: tag id outside [0, 1024), state full,
                                 or more than EKF_AMAX distinct tags in one window */
#define EKF_DMAX 256          /* detections per window for ekf_step_detections (the reference's normal mode: a 0.7 s window
                                 of every camera frame, src/replay_no_ros.py:17 -- 21 frames x a dozen tags) */
#define EKF_AMAX 32           /* distinct tags per window the device-side association takes (two update passes of EKF_MMAX) */

typedef struct ekf_handle ekf_handle;

/* Module constants of src/replay_no_ros.py:15-31 and the literals at :289, :356, :376. */
typedef struct ekf_config {
  double motion_sigma;                /* MOTION_MODEL_VARIANCE       (:15)  default 0.1  */
  double meas_sigma;                  /* MEASUREMENT_MODEL_VARIANCE  (:16)  default 0.7  */
  double arc_threshold;               /* |ang| <= thr -> straight branch (:376) default 1e-2 */
  double landmark_init_var;           /* new-landmark variance (:356-357)   default 1e4  */
  int enable_measurement_model;       /* (:18)  default 1 */
  int enable_circular_interpolation;  /* (:19)  default 1 */
  int disable_motion_model;           /* (:28)  default 0 */
  int reserved;
} ekf_config;

int ekf_config_default(ekf_config *cfg);

/* Number of HIP devices visible to the process (0, and EKF_OK, when there is none). */
int ekf_device_count(int *count);

/* Create a filter bank of `batch` independent trajectories on HIP device `device`, each with room
 * for n_max = 3 + 2*N_max states.  All trajectories start as the reference does
 * (src/replay_no_ros.py:69-70): n = 3, mu = 0, P = motion_sigma * I3.  Fails (EKF_ERR_HIP) when no
 * gfx950 device is present: there is no CPU fallback. */
int ekf_create(int device, int n_max, int batch, const ekf_config *cfg, ekf_handle **out);
int ekf_destroy(ekf_handle *h);

/* Whole state in / out for trajectory b (checkpoint, parity checks).  Blocking.
 * A covariance is symmetric: the device keeps its upper triangle only, so of an uploaded P the upper
 * triangle is authoritative (entries below the diagonal are not even sent: 56 % of the matrix crosses PCIe
 * at n = 4003), and a download returns that triangle mirrored (the reference's own Sigma is symmetric to
 * rounding). */
int ekf_upload_state(ekf_handle *h, int b, const double *mu, const double *P, int n);
int ekf_upload_state_diag(ekf_handle *h, int b, const double *mu, const double *diagP, int n);
int ekf_download_state(ekf_handle *h, int b, double *mu, double *P, int n);
int ekf_download_mean(ekf_handle *h, int b, double *mu, int n);
/* rows [r0, r0+rows) x cols [c0, c0+cols) of the covariance into out (row-major rows x cols), e.g. the 3x3
 * pose block for consistency statistics without shipping n^2 doubles.  Flushes the pending update. */
int ekf_download_block(ekf_handle *h, int b, int r0, int c0, int rows, int cols, double *out);
int ekf_state_size(ekf_handle *h, int b, int *n);

/* State augmentation, src/replay_no_ros.py:341-360: append k landmarks (indices must continue the
 * current count), mean = xy[2*i..], variance = landmark_init_var, zero cross terms. */
int ekf_add_landmarks(ekf_handle *h, int b, int first_index, const double *xy, int k);

/* Pinned (page-locked, device-visible) host memory for the arrays a binding hands to its caller.  The reference's loop
 * gets a fresh n x n covariance back from every call (src/replay_no_ros.py:229-237, :482): in freshly allocated pageable
 * memory a 128 MB download first faults in and pins 32 768 pages (5 ms on top of 2.3 ms of PCIe time at N = 2000); the
 * Python binding recycles buffers from these instead.  ekf_host_alloc returns NULL when the allocation fails. */
void *ekf_host_alloc(size_t bytes);
void ekf_host_free(void *p);

/* predict(): src/replay_no_ros.py:368-430 for every trajectory (lin[b], ang[b]).  O(n) work. */
int ekf_predict(ekf_handle *h, const double *lin, const double *ang);

/* update(): src/replay_no_ros.py:436-480, sequential over idx[b*stride + 0..m[b]) in that order. */
int ekf_update(ekf_handle *h, const int *idx, const double *range, const double *bearing,
               const int *m, int stride);

/* step() = predict + update fused into one pass over P (what EKF_pose_estimation does after
 * association/augmentation, src/replay_no_ros.py:363-482).  Asynchronous on the handle's stream. */
int ekf_step(ekf_handle *h, const double *lin, const double *ang, const int *idx,
             const double *range, const double *bearing, const int *m, int stride);

/* ekf_step followed by ekf_download_state(h, b, mu, P, n), as ONE call: one iteration of the reference's loop
 * (`mean, covariance, tags = EKF_pose_estimation(...)`, src/replay_no_ros.py:229-237; the return at :482 hands back the
 * whole state every call).  Same results as the two calls.  On the small-state path (option "small_state") the step's
 * own launch writes trajectory b's mean, mirrored covariance and flags into pinned host memory: one kernel launch and one
 * synchronisation per call; otherwise the two calls are made for the caller.  Blocking; n must equal the state size. */
int ekf_step_fetch(ekf_handle *h, const double *lin, const double *ang, const int *idx, const double *range,
                   const double *bearing, const int *m, int stride, int b, double *mu, double *P, int n);

/* Device-side front end (association, 1.5 m gate, per-tag averaging, augmentation: src/replay_no_ros.py:280-360)
 * followed by the fused step: one window of raw AprilTag detections per trajectory, frames concatenated in
 * order -- count[b] detections at tag_id / pose_t (x, y, z of tag.pose_t) / pose_err [b*stride + i].  The tag-id
 * -> landmark-index table lives on the device (ekf_download_tag_index); ekf_download_tags returns what the
 * reference returns as tags_positions for the last window.  ekf_set_association sets gate and IGNORE_TAGS. */
int ekf_set_association(ekf_handle *h, double gate_range, const int *ignore_tags, int n_ignore);
int ekf_step_detections(ekf_handle *h, const double *lin, const double *ang, const int *count, const int *tag_id,
                        const double *pose_t, const double *pose_err, int stride);
int ekf_download_tags(ekf_handle *h, int b, int *m, int *idx, int *tag_id, double *xw, double *yw, double *err,
                      double *range, double *bearing);     /* arrays of EKF_AMAX entries */
int ekf_download_tag_index(ekf_handle *h, int b, int *tag_of_index, int capacity, int *n_landmarks);
int ekf_upload_tag_index(ekf_handle *h, int b, const int *tag_of_index, int n_landmarks);

/* Streams of pre-uploaded inputs ([step][batch] and [step][batch][stride] arrays, stride <= EKF_MMAX; m[step][batch] landmarks
 * each -- any count per step and trajectory, 0 included: what the windows of src/replay_no_ros.py:280-301 hold).
 * ekf_stream_upload copies and validates `steps` steps of inputs into HBM (blocking);
 * ekf_stream_run enqueues steps [first, first+count) back to back (asynchronous).  Where nothing is pending they run as packed
 * cadences ("fused_cadence"): per covariance pass every trajectory's next 40 landmark updates, whatever steps they belong to --
 * inside the call the trajectories of a bank advance through the range at their own pace (they are independent filters); when
 * the call returns every one of them has been enqueued up to first+count;
 * ekf_run_stream = upload + run all.
 * Driving a stream in SHORT pieces: a cadence only forms where nothing is pending, and a call's last cadence leaves its ranks
 * pending unless its 40 slots are used up (right for one long run, and for online steps behind it) -- the next call then runs the
 * per-step kernels until the pass is due, so the "40 landmark updates per pass" only hold for long calls.  With
 * ekf_set_option("run_end_flush", 1) every call ends with the covariance pass of what it left pending and every piece runs
 * fused (one pass per call: worth it from ~40 landmark updates per call).  The plan of a call's cadences is made and uploaded
 * per call (32 B per cadence and trajectory). */
int ekf_stream_upload(ekf_handle *h, int steps, const double *lin, const double *ang, const int *idx,
                      const double *range, const double *bearing, const int *m, int stride);
int ekf_stream_run(ekf_handle *h, int first, int count);
int ekf_run_stream(ekf_handle *h, int steps, const double *lin, const double *ang, const int *idx,
                   const double *range, const double *bearing, const int *m, int stride);

/* General dense propagation P <- F P F^T + Q (F, Q row-major n x n) on the fp64 MFMA path: the
 * reference's literal G_F @ P @ G_F.T + F.T @ R @ F product (src/replay_no_ros.py:430) for an
 * arbitrary Jacobian. */
int ekf_predict_dense(ekf_handle *h, int b, const double *F, const double *Q);

/* The covariance is held as P_base + (pending low-rank update of the last few steps, 2 ranks per observed
 * landmark -- exactly: a step is charged the ranks of its busiest trajectory, whatever the landmark count); the O(n^2) pass over
 * P_base is paid once per `flush_every` steps (option; default 0 = as many landmark updates as fit `rank_limit` = 80 pending
 * ranks: 40 -- 5 steps at 8 observations per step, 8 at five, 40 at one).  ekf_flush
 * applies what is pending now (asynchronous).  Every call that reads or rewrites the covariance flushes by
 * itself. */
int ekf_flush(ekf_handle *h);
/* Wait for everything enqueued.  EKF_ERR_STATE if any trajectory carries EKF_FLAG_INTERNAL (ekf_last_error names it). */
int ekf_sync(ekf_handle *h);
/* The sticky flags of trajectory b (always EKF_OK for a valid b: this is how a failed trajectory is identified). */
int ekf_status_flags(ekf_handle *h, int b, unsigned *flags);

/* Message of the last failure on this handle (h == NULL: last failure of ekf_create). */
const char *ekf_last_error(ekf_handle *h);

/* --- measurement hooks (HIP events on the handle's own stream) --- */
int ekf_timer_begin(ekf_handle *h);
int ekf_timer_end(ekf_handle *h, double *elapsed_ms);          /* synchronises */
/* When enabled every launch of the covariance pass (flush) kernel is bracketed by an event pair -- every k-th one with
 * ekf_set_option("profile_stride", k): an event record costs its stream ~6 us, which a single trajectory's 80 us cadence
 * feels.  ekf_profile_read: total time and number of the BRACKETED launches (and resets); ekf_profile_passes: all launches of
 * the pass since profiling was enabled, bracketed or not. */
int ekf_profile_enable(ekf_handle *h, int on);
int ekf_profile_read(ekf_handle *h, double *pass_ms_total, long long *pass_launches); /* and resets */
long long ekf_profile_passes(ekf_handle *h);
/* With ekf_set_option("profile_kernels", 1) (a diagnostic run: every record costs its stream ~6 us) the other launches of a
 * fused cadence carry event pairs too: cls 1 the solve launch, 2 the chain (or look-ahead gather) launch, 3 the panel launch,
 * 0 the pass.  Does not reset: read before ekf_profile_read. */
int ekf_profile_read_class(ekf_handle *h, int cls, double *ms_total, long long *launches);
/* Tuning knobs: "flush_every" (steps per covariance pass, 0 = auto), "rank_limit" (auto cadence: pending
 * ranks that trigger the pass, 2..80), "pass_rows_per_block", "pass_streaming" (-1 auto / 0 resident /
 * 1 nontemporal), "active_bound" (0 = treat every state index as correlated), "pass_kernel" (-1 = auto:
 * the row-slab kernel where the launch streams through HBM and gives every CU work, else k_flush; 0 = k_flush, the
 * column-strip form, 2 = k_flush_rs, the row-slab form; both give the same result bit for bit), "pass_chunk" (row-slab pass: strips per work unit, 0 = auto), "pass_workgroups" (row-slab pass:
 * persistent workgroups, 0 = one per CU; fewer leaves whole CUs to other streams), "fused_step" (1 = small launches
 * run a step as one kernel, the panels gathered beside the solve -- same results; 0 = always two kernels; 2 =
 * diagnostic: the solve never publishes its completion, every bounded wait times out with EKF_FLAG_INTERNAL),
 * "fused_cadence" (1 = ekf_stream_run replays everything between two covariance passes -- a trajectory's next 40 landmark
 * updates, src/replay_no_ros.py:368-480 for each, and every prediction in between -- with one solve launch and one panel launch;
 * "col_gather" = 1: the mirrored column entries of that panel launch are fetched beside the solve by the CUs its chain leaves idle,
 * 0: the panel launch gathers everything itself, same results bit for bit; the fused path is equal to the per-step kernels to
 * rounding (1e-10 relative guaranteed, 1e-13 .. 1e-12 measured; the tests assert 1e-11), not bit for bit; 0 = one step at a time), "lookahead" (1 = where the pass is a
 * small launch, the next cadence's solve runs beside it on the handle's second stream; 0 = strictly in sequence),
 * "pass_share_order" (row-slab pass on static shares: 1 = shares dealt to the XCDs by starting column, 0 = as cut; same
 * result bit for bit), "small_state" (1 = a handle with n_max <= 79 -- up to 38 landmarks; n_max <= 131, 64 landmarks, for a bank of at least 128
 * trajectories -- runs every step, or a whole
 * uploaded stream, as ONE workgroup per trajectory with the covariance resident in LDS: nothing is ever pending; 0 = the
 * general kernels; the default can be set for new handles with the environment variable EKFSLAM_HIP_SMALL_STATE),
 * "zero_copy_inputs" (1 = the small-state kernel reads an online step's record straight from the pinned input ring, 0 = a
 * staged copy first), "fetch_spin" (1 = ekf_step_fetch on the small-state path polls the sequence word its launch releases
 * behind the state it wrote to pinned memory, 0 = it waits for the stream; same results.  The polled hand-over assumes that
 * the kernel's posted writes to coherent pinned memory become visible in fence -> release order -- validated on MI355X;
 * every hand-over carries a second copy of its sequence number written by another wave, compared before the data is
 * trusted, and "fetch_verify" = 1 (the default since round 6; ~1 us per call) also compares an XOR checksum of the whole payload -- a platform that reorders posted writes could deliver the trailer before other payload lines --; a mismatch waits for the stream
 * instead and is counted, ekf_debug_fetch_retries), "pack_dense" (downloads of a whole
 * state into PINNED host memory, e.g. from ekf_host_alloc: 1 = up to 40 MB a kernel mirrors the stored triangle straight into
 * the destination, no mirror pass and no copy engine; 2 = at every size; 0 = never: mirror pass + rectangle copy; same bytes),
 * "panel_shape" (diagnostics: 0 = a fused cadence's panel launch takes the shape its size selects, 1 .. 3 force the row-split
 * latency form / one wave per workgroup / four waves per workgroup; the same result bit for bit),
 * "w_from_v" (1, default: where a fused cadence's covariance pass follows its panel launch at once in the row-slab form, the
 * panel launch writes V only and the pass forms its W fragments from V and the cadence's records -- W = -(V S^-1) per
 * landmark is half of what that launch would write; the same result bit for bit; ekf_debug_snapshot's W view is then stale);
 * unknown names fail.
 * "fused_cadence", "lookahead" and "chain" (1 = where the next cadence's solve runs beside this one's pass -- banks of up to 40
 * trajectories whose pass leaves CUs free: every size with "chain" = 1, from "lookahead_min_mb" = 48 MB of covariance with the
 * round-3 look-ahead --, its block is formed from this cadence's records and the solves of a run follow one another on the
 * handle's stream, panel launch and pass of every cadence on the second one; 0 = the round-3 look-ahead: the block gathered behind
 * the panel launch; sub-options of the chained order: "panel_tform" (1) (small panel launches as a triangular solve on
 * the matrix cores instead of the replay of the landmarks one after the other), "panel_own_gate" (default 0; 1 = small panel launches wait for
 * their solve themselves instead of behind a one-lane gate launch), "pre_positions" (1: a cadence's inputs are formed one cadence
 * ahead), "beside_min_mb" (0: chain at every size)) change the ORDER in which a step's pending ranks are summed (and whether the look-ahead
 * applies depends on the device's CU count and the size of the launch): results are equal to rounding across these
 * settings and across devices, bit-identical only for a fixed setting on a fixed device type. */
int ekf_set_option(ekf_handle *h, const char *name, int value);
/* Which form of the covariance pass the last launch used (-1 = none yet; values as for "pass_kernel"), how many
 * MFMA k-tiles (4 pending ranks each) it applied, and whether it took the nontemporal (streaming) path. */
int ekf_last_pass(ekf_handle *h, int *kernel, int *k_tiles, int *streaming);

/* --- diagnostics (development hooks: counters and raw views the tests use to assert WHICH path ran; no reference
 * interface stands behind them, they change nothing, and a production caller never needs them) --- */
/* Fused cadences ekf_stream_run has launched so far and the steps (of the uploaded stream) they covered. */
int ekf_debug_cadences(ekf_handle *h, long *cadences, long *steps);
/* Host fallbacks of the device-side association: a binding whose window does not fit the device front end's limits takes the
 * host association instead and says so with ekf_debug_note_assoc_fallback; ekf_debug_assoc_fallbacks returns the count. */
long ekf_debug_assoc_fallbacks(ekf_handle *h);
void ekf_debug_note_assoc_fallback(ekf_handle *h);
/* Cadences whose solve ran beside the previous covariance pass (chained solves or look-ahead), and how many of those had
 * their block formed from the previous cadence's records (chained solves, option "chain"). */
long ekf_debug_lookaheads(ekf_handle *h);
long ekf_debug_chained(ekf_handle *h);
/* Covariance passes that formed their W fragments from V and the cadence's records (option "w_from_v"). */
long ekf_debug_w_from_v(ekf_handle *h);
/* ... and whether the last pass was one of them (the fourth template argument of k_flush_rs in a profiler's listing). */
int ekf_debug_last_pass_wv(ekf_handle *h);
/* Pieces of the longest static share the last row-slab pass used (0 = work queues / column strips; -1 = NULL handle). */
int ekf_debug_last_pass_shares(ekf_handle *h);
/* Launches of the small-state path; ekf_step_fetch calls served by the step's own launch; whole-state downloads written
 * by k_pack_dense. */
long ekf_debug_small_launches(ekf_handle *h);
long ekf_debug_fused_fetches(ekf_handle *h);
long ekf_debug_dense_packs(ekf_handle *h);
/* ekf_step_fetch hand-overs whose integrity trailer did not match what the host polled (answered after a stream
 * synchronisation instead): 0 on a platform where the ordering assumption of the polled hand-over holds. */
long ekf_debug_fetch_retries(ekf_handle *h);
/* Raw device views behind a stream synchronisation, no flush: the fused cadence's record of trajectory b (returns its
 * size; copies min(bytes, size)); `which` = 0 P_base (allocated doubles), 1 V, 2 W, 3 the mean buffer the next step reads,
 * 4 the other mean buffer, 5 (chained runs, every trajectory: b is ignored) the mean at the positions of the last chained
 * cadence, 128 doubles per trajectory (dst == NULL: the count); the words behind the row-slab pass's queue heads
 * (where a -DRS_STAMPS build leaves its time stamps). */
long ekf_debug_cad(ekf_handle *h, int b, void *dst, long bytes);
long ekf_debug_snapshot(ekf_handle *h, int b, int which, double *dst, long count);
int ekf_debug_read(ekf_handle *h, void *dst, long bytes);
/* The planning arithmetic of the row-slab pass, host side only (no handle, no device): every unit of the eight per-XCD
 * work queues in hand-out order (returns their number), and the static shares (workgroups x 16 pieces x 4 ints; returns the
 * pieces of the longest share, 0 if one would need more than 16). */
int ekf_debug_pass_units(int batch, int nrb, int nch, int mode, int *out, int cap);
int ekf_debug_pass_shares(int batch, int n_hi, int workgroups, int *out);

#ifdef __cplusplus
}
#endif
#endif

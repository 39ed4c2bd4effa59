#!/usr/bin/env python3
"""EKF-SLAM step throughput on MI355X (BASELINE.json metric: EKF update steps/sec at N landmarks).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (config.workload): N = 2000 landmarks (n = 4003), m = 8 observations per step, B independent
trajectories per GPU (default 32 = BASELINE config 4's per-GPU shard: 256 trajectories over 8 GPUs).
Trajectories are sharded over ranks with no data-path collective (weak scaling: per-GPU work is
fixed).  A bench "step" advances every trajectory of the rank by one EKF step (predict + sequential
update); `value` = trajectory-steps per second over all ranks.  Inputs (the whole odometry+landmark
stream) are resident in HBM before the timed region.  torch is used only for the cross-rank
barrier / max-reduce (gloo); the product path is NumPy + ctypes -> HIP.

The JSON line also carries
  roofline      the covariance pass kernel: algorithmic bytes per launch (B * 8 n (n + 1): one read + one
                write of the stored upper triangle) / average launch duration from HIP event pairs on the
                kernel's own stream around every launch of the pass INSIDE the timed region
  cpu_baseline  the oracle's reference-shaped dense NumPy step timed on this box's host cores (rank 0,
                N = 1 only, a bounded sample) at N = 2000, with N = 500, N = 20 (500 steps) and N = 12 under
                `by_config`
  single_trajectory  BASELINE config 3 (B = 1) steps/s over >= 200 steps; config1 / config2: N = 20 x 1 (500 steps; plus banks
                of 256 and 768 such filters on the small-state path), N = 500 x 1
  online_step   the host-driven call surface: one `EkfSlam.step` per call (indices not known in advance,
                no fused cadence), N = 2000 x 32 and x 1
  drop_in       `EKF_pose_estimation` per call INCLUDING its 8 n^2-byte download (the reference's loop,
                src/replay_no_ros.py:229-237) and, beyond 131 x 131, the upload of the upper triangle: N = 12 (the
                reference's real map, :26), 20, 500, 2000: ms per call
  steady_state  the headline workload timed behind a full sweep of the landmarks (dense covariance: fp64 MFMA power, and the
                clock the part holds, depend on the operands -- ~9 % slower than the young filter the contract times);
                `roofline.frac_steady_state` / `avg_launch_ms_steady_state` / `value_steady_state` repeat it beside the headline
  obs_5, obs_12, variable_m, constant_m4   the shapes the reference's loop produces (src/replay_no_ros.py:280-301, :436): landmark
                counts that are no power of two, and m ~ uniform{0..8} per trajectory and step at scattered indices (beside a
                constant m = 4, the same mean): steps/s, landmark updates/s, passes per step, fused fraction
  sclk_mhz      shader clock sampled during the headline's timed region where that is at least 50 ms (else null), and
                `sclk_mhz_steady_state` over the steady_state leg
  single_trajectory.cadence_us   where a single trajectory's cadence goes: period, the launches' own durations, bubbles
  `--leg NAME` runs one secondary leg alone and prints it.
  (rank 0, N = 1 only, except sclk_mhz and rank_dt_ms).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
MFMA_F64_SPEC_TF = 78.6        # public datasheet figure (the local guide lists no fp64 matrix peak)
MFMA_F64_MEASURED_TF = 77.9    # sustained v_mfma_f64_16x16x4 on this part (accumulators in VGPRs): profiles/mfma_probe.txt


def make_streams(sd_syn, traj_ids, n_landmarks, steps, m, variable=None):
    """`variable` = (m_lo, m_hi): per trajectory and step m ~ uniform{m_lo..m_hi} scattered landmarks (synthetic.variable_stream)
    instead of m consecutive ones; -> ..., mm[steps, batch] (None for a constant m)."""
    made = {}                          # (a bank may repeat trajectory ids: each stream is generated once)
    for t in traj_ids:
        if t not in made:
            made[t] = (sd_syn.variable_stream(n_landmarks, steps, variable[0], variable[1], t) if variable
                       else sd_syn.synthetic_stream(n_landmarks, steps, m, t))
    streams = [made[t] for t in traj_ids]
    lin = np.stack([s[2] for s in streams], axis=1)
    ang = np.stack([s[3] for s in streams], axis=1)
    idx = np.stack([s[4] for s in streams], axis=1)
    zr = np.stack([s[5] for s in streams], axis=1)
    zb = np.stack([s[6] for s in streams], axis=1)
    mm = np.stack([s[7] for s in streams], axis=1) if variable else None
    return streams, lin, ang, idx, zr, zb, mm


def time_filter(sd, sd_syn, shard, grp, device, traj_ids, n_landmarks, m, steps, warmup, profile_leg=True, options=(), clock=None,
                variable=None):
    """Returns (max-over-ranks seconds for `steps` steps, pass_ms_total, pass_launches, device_ms).
    The timed region is sharding.timed_region (device sync + barrier on both sides, max over ranks): the function
    the world_size-2 gloo test covers."""
    n = 3 + 2 * n_landmarks
    total = warmup + steps
    streams, lin, ang, idx, zr, zb, mm = make_streams(sd_syn, traj_ids, n_landmarks, total, m, variable)
    f = sd.EkfSlam(n, batch=len(traj_ids), device=device)
    for opt in options:
        name, value = opt.split("=")
        f.set_option(name, int(value))
    for b, s in enumerate(streams):
        f.set_state_diag(s[0], s[1], b)
    f.stream_upload(lin, ang, idx, zr, zb, mm)
    f.stream_run(0, warmup)
    f.flush()
    cad0 = f.cadence_counters()
    dev = {}
    # launches of the covariance pass inside the timed region are bracketed by HIP event pairs on the stream the pass runs
    # on: every 4th one where the region holds at least 16 (every 2nd from 4 on) -- a record costs its stream ~6 us, which a
    # single trajectory's 80 us cadence feels (round 5 bracketed every launch: 1.4 % of the headline, 12 % of config 3)
    expect = (steps * (m if variable is None else (variable[0] + variable[1]) / 2.0)) / 40.0
    stride = 4 if expect >= 16 else (2 if expect >= 4 else 1)
    f.set_option("profile_stride", stride)
    f.profile_enable(profile_leg)

    def run():
        f.timer_begin()
        f.stream_run(warmup, steps)
        f.flush()                     # any covariance pass still pending belongs to the timed steps
        dev["ms"] = f.timer_end()     # HIP events on the handle's stream (synchronises it)

    if clock is not None:
        clock.start()                                  # the shader clock, sampled while the timed region runs
    dt = shard.timed_region(grp, run, f.sync, time.perf_counter)
    if clock is not None:
        clock.stop()
    dev_ms = dev["ms"]
    pass_ms, launches = 0.0, 0
    time_filter.last_timed = 0
    if profile_leg:
        # (callers divide by `launches`: the timed launches' total is scaled to all of them; `last_timed` says how many were timed)
        pass_ms, timed = f.profile_read()
        launches = f.profile_passes()
        pass_ms = pass_ms / max(timed, 1) * launches
        time_filter.last_timed = timed
        f.profile_enable(False)
    time_filter.last_pass_kernel = f.last_pass()
    cad1 = f.cadence_counters()
    try:
        time_filter.last_chained = int(sd.load_library().ekf_debug_chained(f._h))
    except Exception:
        time_filter.last_chained = None
    # which path the timed steps took: (fused cadences, steps they covered), and the landmark updates they held
    time_filter.last_cadences = (cad1[0] - cad0[0], cad1[1] - cad0[1])
    time_filter.last_updates = int(mm[warmup:].sum()) if mm is not None else len(traj_ids) * steps * m
    flags = [f.flags(b) for b in range(len(traj_ids))]
    mu = f.mean(0)
    assert not any(flags) and np.isfinite(mu).all(), "filter diverged during the benchmark"
    f.close()
    return dt, pass_ms, launches, dev_ms


def kernel_source_sha():
    """Hash of everything the pass's launch shape and code depend on: all of csrc/*.hip and csrc/*.h (the kernel
    lives in ekf_kernels.hip, its choice and launch shape in ekf_api.hip / ekf_cadence.hip, the layout in ekf_device.h)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "slam-duckietown_amd", "csrc")
    for path in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.h"))):
        h.update(os.path.basename(path).encode())
        h.update(open(path, "rb").read())
    return h.hexdigest()[:16]


class ClockSampler:
    """Shader clock (MHz) of the HIP device during a timed region, sampled from a host thread through librocm_smi64
    (ctypes; the device is matched by PCI bus id).  Measurement aid only: every failure degrades to `None`."""

    def __init__(self, device=0, period_s=0.002):
        import ctypes as C
        self.samples, self._stop, self._thread, self._ok = [], False, None, False
        self.period_s = period_s
        try:
            smi = C.CDLL("librocm_smi64.so")
            if smi.rsmi_init(C.c_uint64(0)) != 0:
                return
            count = C.c_uint32(0)
            smi.rsmi_num_monitor_devices(C.byref(count))
            want = None
            try:
                hip = C.CDLL("libamdhip64.so")
                buf = C.create_string_buffer(64)
                if hip.hipDeviceGetPCIBusId(buf, 64, int(device)) == 0:
                    dom, bus, dev_fn = buf.value.decode().split(":")
                    dv, fn = dev_fn.split(".")
                    want = (int(dom, 16) << 32) | (int(bus, 16) << 8) | (int(dv, 16) << 3) | int(fn, 16)
            except Exception:
                want = None
            self._ind = 0
            for i in range(count.value):
                bdf = C.c_uint64(0)
                if smi.rsmi_dev_pci_id_get(C.c_uint32(i), C.byref(bdf)) == 0 and want is not None and \
                        (bdf.value & 0xFFFFFFFF0000FFF8) == (want & 0xFFFFFFFF0000FFF8):   # domain, bus, device
                    self._ind = i

            class Freq(C.Structure):
                _fields_ = [("has_deep_sleep", C.c_bool), ("num_supported", C.c_uint32), ("current", C.c_uint32),
                            ("frequency", C.c_uint64 * 33)]
            self._Freq, self._smi, self._C = Freq, smi, C
            self._ok = self.read() is not None
        except Exception:
            self._ok = False

    def read(self):
        try:
            f = self._Freq()
            if self._smi.rsmi_dev_gpu_clk_freq_get(self._C.c_uint32(self._ind), 0, self._C.byref(f)) != 0:   # RSMI_CLK_TYPE_SYS
                return None
            if f.current >= 33:
                return None
            return f.frequency[f.current] / 1e6
        except Exception:
            return None

    def start(self):
        if self._ok and self._thread is None:
            import threading

            def loop():
                while not self._stop:
                    v = self.read()
                    if v is not None:
                        self.samples.append(v)
                    time.sleep(self.period_s)
            self._thread = threading.Thread(target=loop, daemon=True)
            self._thread.start()

    def stop(self):
        self._stop = True
        if self._thread is not None:
            self._thread.join(timeout=2.0)

    def summary(self):
        if not self.samples:
            return None
        a = np.array(self.samples)
        return {"min": float(a.min()), "mean": float(a.mean()), "max": float(a.max()), "samples": int(a.size),
                "source": "rsmi_dev_gpu_clk_freq_get(RSMI_CLK_TYPE_SYS) sampled from a host thread during the timed region"}


def pmc_traffic(key, options):
    """HBM bytes per launch of the pass kernel from the committed PMC summary (profiles/pass_traffic.json, written by
    tools/update_traffic.py from `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes).  The figure is only
    reported while the kernel source it was measured on is the one in the tree, and for the default options."""
    path = os.path.join(ROOT, "profiles", "pass_traffic.json")
    out = {"traffic": None}
    try:
        entry = json.load(open(path))[key]
    except Exception:
        out["traffic_note"] = "no PMC measurement committed for this configuration"
        return out
    sha = kernel_source_sha()
    if entry.get("kernel_source_sha256_16") != sha:
        out["traffic_note"] = (f"stale: profiles/pass_traffic.json was measured on kernel source "
                               f"{entry.get('kernel_source_sha256_16')}, the tree has {sha}")
    elif [o for o in options if o != "active_bound=0"]:
        out["traffic_note"] = "PMC traffic is measured for the default options only"
    else:
        out["traffic"] = entry["hbm_bytes_per_launch"]
        out["traffic_source"] = entry.get("source", "profiles/pass_traffic.json")
        out["traffic_kernel"] = entry.get("kernel")
        out["traffic_over_algorithmic"] = entry["hbm_bytes_per_launch"] / entry["algorithmic_bytes_per_launch"]
    return out


def dense_propagate_leg(sd, device, n_landmarks, reps=3):
    """ekf_predict_dense (general F, two fp64 MFMA GEMMs of n^3 MACs each): SURVEY 8(d) config 3's MFMA-busy
    evidence.  Device time from HIP events around the GEMM pair (uploads of F, Q excluded)."""
    n = 3 + 2 * n_landmarks
    rng = np.random.default_rng(0)
    F = np.eye(n) + 0.01 * rng.standard_normal((n, n))
    Q = np.eye(n) * 0.01
    f = sd.EkfSlam(n, batch=1, device=device)
    f.set_state_diag(np.zeros(n), np.ones(n))
    f.predict_dense(F, Q)                      # warm-up (allocates the work buffers)
    f.profile_enable(True)
    for _ in range(reps):
        f.predict_dense(F, Q)
    ms, cnt = f.profile_read()
    f.close()
    per = ms / max(cnt, 1)
    tf = 4.0 * n ** 3 / (per * 1e-3) / 1e12 if per > 0 else 0.0
    return {"workload": f"P <- F P F^T + Q, dense F, n={n}", "ms": per, "TFLOPs_fp64": tf,
            "frac_of_spec": tf / MFMA_F64_SPEC_TF, "frac_of_measured_mfma_rate": tf / MFMA_F64_MEASURED_TF}


def config5_leg(sd, sd_syn, shard, grp, device, m, steps=100, warmup=20):
    """BASELINE config 5: N = 8000 landmarks (P = 2.05 GB), 1 trajectory, block-diagonal P0 -- the dense pass
    (every state index treated as correlated) against the skip-unobserved pass (active bound: rows / columns of landmarks
    never observed are exactly zero off the diagonal and are skipped).  Both legs run the library's defaults (look-ahead
    on): equal to rounding -- the bound itself is exact, bit for bit with `lookahead=0`
    (tests/test_gpu_benchmarked_paths.py::test_config5_n8000_active_bound_bit_identical); the dense leg as timed here is
    checked against the oracle by ::test_config5_n8000_dense_leg_as_benchmarked."""
    N = 8000
    n = 3 + 2 * N
    tri = n * (n + 1) / 2.0
    out = {"workload": f"N={N} landmarks (n={n}), m={m} obs/step, 1 trajectory, block-diagonal P0, {steps} steps"}
    for key, bound in (("dense", 0), ("skip_unobserved", 1)):
        dt, pass_ms, launches, _ = time_filter(sd, sd_syn, shard, grp, device, [0], N, m, steps, warmup,
                                               options=[f"active_bound={bound}"])
        avg_ms = pass_ms / max(launches, 1)
        leg = {"value": steps / dt, "unit": "steps/s", "pass_avg_launch_ms": avg_ms, "pass_launches": launches,
               "pass_kernel": getattr(time_filter, "last_pass_kernel", "")}
        if bound == 0:
            gbs = 16.0 * tri / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
            leg.update({"pass_achieved_GBs": gbs, "pass_frac_of_hbm_peak": gbs / HBM_PEAK_GBS,
                        "alg_bytes_per_launch": 16.0 * tri})
            leg.update(pmc_traffic(f"N{N}_B1", []))
        else:
            leg["note"] = ("the pass covers the active triangle only: after warmup + steps the highest observed landmark "
                           f"is {m * (warmup + steps)} of {N}")
        out[key] = leg
    out["skip_over_dense"] = out["skip_unobserved"]["value"] / out["dense"]["value"]
    return out


def stream_leg(sd, sd_syn, shard, grp, device, n_landmarks, m, steps, warmup, options, label):
    """An uploaded stream of one trajectory (BASELINE configs 1 and 2 on the GPU): steps/s over `steps` steps."""
    dt, pass_ms, launches, _ = time_filter(sd, sd_syn, shard, grp, device, [0], n_landmarks, m, steps, warmup, options=options)
    return {"workload": f"N={n_landmarks}, m={m}, 1 trajectory, {steps} steps ({label})", "value": steps / dt,
            "unit": "steps/s", "pass_avg_launch_ms": pass_ms / max(launches, 1), "pass_launches": launches,
            "pass_kernel": getattr(time_filter, "last_pass_kernel", "")}


def observation_shape_leg(sd, sd_syn, shard, grp, device, traj_ids, n_landmarks, steps, warmup, options, m=None, variable=None):
    """The shapes the reference's loop produces (src/replay_no_ros.py:280-301, :436-480: whatever tags the window saw): a
    landmark count that is no power of two (`m`), or one that changes from step to step and from trajectory to trajectory
    with indices in no order (`variable` = (m_lo, m_hi)).  steps/s, landmark updates/s, covariance passes per step and the
    fraction of the timed steps that ran as fused cadences."""
    B = len(traj_ids)
    dt, pass_ms, launches, _ = time_filter(sd, sd_syn, shard, grp, device, traj_ids, n_landmarks, m or 0, steps, warmup,
                                           options=options, variable=variable)
    cads, cad_steps = time_filter.last_cadences
    shape = (f"m ~ uniform{{{variable[0]}..{variable[1]}}} per trajectory and step, scattered indices" if variable
             else f"m={m} obs/step, consecutive indices")
    return {"workload": f"N={n_landmarks}, {shape}, {B} trajectories, {steps} steps behind {warmup}",
            "value": B * steps / dt, "unit": "steps/s", "landmark_updates_per_s": time_filter.last_updates / dt,
            "mean_obs_per_step": time_filter.last_updates / (B * steps),
            "pass_launches": launches, "passes_per_step": launches / steps, "steps_per_pass": steps / max(launches, 1),
            "pass_avg_launch_ms": pass_ms / max(launches, 1), "pass_kernel": getattr(time_filter, "last_pass_kernel", ""),
            "fused_cadences": cads, "fused_fraction": cad_steps / steps}


def online_step_leg(sd, sd_syn, device, n_landmarks, batch, m, steps, warmup, options):
    """The call surface north_star says "drops into the existing Duckietown loop": one host-driven `EkfSlam.step` per
    EKF step (src/replay_no_ros.py:229-237, histogram_lane_filter_node.py:197-199 call the filter once per window) --
    the landmark indices of the next steps are not known in advance, so no fused cadence; inputs cross the C ABI (and
    PCIe: 352 B per trajectory) every call, the state stays in HBM.  Wall clock around `steps` calls + the final sync."""
    n = 3 + 2 * n_landmarks
    burst, bursts = 12, 8
    streams = [sd_syn.synthetic_stream(n_landmarks, warmup + steps + burst * bursts, m, t) for t in range(batch)]
    f = sd.EkfSlam(n, batch=batch, device=device)
    for opt in options:
        name, value = opt.split("=")
        f.set_option(name, int(value))
    for b, s in enumerate(streams):
        f.set_state_diag(s[0], s[1], b)
    cols = [[np.ascontiguousarray(np.stack([s[i][k] for s in streams])) for k in range(warmup + steps + burst * bursts)]
            for i in (2, 3, 4, 5, 6)]

    def one(k):
        if batch == 1:
            f.step(cols[0][k][0], cols[1][k][0], cols[2][k][0], cols[3][k][0], cols[4][k][0])
        else:
            f.step(cols[0][k], cols[1][k], cols[2][k], cols[3][k], cols[4][k])        # [batch, m] arrays: one block copy each

    for k in range(warmup):
        one(k)
    f.flush()
    f.sync()
    t0 = time.perf_counter()
    for k in range(warmup, warmup + steps):
        one(k)
    t_enq = time.perf_counter() - t0                   # host time to enqueue (the calls are asynchronous)
    f.flush()
    f.sync()
    dt = time.perf_counter() - t0
    # What a call costs the HOST: bursts of 12 calls behind a sync.  (The input ring has 16 slots: in the long run above the
    # host waits in the ring for the device, so `host_enqueue_ms_per_call` there is the DEVICE's time per step, not host work.)
    host = []
    k = warmup + steps
    for _ in range(bursts):
        f.flush()
        f.sync()
        t0 = time.perf_counter()
        for _ in range(burst):
            one(k)
            k += 1
        host.append((time.perf_counter() - t0) / burst)
    f.flush()
    f.sync()
    assert not any(f.flags(b) for b in range(batch)) and np.isfinite(f.mean(0)).all()
    f.close()
    return {"workload": f"EkfSlam.step per call, N={n_landmarks}, m={m}, {batch} trajectories, {steps} calls",
            "value": batch * steps / dt, "unit": "steps/s", "ms_per_call": dt / steps * 1e3,
            "host_enqueue_ms_per_call": t_enq / steps * 1e3,
            "host_ms_per_call_unblocked": float(np.median(host)) * 1e3,
            "note": "ms_per_call is device-bound (the host enqueues a call in host_ms_per_call_unblocked and then waits in the "
                    "16-slot input ring for the device)"}


def drop_in_leg(sd, sd_syn, n_landmarks, m, calls, warm, trust_identity=False):
    """`EKF_pose_estimation(ang, lin, mean, cov, dt, detections, TAG_INDEX)` exactly as the reference's loop calls it
    (src/replay_no_ros.py:229-237: the returned mean / covariance are passed back in): host association, the step on the
    GPU, and the download of the n x n covariance EVERY call (8 n^2 bytes over PCIe) -- what a user who changes nothing
    but the import gets.  Pre-sized state and TAG_INDEX (the reference's god mode, :140-157).  ms per call, median."""
    from types import SimpleNamespace
    mean0, diag0, lin, ang, idx, zr, zb = sd_syn.synthetic_stream(n_landmarks, warm + calls, m, 0)
    tag_index = {1000 + i: i for i in range(n_landmarks)}
    mean, cov = mean0.copy(), np.diag(diag0)
    eye = np.eye(3)
    times = []
    from slam_duckietown_amd import ekf_bindings as eb
    eb.DROP_IN_TRUST_IDENTITY = bool(trust_identity)
    try:
        for k in range(warm + calls):
            xr, yr = zr[k] * np.cos(zb[k]), zr[k] * np.sin(zb[k])
            tags = [SimpleNamespace(tag_id=1000 + int(i), pose_R=eye, pose_t=np.array([[-y], [0.0], [x]]), pose_err=0.0)
                    for i, x, y in zip(idx[k], xr, yr)]
            t0 = time.perf_counter()
            mean, cov, _tags = sd.EKF_pose_estimation(ang[k], lin[k], mean, cov, 0.7, [(float(k), tags)], tag_index)
            if k >= warm:
                times.append(time.perf_counter() - t0)
    finally:
        eb.DROP_IN_TRUST_IDENTITY = False
    assert np.isfinite(mean).all() and np.isfinite(cov).all() and len(mean) == 3 + 2 * n_landmarks
    med = float(np.median(times))
    return {"workload": f"EKF_pose_estimation per call incl. the covariance download, N={n_landmarks}, m={m}, {calls} calls"
                        + (", DROP_IN_TRUST_IDENTITY (no upload of the arrays the previous call returned)" if trust_identity else ""),
            "ms_per_call": med * 1e3, "value": 1.0 / med, "unit": "steps/s", "download_bytes_per_call": 8 * len(mean) ** 2}


def cpu_dense_steps(n_landmarks, m, steps, budget_s, min_steps):
    """Median seconds per step of the oracle's reference-shaped dense NumPy step on the host cores."""
    from oracle import ekf_oracle as orc          # checker / baseline only
    mean0, diag0, lin, ang, idx, zr, zb = orc.synthetic_stream(n_landmarks, steps, m, 0)
    cfg = orc.EkfConfig()
    mean, cov = mean0.copy(), np.diag(diag0)
    times = []
    t_all = time.perf_counter()
    for k in range(steps):
        t0 = time.perf_counter()
        mean, cov = orc.ekf_step_dense(mean, cov, lin[k], ang[k], idx[k], zr[k], zb[k], cfg)
        times.append(time.perf_counter() - t0)
        if time.perf_counter() - t_all > budget_s and len(times) >= min_steps:
            break
    return times


def cpu_whole_function(n_landmarks, m, calls):
    """The oracle's restatement of the WHOLE reference function (association + augmentation + dense step,
    oracle.ekf_pose_estimation_dense <-> src/replay_no_ros.py:269-482) on the same detections the `drop_in` leg feeds the
    GPU drop-in: the like-for-like CPU figure for one EKF_pose_estimation call."""
    from types import SimpleNamespace
    from oracle import ekf_oracle as orc          # checker / baseline only
    mean0, diag0, lin, ang, idx, zr, zb = orc.synthetic_stream(n_landmarks, calls, m, 0)
    tag_index = {1000 + i: i for i in range(n_landmarks)}
    mean, cov = mean0.copy(), np.diag(diag0)
    cfg = orc.EkfConfig()
    eye = np.eye(3)
    times = []
    for k in range(calls):
        xr, yr = zr[k] * np.cos(zb[k]), zr[k] * np.sin(zb[k])
        tags = [SimpleNamespace(tag_id=1000 + int(i), pose_R=eye, pose_t=np.array([[-y], [0.0], [x]]), pose_err=0.0)
                for i, x, y in zip(idx[k], xr, yr)]
        t0 = time.perf_counter()
        mean, cov, _ = orc.ekf_pose_estimation_dense(ang[k], lin[k], mean, cov, 0.7, [(float(k), tags)], tag_index, cfg)
        times.append(time.perf_counter() - t0)
    med = float(np.median(times[10:]))
    return {"value": 1.0 / med, "unit": "steps/s", "ms_per_call": med * 1e3,
            "sample": f"{calls} calls of the whole function (association + dense step), N={n_landmarks}, m={m}, median"}


def cpu_baseline(n_landmarks, m, budget_s=14.0):
    """Reference-shaped dense NumPy step (oracle/ekf_oracle.py::ekf_step_dense) on the host cores: the headline size
    (>= 3 steps), and under `by_config` SURVEY 8(d)'s other legs -- N = 500 (>= 10 steps), N = 20 (500 steps) -- plus
    N = 12, m = 3 (the reference's real map, src/replay_no_ros.py:26) for the drop-in comparison."""
    times = cpu_dense_steps(n_landmarks, m, 5, budget_s, 3)
    blas = "BLAS unknown"
    try:
        from threadpoolctl import threadpool_info
        pools = threadpool_info()
        cores = max([p.get("num_threads", 1) for p in pools] or [1])
        blas = ", ".join(sorted({f"{p.get('internal_api', '?')} {p.get('version', '?')}" for p in pools})) or blas
    except Exception:
        cores = os.cpu_count() or 1
    med = float(np.median(times))
    by = {}
    by["drop_in_N12"] = cpu_whole_function(12, 3, 300)
    by["drop_in_N20"] = cpu_whole_function(20, m, 200)
    for key, (N2, m2, steps2, min2) in {"N500": (500, m, 12, 10), "N20": (20, m, 500, 500), "N12": (12, 3, 500, 500)}.items():
        t2 = cpu_dense_steps(N2, m2, steps2, 6.0, min2)
        med2 = float(np.median(t2))
        by[key] = {"value": 1.0 / med2, "unit": "steps/s", "ms_per_step": med2 * 1e3,
                   "sample": f"{len(t2)} steps of N={N2}, m={m2}, 1 trajectory, dense NumPy (oracle.ekf_step_dense), median"}
    return {"value": 1.0 / med, "unit": "steps/s", "cores": int(cores), "kind": "port",
            "sample": f"{len(times)} steps of N={n_landmarks}, m={m}, 1 trajectory, dense NumPy "
                      f"(oracle.ekf_step_dense), median {med * 1e3:.0f} ms/step; NumPy {np.__version__}, {blas}, "
                      f"os.cpu_count() = {os.cpu_count()}",
            "by_config": by}


def cadence_breakdown(sd, sd_syn, device, n_landmarks, m, options, steps=100, warmup=20):
    """Per-kernel device time of a fused cadence from HIP event pairs around EVERY launch ("profile_kernels": a diagnostic
    run of its own -- each record costs its stream ~6 us, so this run is slower than the timed one; the durations are what
    it is for).  -> us per launch: solve, chain (or look-ahead gather), panel, pass."""
    n = 3 + 2 * n_landmarks
    streams, lin, ang, idx, zr, zb, mm = make_streams(sd_syn, [0], n_landmarks, warmup + steps, m)
    f = sd.EkfSlam(n, batch=1, device=device)
    for opt in options:
        name, value = opt.split("=")
        f.set_option(name, int(value))
    f.set_state_diag(streams[0][0], streams[0][1], 0)
    f.stream_upload(lin, ang, idx, zr, zb, mm)
    f.stream_run(0, warmup)
    f.flush()
    f.set_option("profile_kernels", 1)
    f.set_option("profile_stride", 1)
    f.profile_enable(True)
    f.stream_run(warmup, steps)
    f.flush()
    f.sync()
    out = {}
    for key, cls in (("solve", 1), ("chain_or_gather", 2), ("panel", 3)):
        ms, cnt = f.profile_read_class(cls)
        out[key] = ms / max(cnt, 1) * 1e3
    ms, cnt = f.profile_read()
    out["pass"] = ms / max(cnt, 1) * 1e3
    f.profile_enable(False)
    f.close()
    return out


SECONDARY_LEGS = ["single_trajectory", "obs_1_per_step", "obs_5", "obs_12", "variable_m", "constant_m4", "config5", "steady_state", "config1", "config2", "online_step",
                  "drop_in", "dense_propagate"]


def secondary_leg(name, args):
    """One secondary leg of the JSON line -> {key: value}."""
    import slam_duckietown_amd as sd
    import slam_duckietown_amd.sharding as shard
    import slam_duckietown_amd.synthetic as sd_syn
    grp = shard.RankGroup()
    sd.load_library()
    dev, B, n = 0, args.trajectories, 3 + 2 * args.landmarks
    tri = n * (n + 1) / 2.0
    traj_ids = list(range(B))
    if name == "single_trajectory":
        # (at least 200 steps behind 20: the driver's 20 steps of one trajectory are 0.45 ms, which times the first launches)
        steps1, warm1 = max(args.steps, 200), max(args.warmup, 20)
        dt1, p1, l1, _ = time_filter(sd, sd_syn, shard, grp, dev, [0], args.landmarks, args.obs, steps1, warm1,
                                     options=args.option)
        a1 = 16.0 * tri / ((p1 / max(l1, 1)) * 1e-3) / 1e9 if p1 > 0 else 0.0
        leg = {"workload": f"N={args.landmarks}, m={args.obs}, 1 trajectory (BASELINE config 3), {steps1} steps",
               "value": steps1 / dt1, "unit": "steps/s", "pass_avg_launch_ms": p1 / max(l1, 1), "pass_launches": l1,
               "pass_launches_timed": getattr(time_filter, "last_timed", 0),
               "pass_achieved_GBs": a1, "pass_frac_of_hbm_peak": a1 / HBM_PEAK_GBS,
               "chained_solves": getattr(time_filter, "last_chained", None)}
        # where a cadence's time goes: the period from the timed run above, the launches' own durations from an instrumented
        # run (event pairs around every launch).  Chained: the handle's stream runs solve + chain, the second stream panel +
        # pass, side by side; look-ahead ("chain=0"): panel -> gather -> { solve | pass }
        try:
            br = cadence_breakdown(sd, sd_syn, dev, args.landmarks, args.obs, args.option)
            period = dt1 / steps1 * (40.0 / max(args.obs, 1)) * 1e6
            main_busy, second_busy = br["solve"] + br["chain_or_gather"], br["panel"] + br["pass"]
            chained_run = bool(leg["chained_solves"])
            critical = max(main_busy, second_busy) if chained_run else br["panel"] + br["chain_or_gather"] + max(br["solve"], br["pass"])
            leg["cadence_us"] = {"period": period, **br, "stream_busy": main_busy, "second_stream_busy": second_busy,
                                 "critical_path": critical, "bubbles": period - critical,
                                 "note": "period: the timed run; kernel durations: HIP event pairs around every launch in an "
                                         "instrumented run of its own (option profile_kernels; every record costs its stream "
                                         "~6 us, and in a chained run a launch's duration includes what it waits for on the "
                                         "device-side counters: the panel launch its solve and the next chain launch's "
                                         "gathers, the chain launch the previous pass); critical_path: the busier stream when "
                                         "the solves are chained, else panel + gather + max(solve, pass); the un-instrumented "
                                         "timeline is profiles/r06_chained_solves.txt"}
        except Exception as e:                          # (a measurement aid: never fails the line)
            leg["cadence_us"] = {"error": str(e)}
        return {name: leg}
    if name == "obs_1_per_step":
        # one observation per step: 2 ranks per step, a covariance pass every 40 steps -- timed over whole
        # cadences only (a shorter run would charge a full pass to a fraction of the steps it serves)
        steps_m1 = -(-args.steps // 40) * 40
        dtm, _, _, _ = time_filter(sd, sd_syn, shard, grp, dev, traj_ids, args.landmarks, 1, steps_m1, 40,
                                   profile_leg=False, options=args.option)
        return {name: {"workload": f"N={args.landmarks}, m=1 obs/step, {B} trajectories, {steps_m1} steps (whole 40-step pass "
                                   "cadences)", "value": B * steps_m1 / dtm, "unit": "steps/s"}}
    if name in ("obs_5", "obs_12", "constant_m4"):
        # 240 steps = whole cadences at every packing (m = 4: 10 steps per pass, m = 5: 8, m = 12: 3)
        mo = {"obs_5": 5, "obs_12": 12, "constant_m4": 4}[name]
        return {name: observation_shape_leg(sd, sd_syn, shard, grp, dev, traj_ids, args.landmarks, 240, 40, args.option, m=mo)}
    if name == "variable_m":
        # m ~ uniform{0..8}: mean 4 landmarks per step -- compare with `constant_m4` (the same mean rank count)
        leg = observation_shape_leg(sd, sd_syn, shard, grp, dev, traj_ids, args.landmarks, 240, 40, args.option, variable=(0, 8))
        # ... and what SURVEY 8 calls the real data's shape, 0 - 3 tags per window, one trajectory (until round 4 such a stream
        # never formed a fused cadence)
        leg["x1_m0to3"] = observation_shape_leg(sd, sd_syn, shard, grp, dev, [0], args.landmarks, 600, 60, args.option,
                                                variable=(0, 3))
        return {name: leg}
    if name == "config5":
        return {name: config5_leg(sd, sd_syn, shard, grp, dev, args.obs)}
    if name == "steady_state":
        # The contract's timed region starts 20 steps (the driver: 5) into a stream from a block-diagonal P0: most landmarks
        # have not been observed yet and most of V / W is exact zeros.  fp64 MFMA power -- and with it the clock the part
        # holds under the pass -- depends on the operands: once every landmark has been observed (one sweep = N / m steps) the
        # covariance is dense and the same pass takes ~9 % longer (profiles/r04_dense_operands.txt).  This leg starts its
        # timed region behind a full sweep: what a long-running filter sees.
        sweep = -(-args.landmarks // max(args.obs, 1)) + 10
        steps = max(args.steps, 100)
        clk = ClockSampler(dev, period_s=0.001)
        dts, ps, ls, _ = time_filter(sd, sd_syn, shard, grp, dev, traj_ids, args.landmarks, args.obs, steps, sweep,
                                     options=args.option, clock=clk)
        avg = ps / max(ls, 1)
        per_launch_s = dts / steps * (steps / max(ls, 1))           # the whole cadence: solve + panel launch + pass + gaps
        return {name: {"workload": f"the headline workload timed behind a full sweep of the landmarks ({sweep} warm-up steps: "
                                   f"dense covariance), {steps} steps",
                       "value": B * steps / dts, "unit": "steps/s", "pass_avg_launch_ms": avg, "pass_launches": ls,
                       "pass_launches_timed": getattr(time_filter, "last_timed", 0),
                       "pass_frac_of_hbm_peak": (B * 16.0 * tri / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS) if avg > 0 else 0.0,
                       "whole_cadence_frac": (B * 16.0 * tri / per_launch_s / 1e9 / HBM_PEAK_GBS) if ls > 0 else 0.0,
                       # the shader clock over this leg's timed region (tens of ms: enough samples to mean something)
                       "sclk_mhz": clk.summary() if dts >= 0.015 else None}}
    if name == "config1":
        # N = 20 fits a CU's LDS: the small-state path (csrc/ekf_small.hip) runs the whole stream as ONE launch, P resident in LDS;
        # a bank of such filters (a Monte-Carlo run at the reference's map size) is one workgroup per trajectory
        leg = stream_leg(sd, sd_syn, shard, grp, dev, 20, args.obs, 500, 20, args.option, "BASELINE config 1's size on the GPU")
        dtb, _, _, _ = time_filter(sd, sd_syn, shard, grp, dev, list(range(256)), 20, args.obs, 500, 20, profile_leg=False,
                                   options=args.option)
        leg["bank_of_256"] = {"workload": "N=20, m=8, 256 trajectories (one workgroup each), 500 steps", "value": 256 * 500 / dtb,
                              "unit": "steps/s"}
        # three workgroups are resident per CU (148 VGPRs): the throughput of the path (tools/small_bank_sweep.py)
        dtc, _, _, _ = time_filter(sd, sd_syn, shard, grp, dev, [t % 256 for t in range(768)], 20, args.obs, 500, 20,
                                   profile_leg=False, options=args.option)
        leg["bank_of_768"] = {"workload": "N=20, m=8, 768 trajectories (the 256 above, three times), 500 steps",
                              "value": 768 * 500 / dtc, "unit": "steps/s"}
        return {name: leg}
    if name == "config2":
        return {name: stream_leg(sd, sd_syn, shard, grp, dev, 500, args.obs, 500, 20, args.option, "BASELINE config 2")}
    if name == "online_step":
        return {name: {f"N{args.landmarks}_x{B}": online_step_leg(sd, sd_syn, dev, args.landmarks, B, args.obs, 100, 10, args.option),
                       f"N{args.landmarks}_x1": online_step_leg(sd, sd_syn, dev, args.landmarks, 1, args.obs, 200, 10, args.option)}}
    if name == "drop_in":
        return {name: {"N12": drop_in_leg(sd, sd_syn, 12, 3, 200, 10), "N20": drop_in_leg(sd, sd_syn, 20, args.obs, 200, 10),
                       "N500": drop_in_leg(sd, sd_syn, 500, args.obs, 40, 5),
                       f"N{args.landmarks}": drop_in_leg(sd, sd_syn, args.landmarks, args.obs, 12, 3),
                       # the same with the opt-in residency for large states (ekf_bindings.DROP_IN_TRUST_IDENTITY)
                       f"N{args.landmarks}_trusted": drop_in_leg(sd, sd_syn, args.landmarks, args.obs, 12, 3, trust_identity=True)}}
    if name == "dense_propagate":
        return {name: dense_propagate_leg(sd, dev, args.landmarks)}
    raise SystemExit(f"unknown leg {name}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--landmarks", type=int, default=2000)
    ap.add_argument("--obs", type=int, default=8)
    ap.add_argument("--trajectories", type=int, default=32, help="trajectories per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-single", action="store_true")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE",
                    help="ekf_set_option knob, e.g. flush_every=3 (default: library defaults)")
    ap.add_argument("--leg", default=None, help="run ONE secondary leg (see SECONDARY_LEGS) and print it as a JSON line")
    args = ap.parse_args()
    if args.leg:
        if not any(o.startswith("active_bound=") for o in args.option):
            args.option = ["active_bound=0"] + args.option
        sys.stdout.flush()
        fd = os.dup(1)
        os.dup2(2, 1)
        leg = secondary_leg(args.leg, args)
        os.write(fd, (json.dumps(leg) + "\n").encode())
        return
    # stdout carries exactly ONE line, the JSON: libraries that chat on fd 1 (gloo prints its peer count there)
    # are sent to stderr for the whole run, the result is written to the saved descriptor at the end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    # The headline treats every state index as correlated: the active-bound shortcut (rows/cols never correlated yet are
    # skipped, exact) would make the first N/m steps of a block-diagonal start cheaper than a dense covariance, so it is
    # off unless asked for.  (The OPERANDS of the timed region are still those of a young filter -- most landmarks not yet
    # observed, most of V / W exact zeros -- see the `steady_state` leg for the same workload on a dense covariance.)
    if not any(o.startswith("active_bound=") for o in args.option):
        args.option = ["active_bound=0"] + args.option

    import slam_duckietown_amd.sharding as shard
    grp = shard.RankGroup()
    rank, local_rank, world = grp.rank, grp.local_rank, grp.world
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import slam_duckietown_amd as sd
    import slam_duckietown_amd.synthetic as sd_syn
    sd.load_library()

    # one rank per GPU (LOCAL_RANK = device index on the driver's 8-GPU node); the modulo only matters
    # when more ranks than devices are launched to rehearse the multi-rank path on a smaller box
    n_dev = sd.device_count()
    if n_dev < 1:
        raise SystemExit("no MI355X visible: the HIP path has no CPU fallback")
    local_rank = local_rank % n_dev
    B = args.trajectories
    n = 3 + 2 * args.landmarks
    traj_ids = shard.shard_trajectories(B * world, world, rank)
    clk = ClockSampler(local_rank)
    dt, pass_ms, launches, dev_ms = time_filter(sd, sd_syn, shard, grp, local_rank, traj_ids, args.landmarks, args.obs,
                                                args.steps, args.warmup, options=args.option, clock=clk)
    value = shard.aggregate_steps_per_second(len(traj_ids) * args.steps, grp, dt)
    rank_dts = grp.gather_over_ranks(getattr(shard.timed_region, "last_local_seconds", dt))
    rank_dev_ms = grp.gather_over_ranks(dev_ms / args.steps)

    out = None
    if rank == 0:
        # The covariance pass (k_flush) reads and writes the stored (upper) triangle of every trajectory's P
        # exactly once per launch, whatever the number of steps it folds in: 2 * 8 * n(n+1)/2 bytes each.
        tri = n * (n + 1) / 2.0
        alg_bytes = B * 16.0 * tri
        avg_s = (pass_ms / max(launches, 1)) * 1e-3
        achieved = alg_bytes / avg_s / 1e9 if avg_s > 0 else 0.0
        steps_per_launch = args.steps / max(launches, 1)
        ranks = steps_per_launch * 2 * args.obs                               # exactly 2 ranks per landmark update (packed cadences)
        out = {
            "metric": "EKF update steps/sec", "value": value, "unit": "steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"N={args.landmarks} landmarks (n={n}), m={args.obs} obs/step, "
                                   f"{B} trajectories/GPU x {world} GPU(s), fused predict+update step",
                       "landmarks": args.landmarks, "state_dim": n, "obs_per_step": args.obs,
                       "trajectories_per_gpu": B, "parallelism": f"trajectory-sharded x{world}, no collective",
                       "options": args.option},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": getattr(time_filter, "last_pass_kernel", "") or "ekf::k_flush",
                         "alg_bytes_per_launch": alg_bytes, "avg_launch_ms": avg_s * 1e3, "launches": launches,
                         "launches_timed": getattr(time_filter, "last_timed", 0),
                         "steps_per_launch": steps_per_launch,
                         "step_equivalent_GBs": (B * 16.0 * n * n * steps_per_launch / avg_s / 1e9) if avg_s > 0 else 0.0,
                         # the same bytes over the WHOLE cadence (solve + panel launch + pass + gaps: what `value` is made of)
                         "whole_cadence_frac": (alg_bytes / (dt / args.steps * steps_per_launch) / 1e9 / HBM_PEAK_GBS)
                                               if launches > 0 else 0.0,
                         "mfma": {"achieved": 2.0 * ranks * B * tri / avg_s / 1e12 if avg_s > 0 else 0.0,
                                  "unit": "TFLOP/s fp64", "peak_spec": MFMA_F64_SPEC_TF,
                                  "peak_measured": MFMA_F64_MEASURED_TF,
                                  "frac_of_spec": (2.0 * ranks * B * tri / avg_s / 1e12 / MFMA_F64_SPEC_TF) if avg_s > 0 else 0.0,
                                  "frac_of_measured": (2.0 * ranks * B * tri / avg_s / 1e12 / MFMA_F64_MEASURED_TF) if avg_s > 0 else 0.0},
                         "note": "one launch applies the pending rank-K update of steps_per_launch steps to the "
                                 "stored upper triangle of P (P is symmetric; the lower triangle is never read): "
                                 "SURVEY 8(d)'s 16 n^2 bytes per step become 8 n(n+1) bytes per LAUNCH; achieved/frac "
                                 "count the bytes this launch must move (one read + one write of the triangle), "
                                 "step_equivalent_GBs = SURVEY's 16 n^2 per step x steps folded in / launch time; "
                                 "at the default 5 steps (80 ranks = 10 flop/B, the ridge of this part) per launch the matrix side "
                                 "(0.65 ms alone) and the memory side (0.68 ms alone, 6.05 TB/s) of the row-slab pass are balanced "
                                 "(DESIGN.md section 4), see `mfma`"},
            "device_ms_per_step": dev_ms / args.steps,
            # every rank's own elapsed time of the timed region (value uses their maximum) beside its own DEVICE time per step
            # (HIP events on the handle's stream): host dispatch skew and device time can be told apart from one line
            "rank_dt_ms": [x * 1e3 for x in rank_dts],
            "rank_device_ms_per_step": list(rank_dev_ms),
            # the shader clock while the timed region ran (sampled every 2 ms from a host thread) -- only where the region is
            # long enough for the samples to see it (>= 50 ms; the driver's 20 steps are 3.6 ms: two samples of the idle clock);
            # `sclk_mhz_steady_state` is the same over the steady_state leg
            "sclk_mhz": clk.summary() if dt >= 0.05 else None,
        }
        if os.environ.get("EKFSLAM_HIP_VARIANT"):      # a diagnostic build of the library was timed: not a product number
            out["library_variant"] = os.environ["EKFSLAM_HIP_VARIANT"]
        out["roofline"].update(pmc_traffic(f"N{args.landmarks}_B{B}", args.option))
    if world == 1 and rank == 0:
        if not args.no_single:
            # Secondary legs, one handle after another in this process.  (Until round 4 the second handle of a process ran its
            # look-ahead -- the pass on a second stream beside the next solve -- 10 % slower than the first: HIP's mapping of
            # freshly created streams onto hardware queues after the first handle's streams had been destroyed.  The library now
            # parks a destroyed handle's stream pair and hands it to the next handle: tools/leg_order_probe.py,
            # profiles/r04_dense_operands.txt part 2.  `--leg NAME` runs one leg in a process of its own.)
            for name in SECONDARY_LEGS:
                out.update(secondary_leg(name, args))
            # the same pass on a dense covariance (the `steady_state` leg): what a long-running filter sees; `frac` above is the
            # young filter the contract's protocol times (most of V / W still exact zeros)
            ss = out.get("steady_state", {})
            if ss.get("pass_avg_launch_ms"):
                out["roofline"]["avg_launch_ms_steady_state"] = ss["pass_avg_launch_ms"]
                out["roofline"]["frac_steady_state"] = ss["pass_frac_of_hbm_peak"]
                out["roofline"]["achieved_steady_state"] = ss["pass_frac_of_hbm_peak"] * HBM_PEAK_GBS
                out["roofline"]["whole_cadence_frac_steady_state"] = ss.get("whole_cadence_frac")
                out["value_steady_state"] = ss["value"]
                out["sclk_mhz_steady_state"] = ss.get("sclk_mhz")
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.landmarks, args.obs)
    if rank == 0:
        if "cpu_baseline" not in out:
            out["cpu_baseline"] = None
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    grp.close()


if __name__ == "__main__":
    main()

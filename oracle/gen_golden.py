#!/usr/bin/env python3
"""Generate tests/golden/*.npz by executing the REFERENCE's own function bodies.

Runs only in the build container (needs /root/reference); the GPU box gets the
fixtures, never the reference.  The reference modules cannot be imported whole
(they need cv2 / dt_apriltags / matplotlib windows and src/EKF-SLAM.py runs broken
module-level code), so the functions on the hot path are
lifted with ``ast`` -- FunctionDef nodes plus the upper-case module constants --
and executed unmodified against NumPy:

  src/replay_no_ros.py : EKF_pose_estimation (:269-482), delta_phi (:250-266),
                         displacement (:484-497), flags (:15-36, incl. IGNORE_TAGS :36-37 and
                         ENABLE_GOD_EKF / GOD_SECRET_KEY :23-26), and the offline
                         loop replay (:66-248) with its I/O boundary replaced
                         (frames looked up in a table instead of cv2 + dt_apriltags,
                         plot_path recording its arguments instead of drawing)
  src/EKF-SLAM.py      : predict (:29-56), update (:59-84), noise globals (:12-13)

Fixtures hold inputs and the reference's outputs only (data, no source text).
Usage:  python oracle/gen_golden.py [--ref /root/reference] [--out tests/golden]
"""
from __future__ import annotations

import argparse
import ast
import os
import sys
from collections import defaultdict
from types import SimpleNamespace
from typing import Tuple

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle.ekf_oracle import synthetic_stream  # noqa: E402  (input generator only)


def lift(path: str, func_names, const_pred):
    tree = ast.parse(open(path).read())
    keep = []
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in func_names:
            keep.append(node)
        elif isinstance(node, ast.Assign) and all(isinstance(t, ast.Name) and const_pred(t.id) for t in node.targets):
            keep.append(node)
    ns = {"np": np, "defaultdict": defaultdict, "Tuple": Tuple}
    exec(compile(ast.Module(body=keep, type_ignores=[]), os.path.basename(path), "exec"), ns)
    return ns


def load_reference(ref_root: str):
    slam = lift(os.path.join(ref_root, "src", "replay_no_ros.py"),
                {"EKF_pose_estimation", "delta_phi", "displacement", "replay"}, str.isupper)
    proto = lift(os.path.join(ref_root, "src", "EKF-SLAM.py"),
                 {"predict", "update"}, lambda s: s in ("motion_noise", "observation_noise"))
    return slam, proto


def make_tag(tag_id, x_r, y_r, err):
    # robot frame (x_r forward, y_r left) -> camera pose_t = [[-y_r],[h],[x_r]]  (replay_no_ros.py:321)
    return SimpleNamespace(tag_id=int(tag_id), pose_R=np.eye(3),
                           pose_t=np.array([[-y_r], [0.05], [x_r]]), pose_err=float(err))


def replay_scenario(seed: int, steps: int, n_tags: int, big_turns: bool):
    """A small world with progressively discovered tags, several frames per window."""
    rng = np.random.default_rng(seed)
    lm = np.stack([rng.uniform(-1.0, 1.0, n_tags), rng.uniform(-0.8, 1.2, n_tags)], axis=1)
    tag_ids = rng.permutation(np.arange(20, 20 + 3 * n_tags))[:n_tags]
    pose = np.zeros(3)
    rec = dict(step=[], frame=[], tag_id=[], pose_t=[], err=[])
    lin = np.zeros(steps)
    ang = np.zeros(steps)
    for k in range(steps):
        lin[k] = rng.uniform(0.0, 0.08)
        if k % 7 == 3:
            ang[k] = rng.uniform(-0.009, 0.009)          # straight branch (:376)
        elif big_turns:
            ang[k] = rng.uniform(0.5, 0.9)               # crosses +-pi quickly -> theta wrap (:397)
        else:
            ang[k] = rng.uniform(-0.25, 0.35)
        n_frames = int(rng.integers(0, 4))
        th = pose[2]
        for f in range(n_frames):
            order = rng.permutation(n_tags)
            for t in order:
                if rng.random() < 0.45:
                    continue
                d = lm[t] - pose[0:2]
                xr = np.cos(th) * d[0] + np.sin(th) * d[1] + rng.normal(0, 0.02)
                yr = -np.sin(th) * d[0] + np.cos(th) * d[1] + rng.normal(0, 0.02)
                # tags behind the robot give |bearing| near pi -> innovation wrap (:458)
                rec["step"].append(k); rec["frame"].append(f); rec["tag_id"].append(tag_ids[t])
                rec["pose_t"].append([-yr, 0.05, xr]); rec["err"].append(rng.uniform(1e-4, 1e-2))
        # truth moves with the same kinematics, loosely (only inputs matter for parity)
        if abs(ang[k]) > 1e-2:
            r = lin[k] / ang[k]
            pose = pose + np.array([-r * np.sin(th) + r * np.sin(th + ang[k]),
                                    r * np.cos(th) - r * np.cos(th + ang[k]), ang[k]])
        else:
            pose = pose + np.array([lin[k] * np.cos(th), lin[k] * np.sin(th), 0.0])
    out = {k: np.asarray(v) for k, v in rec.items()}
    out["pose_t"] = out["pose_t"].reshape(-1, 3).astype(float)
    return lin, ang, out


def detections_for_step(rec, k):
    sel = np.nonzero(rec["step"] == k)[0]
    frames = {}
    for i in sel:
        frames.setdefault(int(rec["frame"][i]), []).append(
            SimpleNamespace(tag_id=int(rec["tag_id"][i]), pose_R=np.eye(3),
                            pose_t=rec["pose_t"][i].reshape(3, 1).copy(), pose_err=float(rec["err"][i])))
    return [(float(k) + 0.1 * f, tags) for f, tags in sorted(frames.items())]


def run_replay(slam, lin, ang, rec, flags):
    saved = {k: slam[k] for k in flags}
    slam.update(flags)
    try:
        mean = np.array([0.0, 0.0, 0.0])
        cov = np.eye(3) * slam["MOTION_MODEL_VARIANCE"]          # replay_no_ros.py:69-70
        tag_index = {}
        means, covs, sizes, obs_idx = [], [], [], []
        for k in range(len(lin)):
            det = detections_for_step(rec, k)
            mean, cov, tp = slam["EKF_pose_estimation"](ang[k], lin[k], mean.copy(), cov.copy(), 0.7, det, tag_index)
            means.append(mean.copy()); covs.append(cov.copy()); sizes.append(len(mean))
            obs_idx.append(list(tp.keys()))
        nmax = max(sizes)
        M = np.zeros((len(lin), nmax)); Pm = np.zeros((len(lin), nmax, nmax))
        for k, (mu, P) in enumerate(zip(means, covs)):
            M[k, :len(mu)] = mu; Pm[k, :len(mu), :len(mu)] = P
        order = np.full((len(lin), 16), -1, dtype=np.int64)
        for k, o in enumerate(obs_idx):
            order[k, :len(o)] = o
        ti = np.array(sorted(tag_index.items(), key=lambda kv: kv[1]), dtype=np.int64).reshape(-1, 2)
        return dict(out_mean=M, out_cov=Pm, out_size=np.array(sizes), out_obs_order=order, out_tag_index=ti)
    finally:
        slam.update(saved)


def tags_from_obs(idx, zr, zb):
    tags = []
    for i, r, b in zip(idx, zr, zb):
        xr, yr = r * np.cos(b), r * np.sin(b)
        tags.append(SimpleNamespace(tag_id=1000 + int(i), pose_R=np.eye(3),
                                    pose_t=np.array([[-yr], [0.0], [xr]]), pose_err=0.0))
    return tags


def run_stream(slam, n_landmarks, steps, m, keep_every, rows_only=None):
    """Synthetic stream of SURVEY 8(d) through the reference, god-mode style pre-sized state."""
    mean0, diag0, lin, ang, idx, zr, zb = synthetic_stream(n_landmarks, steps, m, 0)
    mean = mean0.copy(); cov = np.diag(diag0)
    tag_index = {1000 + i: i for i in range(n_landmarks)}
    means = np.zeros((steps, len(mean0)))
    zr_eff = np.zeros_like(zr); zb_eff = np.zeros_like(zb)
    kept, kept_steps, diags, fro = [], [], [], []
    for k in range(steps):
        det = [(float(k), tags_from_obs(idx[k], zr[k], zb[k]))]
        mean, cov, tp = slam["EKF_pose_estimation"](ang[k], lin[k], mean, cov, 0.7, det, tag_index)
        assert list(tp.keys()) == list(idx[k]), "gate dropped a synthetic observation"
        # what the reference actually used after its pose_t round trip (:321-330)
        zr_eff[k] = [tp[i][4] for i in idx[k]]; zb_eff[k] = [tp[i][5] for i in idx[k]]
        means[k] = mean
        diags.append(np.diag(cov).copy()); fro.append(np.linalg.norm(cov))
        if (k + 1) % keep_every == 0 or k == steps - 1:
            kept.append(cov.copy() if rows_only is None else cov[rows_only, :].copy()); kept_steps.append(k)
    out = dict(n_landmarks=np.int64(n_landmarks), m=np.int64(m), lin=lin, ang=ang, idx=idx, zr=zr_eff, zb=zb_eff,
               mean0=mean0, diag0=diag0, out_mean=means, out_diag=np.array(diags), out_fro=np.array(fro),
               out_cov_steps=np.array(kept_steps), out_cov=np.array(kept))
    if rows_only is not None:
        out["out_cov_rows"] = np.asarray(rows_only)
        out["out_cov_rowsum"] = cov.sum(axis=1)
        out["out_cov_colsum"] = cov.sum(axis=0)
    return out


def run_proto(proto, seed, steps):
    rng = np.random.default_rng(seed)
    state = np.array([0.0, 0.0, 0.0]); cov = np.eye(3) * 0.1        # EKF-SLAM.py:8-9
    lms = rng.uniform(-2, 2, (5, 2))
    ctrl = np.zeros((steps, 2)); dts = np.zeros(steps); obs = np.zeros((steps, 2)); lmk = np.zeros((steps, 2))
    s_pred = np.zeros((steps, 3)); p_pred = np.zeros((steps, 3, 3)); s_upd = np.zeros((steps, 3)); p_upd = np.zeros((steps, 3, 3))
    for k in range(steps):
        v = rng.uniform(0.0, 0.5)
        om = 0.0 if k % 5 == 2 else (5e-7 if k % 5 == 4 else rng.uniform(-1.5, 1.5))
        dt = rng.uniform(0.05, 0.8)
        ctrl[k] = (v, om); dts[k] = dt
        state, cov = proto["predict"](state, cov, (v, om), dt)
        s_pred[k] = state; p_pred[k] = cov
        lm = lms[k % 5]
        d = lm - state[0:2]
        z = (np.hypot(*d) + rng.normal(0, 0.05), np.arctan2(d[1], d[0]) - state[2] + rng.normal(0, 0.05) + (TWO_PI if k % 3 == 0 else 0))
        obs[k] = z; lmk[k] = lm
        state, cov = proto["update"](state, cov, z, lm)
        s_upd[k] = state; p_upd[k] = cov
    return dict(control=ctrl, dt=dts, observation=obs, landmark=lmk,
                pred_state=s_pred, pred_cov=p_pred, upd_state=s_upd, upd_cov=p_upd)


TWO_PI = 2 * np.pi


def make_events_csv(seed: int, duration: float, n_tags: int):
    """A synthetic log in the writer's format (scripts/decode_bag_file.py:337-347): wheel ticks at ~30 Hz,
    'image' frames at ~8 Hz, Vicon ground truth, one landmarks line, one camera_intrinsis line.
    Returns (csv text, {frame name: [(tag_id, pose_t(3), pose_err), ...]})."""
    rng = np.random.default_rng(seed)
    lm = np.stack([rng.uniform(-1.0, 1.2, n_tags), rng.uniform(-0.9, 1.1, n_tags)], axis=1)
    tag_ids = rng.permutation(np.arange(10, 10 + 4 * n_tags))[:n_tags]
    t0 = 1700000000.25
    events = [(t0 - 1.0, "camera_intrinsis", repr([(0.1, -0.2, 0.0, 0.0, 0.0), (320.5, 0.0, 331.25, 0.0, 318.75, 247.5, 0.0, 0.0, 1.0),
                                                   (1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0), (320.5, 0.0, 331.25, 0.0)]))]
    events.append((t0 + 0.01, "landmarks", repr([(float(x), float(y)) for x, y in lm])))
    pose = np.zeros(3)
    lt, rt = 3, 0              # the right wheel starts at tick 0: exercises the `== False` re-latch (:114)
    frames = {}
    t = t0
    k = 0
    while t < t0 + duration:
        t += 1.0 / 30.0 + rng.uniform(0, 2e-3)
        v = 0.12 + 0.05 * np.sin(0.7 * (t - t0))
        w = 0.5 * np.sin(0.31 * (t - t0)) + (0.0 if (t - t0) % 9 > 2 else 0.003)
        dt = 1.0 / 30.0
        pose = pose + np.array([v * dt * np.cos(pose[2]), v * dt * np.sin(pose[2]), w * dt])
        dl = (v - w * 0.05) * dt / 0.0318 * 135 / TWO_PI
        dr = (v + w * 0.05) * dt / 0.0318 * 135 / TWO_PI
        lt += int(round(dl + rng.normal(0, 0.2)))
        rt += int(round(dr + rng.normal(0, 0.2)))
        events.append((t, "left_wheel", lt))
        events.append((t + 1e-4, "right_wheel", rt))
        if k % 4 == 0:
            name = "frame%06i.png" % (k // 4)
            tags = []
            for j in rng.permutation(n_tags):
                d = lm[j] - pose[0:2]
                xr = np.cos(pose[2]) * d[0] + np.sin(pose[2]) * d[1]
                yr = -np.sin(pose[2]) * d[0] + np.cos(pose[2]) * d[1]
                if xr > 0.05 and abs(np.arctan2(yr, xr)) < 0.9 and rng.random() < 0.8:
                    tags.append((int(tag_ids[j]), [float(-yr + rng.normal(0, 0.01)), 0.04, float(xr + rng.normal(0, 0.01))],
                                 float(rng.uniform(1e-4, 1e-2))))
            frames[name] = tags
            events.append((t + 2e-4, "image", name))
        if k % 3 == 0:
            events.append((t + 3e-4, "ground_truth", "%r,%r" % (float(pose[0]), float(pose[1]))))
        k += 1
    events.sort(key=lambda e: e[0])
    text = "".join(str(e[0]) + "," + e[1] + "," + (e[2] if isinstance(e[2], str) else repr(e[2])) + "\n" for e in events)
    return text, frames


def events_tag_ids(seed: int, n_tags: int):
    """The tag ids make_events_csv(seed, ., n_tags) gives its landmarks, in landmark order (its first two draws)."""
    rng = np.random.default_rng(seed)
    rng.uniform(-1.0, 1.2, n_tags); rng.uniform(-0.9, 1.1, n_tags)
    return [int(t) for t in rng.permutation(np.arange(10, 10 + 4 * n_tags))[:n_tags]]


def run_reference_replay(slam, text, frames, tmpdir, fast_mode=False, god_key=None):
    """Execute the reference's own replay() (src/replay_no_ros.py:66-248) on the synthetic log.  Its I/O
    boundary is replaced: frames are looked up in `frames` instead of cv2.imread + dt_apriltags
    (:587-597), and plot_path (:499-580) records what it is handed instead of drawing.
    fast_mode: the reference's ENABLE_FAST_MODE (:32, :122-123, :216-227).
    god_key: ENABLE_GOD_EKF = True with GOD_SECRET_KEY = god_key (:23-26, :140-157): at the `landmarks` event the
    state is pre-sized with the true positions at zero variance and TAG_INDEX maps god_key[i] -> i."""
    os.makedirs(tmpdir, exist_ok=True)
    with open(os.path.join(tmpdir, "events.csv"), "w") as fh:
        fh.write(text)
    rec = dict(mean=[], cov=[], path=[], ntags=[], gt=[])

    def fake_detect(img, camera_params):
        rec.setdefault("camera_params", list(camera_params))
        return [SimpleNamespace(tag_id=i, pose_R=np.eye(3), pose_t=np.array(t, dtype=float).reshape(3, 1), pose_err=e,
                                center=np.zeros(2), corners=np.zeros((4, 2))) for i, t, e in frames[os.path.basename(img)]]

    def fake_plot(vertices, gt, landmarks, mean, cov, measured, tag_index):
        rec["mean"].append(np.array(mean)); rec["cov"].append(np.array(cov)); rec["path"].append(tuple(vertices[-1]))
        rec["ntags"].append(len(tag_index)); rec["gt"].append(len(gt)); rec["landmarks"] = landmarks
        rec["tag_index"] = dict(tag_index)

    saved = {k: slam.get(k) for k in ("load_grayscale", "detect_tags", "plot_path", "plt", "visualize_bounding_boxes",
                                      "ENABLE_CAMERA_VISUALIZATION", "os", "image_list", "print", "ENABLE_FAST_MODE",
                                      "ENABLE_GOD_EKF", "GOD_SECRET_KEY")}
    slam.update(load_grayscale=lambda path: path, detect_tags=fake_detect, plot_path=fake_plot,
                plt=SimpleNamespace(pause=lambda *_a: None), visualize_bounding_boxes=lambda *_a: None,
                ENABLE_CAMERA_VISUALIZATION=False, os=os, image_list=[], print=lambda *_a: None,
                ENABLE_FAST_MODE=bool(fast_mode))
    if god_key is not None:
        slam.update(ENABLE_GOD_EKF=True, GOD_SECRET_KEY=list(god_key))
    try:
        slam["replay"](tmpdir)
    finally:
        for k, v in saved.items():
            if v is None:
                slam.pop(k, None)
            else:
                slam[k] = v
    return rec


def pack_replay(rec):
    nmax = max(len(m) for m in rec["mean"])
    W = len(rec["mean"])
    M = np.zeros((W, nmax)); Pm = np.zeros((W, nmax, nmax)); sizes = np.zeros(W, dtype=np.int64)
    for k in range(W):
        n = len(rec["mean"][k]); sizes[k] = n
        M[k, :n] = rec["mean"][k]; Pm[k, :n, :n] = rec["cov"][k]
    return dict(out_mean=M, out_cov=Pm, out_size=sizes, out_path=np.array(rec["path"], dtype=float),
                out_ntags=np.array(rec["ntags"]), out_gt_count=np.array(rec["gt"]),
                out_tag_index=np.array(sorted(rec["tag_index"].items(), key=lambda kv: kv[1]), dtype=np.int64).reshape(-1, 2))


def scaled_ints(rows, width):
    """Decimal fields of a Vicon CSV as int64 of value * 1e12 (missing -> INT64_MIN): the CSV's decimals have at most
    12 places, so int / 1e12 (a correctly rounded division of two exact doubles) gives back exactly the double
    float(text) gives."""
    from decimal import Decimal
    out = np.full((len(rows), width), np.iinfo(np.int64).min, dtype=np.int64)
    for r, row in enumerate(rows):
        for c, field in enumerate(row[:width]):
            if field != "":
                v = Decimal(field) * 1000000000000
                assert v == v.to_integral_value() and abs(int(v)) < 2 ** 53 and float(int(v)) / 1e12 == float(field)
                out[r, c] = int(v)
    return out


def run_reference_vicon(ref_root: str):
    """The reference's own ground-truth decoder, scripts/decode_bag_file.py:107-253 (`rotate_around`,
    `get_ground_truth`), executed on the recorded Vicon data under bags/ (robot track + marker trajectories +
    the .xcp capture times).  The matching .bag is an absent LFS blob, so `first_timestamp` (the first bag event,
    which the decoder only uses to pick the Vicon frame that defines the origin) is set 1.234 s into the capture.
    Only plot_path (:257-, matplotlib) is replaced, by a no-op."""
    import datetime as _dt
    import xml.etree.ElementTree as ET
    path = os.path.join(ref_root, "scripts", "decode_bag_file.py")
    tree = ast.parse(open(path).read())
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ("rotate_around", "get_ground_truth")]
    ns = {"os": os, "plot_path": lambda *_a: None, "print": lambda *_a: None}
    exec(compile(ast.Module(body=keep, type_ignores=[]), "decode_bag_file.py", "exec"), ns)
    prefix = os.path.join(ref_root, "bags", "quackgpt_small_town_joystick")
    cap = ET.parse(prefix + ".xcp").getroot().find("Camera/Capture")
    # the decoder's own expressions (:131-132), evaluated here only to choose first_timestamp and to record what the
    # time axis was in this container (naive local time + 5 h)
    start = _dt.datetime.fromisoformat(cap.get("START_TIME")).timestamp() + _dt.timedelta(hours=5).total_seconds()
    end = _dt.datetime.fromisoformat(cap.get("END_TIME")).timestamp() + _dt.timedelta(hours=5).total_seconds()
    first_timestamp, delay = start + 1.234, 0.0
    events = ns["get_ground_truth"](first_timestamp, prefix, delay)
    assert events[0][1] == "landmarks" and all(e[1] == "ground_truth" for e in events[1:])
    landmarks = np.array(ast.literal_eval(events[0][2]), dtype=float)
    gt_t = np.array([e[0] for e in events[1:]])
    gt_xy = np.array([[float(v) for v in e[2].split(",")] for e in events[1:]])

    def rows_of(path):
        with open(path) as fh:
            for _ in range(5):
                fh.readline()
            return [ln.split(",") for ln in fh.read().strip().splitlines()]
    robot_rows, marker_rows = rows_of(prefix + ".csv"), rows_of(prefix + "_trajectories.csv")
    width = max(len(r) for r in marker_rows)
    return dict(robot_fields=scaled_ints(robot_rows, 8), robot_len=np.array([len(r) for r in robot_rows]),
                marker_fields=scaled_ints(marker_rows, width), marker_len=np.array([len(r) for r in marker_rows]),
                start_capture_time=start, end_capture_time=end, first_timestamp=first_timestamp, delay=delay,
                out_landmarks=landmarks, out_landmarks_time=np.float64(events[0][0]), out_gt_time=gt_t, out_gt_xy=gt_xy)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="", help="comma-separated fixture names to (re)generate; default all")
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(HERE), "tests", "golden"))
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    slam, proto = load_reference(args.ref)

    only = set(filter(None, args.only.split(",")))

    def save(name, **arrs):
        if only and name not in only:
            return
        path = os.path.join(args.out, name + ".npz")
        np.savez_compressed(path, **arrs)
        print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB")

    # 1. whole-function replay scenarios: association, gate, averaging, augmentation, both branches, wraps
    base_flags = dict(ENABLE_MEASUREMENT_MODEL=True, ENABLE_CIRCULAR_INTERPOLATION=True, DISABLE_MOTION_MODEL=False)
    variants = {
        "replay_default": (dict(base_flags), False),
        "replay_bigturns": (dict(base_flags), True),
        "replay_no_measurement": (dict(base_flags, ENABLE_MEASUREMENT_MODEL=False), False),
        "replay_no_motion": (dict(base_flags, DISABLE_MOTION_MODEL=True), False),
        "replay_linear_interp": (dict(base_flags, ENABLE_CIRCULAR_INTERPOLATION=False), True),
    }
    for i, (name, (flags, big)) in enumerate(variants.items()):
        lin, ang, rec = replay_scenario(100 + i, 40, 7, big)
        out = run_replay(slam, lin, ang, rec, flags)
        save(name, lin=lin, ang=ang, det_step=rec["step"], det_frame=rec["frame"], det_tag_id=rec["tag_id"],
             det_pose_t=rec["pose_t"], det_err=rec["err"],
             flag_measurement=np.bool_(flags["ENABLE_MEASUREMENT_MODEL"]),
             flag_circular=np.bool_(flags["ENABLE_CIRCULAR_INTERPOLATION"]),
             flag_no_motion=np.bool_(flags["DISABLE_MOTION_MODEL"]), **out)

    # 1b. IGNORE_TAGS (:36-37, :286): two of the scenario's tags are dropped before the gate, so they never get a
    # landmark index.  The detections themselves (ignored ones included) are the fixture's inputs.
    lin, ang, rec = replay_scenario(105, 40, 7, False)
    seen_ids = list(dict.fromkeys(int(t) for t in rec["tag_id"]))
    ignore = [seen_ids[1], seen_ids[4]]
    out = run_replay(slam, lin, ang, rec, dict(base_flags, IGNORE_TAGS=list(ignore)))
    assert not set(ignore) & {int(t) for t, _ in out["out_tag_index"]} and len(out["out_tag_index"]) == len(seen_ids) - 2
    save("replay_ignore_tags", lin=lin, ang=ang, det_step=rec["step"], det_frame=rec["frame"], det_tag_id=rec["tag_id"],
         det_pose_t=rec["pose_t"], det_err=rec["err"], flag_measurement=np.bool_(True), flag_circular=np.bool_(True),
         flag_no_motion=np.bool_(False), ignore_tags=np.array(ignore, dtype=np.int64), **out)

    # 2. synthetic stream: config 1 (N=20, 500 steps), N=50, N=500 (rows + sums only)
    save("stream_n20_m8", **run_stream(slam, 20, 500, 8, keep_every=25))
    save("stream_n20_m1", **run_stream(slam, 20, 100, 1, keep_every=25))
    save("stream_n50_m8", **run_stream(slam, 50, 30, 8, keep_every=5))
    rows = np.array([0, 1, 2, 3, 4, 17, 18, 101, 500, 501, 777, 1001, 1002])
    save("stream_n500_m8", **run_stream(slam, 500, 6, 8, keep_every=3, rows_only=rows))

    # 2b. the offline replay loop itself (windowing, tick latching, odometry) on a synthetic events.csv
    import tempfile
    text, frames = make_events_csv(5, 40.0, 9)
    with tempfile.TemporaryDirectory() as tmp:
        rec = run_reference_replay(slam, text, frames, tmp)
    nmax = max(len(m) for m in rec["mean"])
    W = len(rec["mean"])
    M = np.zeros((W, nmax)); Pm = np.zeros((W, nmax, nmax)); sizes = np.zeros(W, dtype=np.int64)
    for k in range(W):
        n = len(rec["mean"][k]); sizes[k] = n
        M[k, :n] = rec["mean"][k]; Pm[k, :n, :n] = rec["cov"][k]
    names = sorted(frames)
    det_frame, det_id, det_t, det_err = [], [], [], []
    for fi, name in enumerate(names):
        for tid, t, e in frames[name]:
            det_frame.append(fi); det_id.append(tid); det_t.append(t); det_err.append(e)
    save("replay_events", events_csv=np.array(text), frame_names=np.array(names), det_frame=np.array(det_frame),
         det_tag_id=np.array(det_id), det_pose_t=np.array(det_t, dtype=float).reshape(-1, 3), det_err=np.array(det_err),
         out_mean=M, out_cov=Pm, out_size=sizes, out_path=np.array(rec["path"], dtype=float),
         out_ntags=np.array(rec["ntags"]), out_gt_count=np.array(rec["gt"]),
         out_tag_index=np.array(sorted(rec["tag_index"].items(), key=lambda kv: kv[1]), dtype=np.int64).reshape(-1, 2),
         out_camera_params=np.array(rec["camera_params"], dtype=float),
         out_landmarks=np.array(rec["landmarks"], dtype=float))

    # 2c. the same log with ENABLE_FAST_MODE (:32): detection deferred to the window boundary, last 5 images of the
    # ever-growing image list (:216-227).  Inputs are replay_events.npz's; only the outputs are stored.
    with tempfile.TemporaryDirectory() as tmp:
        rec_fast = run_reference_replay(slam, text, frames, tmp, fast_mode=True)
    save("replay_events_fast", **pack_replay(rec_fast))

    # 2c'. the same log with ENABLE_GOD_EKF (:23-26, :140-157): GOD_SECRET_KEY = the log's tag ids in landmark order.
    god_key = events_tag_ids(5, 9)
    with tempfile.TemporaryDirectory() as tmp:
        rec_god = run_reference_replay(slam, text, frames, tmp, god_key=god_key)
    assert all(len(m) == 3 + 2 * 9 for m in rec_god["mean"]) and rec_god["tag_index"] == {t: i for i, t in enumerate(god_key)}
    save("replay_events_god", god_key=np.array(god_key, dtype=np.int64), **pack_replay(rec_god))

    # 2d. Vicon ground-truth alignment on the recorded data
    if not only or "vicon_alignment" in only:
        save("vicon_alignment", **run_reference_vicon(args.ref))

    # 3. 3-state predict/update prototype
    save("proto3", **run_proto(proto, 7, 60))

    # 4. odometry scalars
    rng = np.random.default_rng(11)
    ticks = rng.integers(-500, 500, (64, 4))
    dphi = np.array([[slam["delta_phi"](int(a), int(b), 135), slam["delta_phi"](int(c), int(d), 135)] for a, b, c, d in ticks])
    disp = np.array([slam["displacement"](0.0318, 0.1, l, r) for l, r in dphi])
    save("odometry", ticks=ticks, resolution=np.int64(135), wheel_radius=0.0318, baseline=0.1, dphi=dphi, disp=disp)


if __name__ == "__main__":
    main()

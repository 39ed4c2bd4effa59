"""ctypes bindings of libekfslam_hip.so -- the module the reference's empty ``src/ekf_bindings.py`` stands for.

Call surface:

* ``EKF_pose_estimation(ang, lin, mean, cov, delta_t, detections, TAG_INDEX)``
  drop-in for ``src/replay_no_ros.py:269-482`` (same arguments, same return triple, ``TAG_INDEX``
  mutated the same way).  Association stays on the host (``frontend.associate``); augmentation,
  prediction and the sequential update run on the GPU.
* ``EkfSlam(n_max, batch=1, ...)`` with ``predict / update / step / run_stream`` keeps the state
  resident in HBM -- this is the fast path.
* ``predict(state, covariance, control, dt)`` / ``update(state, covariance, observation,
  landmark_pos)``: the 3-state prototype surface of ``src/EKF-SLAM.py:29-84``.

There is NO CPU fallback: if the shared library or a gfx950 device is missing, calls raise
``EkfError``.  Host code is NumPy + ctypes only (no PyTorch, no CuPy).
"""
from __future__ import annotations

import collections
import ctypes as C
import dataclasses
import os
import subprocess
import threading
import warnings
import weakref
from typing import Optional, Sequence

import numpy as np

from .frontend import associate

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_NAME = "libekfslam_hip.so"
EKF_MMAX = 16
EKF_FLAG_NONFINITE = 1
EKF_FLAG_ASSOC = 2
EKF_FLAG_INTERNAL = 4             # a bounded wait of a single-launch step timed out: sync()/state()/mean() raise EkfError
EKF_N_MAX_LIMIT = 21823           # largest n_max (one covariance stays below 4 GiB: 32-bit byte offsets in the kernels)
EKF_DMAX = 256                    # detections per window the device front end takes
EKF_AMAX = 32                     # distinct tags per window the device front end takes (two update passes of EKF_MMAX)
EKF_TAGMAX = 1024


class EkfError(RuntimeError):
    """A C-ABI call returned a non-zero status (message from ``ekf_last_error``)."""


class _CConfig(C.Structure):
    _fields_ = [("motion_sigma", C.c_double), ("meas_sigma", C.c_double), ("arc_threshold", C.c_double),
                ("landmark_init_var", C.c_double), ("enable_measurement_model", C.c_int),
                ("enable_circular_interpolation", C.c_int), ("disable_motion_model", C.c_int),
                ("reserved", C.c_int)]


@dataclasses.dataclass
class EkfConfig:
    """The reference's module constants (src/replay_no_ros.py:15-36) as explicit fields."""
    motion_sigma: float = 0.1                  # MOTION_MODEL_VARIANCE
    meas_sigma: float = 0.7                    # MEASUREMENT_MODEL_VARIANCE
    arc_threshold: float = 1e-2                # :376
    landmark_init_var: float = 10000.0         # :356
    enable_measurement_model: bool = True      # ENABLE_MEASUREMENT_MODEL
    enable_circular_interpolation: bool = True  # ENABLE_CIRCULAR_INTERPOLATION
    disable_motion_model: bool = False         # DISABLE_MOTION_MODEL
    gate_range: float = 1.5                    # :289 (host side)
    ignore_tags: tuple = ()                    # IGNORE_TAGS (host side)

    def _c(self) -> _CConfig:
        return _CConfig(self.motion_sigma, self.meas_sigma, self.arc_threshold, self.landmark_init_var,
                        int(self.enable_measurement_model), int(self.enable_circular_interpolation),
                        int(self.disable_motion_model), 0)


def library_path() -> str:
    """The in-tree library.  EKFSLAM_HIP_VARIANT=<tag> selects a diagnostic build libekfslam_hip_<tag>.so made with
    `make -C csrc variant TAG=<tag> EXTRA=...` (kernel experiments; never set in production)."""
    tag = os.environ.get("EKFSLAM_HIP_VARIANT")
    if tag:
        warnings.warn(f"EKFSLAM_HIP_VARIANT={tag}: loading the diagnostic build libekfslam_hip_{tag}.so instead of the "
                      "product library -- such builds may knowingly compute wrong covariances (timing experiments only)",
                      RuntimeWarning, stacklevel=2)
    return os.path.join(_HERE, f"libekfslam_hip_{tag}.so" if tag else _LIB_NAME)


def build_library(force: bool = False) -> str:
    """Compile the HIP sources for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    path = library_path()
    srcdir = os.path.join(_HERE, "csrc")
    if force:
        subprocess.run(["make", "-C", srcdir, "clean"], check=True, stdout=subprocess.DEVNULL)
    subprocess.run(["make", "-C", srcdir, "-j4"], check=True, stdout=subprocess.DEVNULL)
    if not os.path.exists(path):
        raise EkfError("build did not produce " + path)
    return path


_lib = None
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)

# name -> (restype, argtypes): every symbol include/ekfslam_hip.h declares
ABI = {
    "ekf_config_default": (C.c_int, [C.POINTER(_CConfig)]),
    "ekf_device_count": (C.c_int, [_ip]),
    "ekf_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(_CConfig), C.POINTER(C.c_void_p)]),
    "ekf_destroy": (C.c_int, [C.c_void_p]),
    "ekf_upload_state": (C.c_int, [C.c_void_p, C.c_int, _dp, _dp, C.c_int]),
    "ekf_upload_state_diag": (C.c_int, [C.c_void_p, C.c_int, _dp, _dp, C.c_int]),
    "ekf_download_state": (C.c_int, [C.c_void_p, C.c_int, _dp, _dp, C.c_int]),
    "ekf_download_mean": (C.c_int, [C.c_void_p, C.c_int, _dp, C.c_int]),
    "ekf_download_block": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _dp]),
    "ekf_state_size": (C.c_int, [C.c_void_p, C.c_int, _ip]),
    "ekf_add_landmarks": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _dp, C.c_int]),
    "ekf_predict": (C.c_int, [C.c_void_p, _dp, _dp]),
    "ekf_update": (C.c_int, [C.c_void_p, _ip, _dp, _dp, _ip, C.c_int]),
    "ekf_step": (C.c_int, [C.c_void_p, _dp, _dp, _ip, _dp, _dp, _ip, C.c_int]),
    "ekf_host_alloc": (C.c_void_p, [C.c_size_t]),
    "ekf_host_free": (None, [C.c_void_p]),
    "ekf_step_fetch": (C.c_int, [C.c_void_p, _dp, _dp, _ip, _dp, _dp, _ip, C.c_int, C.c_int, _dp, _dp, C.c_int]),
    "ekf_set_association": (C.c_int, [C.c_void_p, C.c_double, _ip, C.c_int]),
    "ekf_step_detections": (C.c_int, [C.c_void_p, _dp, _dp, _ip, _ip, _dp, _dp, C.c_int]),
    "ekf_download_tags": (C.c_int, [C.c_void_p, C.c_int, _ip, _ip, _ip, _dp, _dp, _dp, _dp, _dp]),
    "ekf_download_tag_index": (C.c_int, [C.c_void_p, C.c_int, _ip, C.c_int, _ip]),
    "ekf_upload_tag_index": (C.c_int, [C.c_void_p, C.c_int, _ip, C.c_int]),
    "ekf_stream_upload": (C.c_int, [C.c_void_p, C.c_int, _dp, _dp, _ip, _dp, _dp, _ip, C.c_int]),
    "ekf_stream_run": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "ekf_run_stream": (C.c_int, [C.c_void_p, C.c_int, _dp, _dp, _ip, _dp, _dp, _ip, C.c_int]),
    "ekf_predict_dense": (C.c_int, [C.c_void_p, C.c_int, _dp, _dp]),
    "ekf_flush": (C.c_int, [C.c_void_p]),
    "ekf_sync": (C.c_int, [C.c_void_p]),
    "ekf_status_flags": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_uint)]),
    "ekf_last_error": (C.c_char_p, [C.c_void_p]),
    "ekf_timer_begin": (C.c_int, [C.c_void_p]),
    "ekf_timer_end": (C.c_int, [C.c_void_p, _dp]),
    "ekf_profile_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "ekf_profile_read": (C.c_int, [C.c_void_p, _dp, C.POINTER(C.c_longlong)]),
    "ekf_profile_passes": (C.c_longlong, [C.c_void_p]),
    "ekf_profile_read_class": (C.c_int, [C.c_void_p, C.c_int, _dp, C.POINTER(C.c_longlong)]),
    "ekf_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "ekf_last_pass": (C.c_int, [C.c_void_p, _ip, _ip, _ip]),
    # diagnostics (the header's last section)
    "ekf_debug_cadences": (C.c_int, [C.c_void_p, C.POINTER(C.c_long), C.POINTER(C.c_long)]),
    "ekf_debug_lookaheads": (C.c_long, [C.c_void_p]),
    "ekf_debug_chained": (C.c_long, [C.c_void_p]),
    "ekf_debug_assoc_fallbacks": (C.c_long, [C.c_void_p]),
    "ekf_debug_w_from_v": (C.c_long, [C.c_void_p]),
    "ekf_debug_last_pass_wv": (C.c_int, [C.c_void_p]),
    "ekf_debug_note_assoc_fallback": (None, [C.c_void_p]),
    "ekf_debug_last_pass_shares": (C.c_int, [C.c_void_p]),
    "ekf_debug_small_launches": (C.c_long, [C.c_void_p]),
    "ekf_debug_fused_fetches": (C.c_long, [C.c_void_p]),
    "ekf_debug_dense_packs": (C.c_long, [C.c_void_p]),
    "ekf_debug_fetch_retries": (C.c_long, [C.c_void_p]),
    "ekf_debug_cad": (C.c_long, [C.c_void_p, C.c_int, C.c_void_p, C.c_long]),
    "ekf_debug_snapshot": (C.c_long, [C.c_void_p, C.c_int, C.c_int, _dp, C.c_long]),
    "ekf_debug_read": (C.c_int, [C.c_void_p, C.c_void_p, C.c_long]),
    "ekf_debug_pass_units": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, _ip, C.c_int]),
    "ekf_debug_pass_shares": (C.c_int, [C.c_int, C.c_int, C.c_int, _ip]),
}


def load_library():
    """dlopen the in-tree library and type every entry point.  Raises EkfError if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise EkfError(f"{path} not found: build it with __graft_entry__.build() or `make -C "
                       f"{os.path.join(_HERE, 'csrc')}` -- there is no CPU fallback")
    try:
        lib = C.CDLL(path)
    except OSError as e:
        raise EkfError(f"cannot load {path}: {e}") from e
    for name, (res, args) in ABI.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def device_count() -> int:
    """Number of HIP devices visible to this process (0 without a GPU)."""
    n = C.c_int(0)
    load_library().ekf_device_count(C.byref(n))
    return n.value


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and a.shape != tuple(shape):
        raise ValueError(f"expected shape {shape}, got {a.shape}")
    return a


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a, typ=_dp):
    return a.ctypes.data_as(typ)


_SMALL_OUT_N = 131        # (PACK_SMALL_N of csrc/ekf_api.hip: states the library hands over through one pinned buffer)


class _PinnedPool:
    """Recycled pinned host buffers behind the LARGE arrays handed to the caller (covariances beyond 131 x 131).

    The reference's loop gets a fresh covariance from every call (src/replay_no_ros.py:229-237).  A fresh ``np.empty`` of
    128 MB (N = 2000) is 32 768 untouched pages: the download faults them in and pins them (5 ms on top of 2.3 ms of PCIe
    time) and dropping the previous array unmaps as many (another 5 ms).  Here the array is a view of a buffer from
    ``ekf_host_alloc``; the buffer comes back to the pool when the LAST view of it is gone (the ctypes object every view's
    ``base`` chain ends in is what the finaliser watches), and is handed out again for the next array of that size.
    Beyond ``LIMIT`` bytes of live buffers -- a caller that keeps every covariance -- arrays are ordinary ``np.empty``."""
    LIMIT = 2 << 30
    KEEP = 2                      # free buffers kept per size (the loop holds one array while the next is filled)

    def __init__(self):
        self.free = collections.OrderedDict()      # size -> free buffers of that size, least recently used size first
        self.live = 0                              # bytes allocated (handed out + free)
        # handles may be driven from several host threads (one each); finalisers run wherever the last reference dies,
        # possibly inside empty() of the same thread: a re-entrant lock
        self.lock = threading.RLock()

    def _evict(self, need):
        """(under the lock) Free buffers, least recently used size first, until `need` more bytes fit LIMIT; returns the
        pointers to free -- outside the lock: hipHostFree synchronises the device."""
        out = []
        for size in list(self.free):
            while self.free[size] and self.live + need > self.LIMIT:
                out.append(self.free[size].pop())
                self.live -= size
            if not self.free[size]:
                del self.free[size]
            if self.live + need <= self.LIMIT:
                break
        return out

    def empty(self, lib, shape):
        count = int(np.prod(shape))
        nbytes = 8 * count
        ptr, evicted = None, []
        with self.lock:
            kept = self.free.get(nbytes)
            if kept:
                ptr = kept.pop()
                self.free.move_to_end(nbytes)
            else:
                # buffers of OTHER sizes stay (two handles of different size, a map that grows a landmark at a time): they
                # go only when this allocation would not fit LIMIT otherwise
                if self.live + nbytes > self.LIMIT:
                    evicted = self._evict(nbytes)
                if self.live + nbytes <= self.LIMIT:
                    self.live += nbytes                # (reserved; the allocation itself happens outside the lock)
                    ptr = -1
        for q in evicted:
            lib.ekf_host_free(q)
        if ptr == -1:
            ptr = lib.ekf_host_alloc(nbytes)
            if not ptr:
                with self.lock:
                    self.live -= nbytes
        if not ptr:
            return np.empty(shape)
        buf = (C.c_double * count).from_address(ptr)
        weakref.finalize(buf, self._release, lib, ptr, nbytes).atexit = False
        return np.frombuffer(buf, dtype=np.float64).reshape(shape)

    def _release(self, lib, ptr, nbytes):
        drop = False
        with self.lock:
            kept = self.free.setdefault(nbytes, [])
            self.free.move_to_end(nbytes)
            if len(kept) < self.KEEP:
                kept.append(ptr)
            else:
                drop = True
                self.live -= nbytes
        if drop:
            lib.ekf_host_free(ptr)


_pinned = _PinnedPool()


class EkfSlam:
    """A bank of ``batch`` independent EKF-SLAM filters resident on one MI355X.

    State per trajectory: ``mu`` (3+2N,) and ``P`` (3+2N, 3+2N) float64 in HBM.  All stepping calls
    are asynchronous on the handle's stream; ``mean()/covariance()/sync()`` block.
    """

    def __init__(self, n_max: int, batch: int = 1, device: int = 0, config: Optional[EkfConfig] = None):
        self._lib = load_library()
        self.config = config or EkfConfig()
        self.n_max = int(n_max)
        self.batch = int(batch)
        self._h = C.c_void_p()
        cc = self.config._c()
        rc = self._lib.ekf_create(int(device), self.n_max, self.batch, C.byref(cc), C.byref(self._h))
        if rc != 0:
            msg = self._lib.ekf_last_error(None).decode()
            self._h = C.c_void_p()
            raise EkfError(f"ekf_create failed ({rc}): {msg}")
        # windows beyond the device front end's limits are associated on the host (step_detections): per trajectory, the
        # complete TAG_INDEX while it holds ids the device table cannot (>= 1024), and the last host-made tags_positions
        self._assoc_gate, self._assoc_ignore = self.config.gate_range, tuple(self.config.ignore_tags)
        self._host_index = {}
        self._host_tags = {}
        # per-call staging arrays and their ctypes pointers, made once: `ndarray.ctypes.data_as` costs ~2 us per argument,
        # seven of them were most of what an online step of a small filter cost on the host (the library copies everything
        # it needs out of these arrays before the call returns)
        self._lin, self._ang = np.zeros(self.batch), np.zeros(self.batch)
        self._m = np.zeros(self.batch, dtype=np.int32)
        self._plin, self._pang, self._pm = _p(self._lin), _p(self._ang), _p(self._m, _ip)
        self._stages = {}
        self._out = None

    # -- plumbing ------------------------------------------------------------------------------
    def _check(self, rc: int):
        if rc != 0:
            raise EkfError(f"status {rc}: {self._lib.ekf_last_error(self._h).decode()}")

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.ekf_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _per_traj(self, x, name, out):
        """One value per trajectory (a scalar or a one-element array is repeated) into the staging array `out`."""
        if np.ndim(x) > 1 or (np.ndim(x) == 1 and len(x) not in (1, self.batch)):
            raise ValueError(f"{name}: expected {self.batch} values")
        out[:] = x

    def _obs(self, idx, ranges, bearings):
        """Observations into the padded [batch, stride] staging arrays + m[batch]; returns their pointers and the stride
        (entries beyond m[b] are whatever an earlier call left there: the library reads m[b] of them)."""
        if isinstance(idx, np.ndarray) and idx.ndim == 2:
            # a whole bank at once, the same number of observations for every trajectory: three block copies instead of three
            # per trajectory (256 trajectories: 160 -> 10 us of host time per call)
            if idx.shape[0] != self.batch or np.shape(ranges) != idx.shape or np.shape(bearings) != idx.shape:
                raise ValueError("observations: [batch, m] arrays of equal shape expected")
            m = idx.shape[1]
            st = self._stage(max(1, m))
            st[0][:, :m], st[1][:, :m], st[2][:, :m] = idx, ranges, bearings
            self._m[:] = m
            return st[3], st[4], st[5], self._pm, max(1, m)
        if self.batch == 1 and (len(idx) == 0 or np.ndim(idx[0]) == 0):
            idx, ranges, bearings = (idx,), (ranges,), (bearings,)
        if len(idx) != self.batch:
            raise ValueError("observations: one list per trajectory expected")
        lens = [len(i) for i in idx]
        stride = max(1, max(lens))
        I, R, B, pI, pR, pB = self._stage(stride)
        for b, mb in enumerate(lens):
            if len(ranges[b]) != mb or len(bearings[b]) != mb:
                raise ValueError("idx / ranges / bearings lengths differ")
            I[b, :mb] = idx[b]
            R[b, :mb] = ranges[b]
            B[b, :mb] = bearings[b]
        self._m[:] = lens
        return pI, pR, pB, self._pm, stride

    def _stage(self, stride):
        st = self._stages.get(stride)
        if st is None:
            I = np.zeros((self.batch, stride), dtype=np.int32)
            R, B = np.zeros((self.batch, stride)), np.zeros((self.batch, stride))
            st = self._stages[stride] = (I, R, B, _p(I, _ip), _p(R), _p(B))
        return st

    def _small_out(self):
        """Output staging for small states (what the library packs into one pinned buffer anyway): pointers made once."""
        if self._out is None:
            mu, P = np.empty(_SMALL_OUT_N), np.empty(_SMALL_OUT_N * _SMALL_OUT_N)
            self._out = (mu, P, _p(mu), _p(P))
        return self._out

    # -- state ---------------------------------------------------------------------------------
    def size(self, b: int = 0) -> int:
        n = C.c_int()
        self._check(self._lib.ekf_state_size(self._h, b, C.byref(n)))
        return n.value

    def set_state(self, mean, cov, b: int = 0):
        """Upload mean and covariance of trajectory b (the upper triangle of `cov` is authoritative: the device
        stores a covariance as its upper triangle and returns it mirrored)."""
        mean = _f64(mean)
        n = mean.shape[0]
        cov = _f64(cov, (n, n))
        self._check(self._lib.ekf_upload_state(self._h, b, _p(mean), _p(cov), n))

    def set_state_diag(self, mean, diag, b: int = 0):
        """P = diag(diag): avoids shipping a dense n x n host matrix for block-diagonal starts."""
        mean = _f64(mean)
        n = mean.shape[0]
        diag = _f64(diag, (n,))
        self._check(self._lib.ekf_upload_state_diag(self._h, b, _p(mean), _p(diag), n))

    def mean(self, b: int = 0) -> np.ndarray:
        n = self.size(b)
        out = np.empty(n)
        self._check(self._lib.ekf_download_mean(self._h, b, _p(out), n))
        return out

    def covariance(self, b: int = 0) -> np.ndarray:
        n = self.size(b)
        out = _pinned.empty(self._lib, (n, n)) if n > _SMALL_OUT_N else np.empty((n, n))
        self._check(self._lib.ekf_download_state(self._h, b, None, _p(out), n))
        return out

    def covariance_block(self, r0: int, c0: int, rows: int, cols: int, b: int = 0) -> np.ndarray:
        """P[r0:r0+rows, c0:c0+cols] only (e.g. the 3x3 pose block) -- O(rows*cols) over PCIe."""
        out = np.empty((rows, cols))
        self._check(self._lib.ekf_download_block(self._h, b, r0, c0, rows, cols, _p(out)))
        return out

    def state(self, b: int = 0):
        n = self.size(b)
        if n <= _SMALL_OUT_N:
            mu, P, pmu, pP = self._small_out()
            self._check(self._lib.ekf_download_state(self._h, b, pmu, pP, n))
            return mu[:n].copy(), P[:n * n].reshape(n, n).copy()
        mu, P = np.empty(n), _pinned.empty(self._lib, (n, n))
        self._check(self._lib.ekf_download_state(self._h, b, _p(mu), _p(P), n))
        return mu, P

    def add_landmarks(self, xy, b: int = 0):
        """Append landmarks with world guesses xy (k,2): variance landmark_init_var, zero cross terms."""
        xy = _f64(xy).reshape(-1, 2)
        first = (self.size(b) - 3) // 2
        self._check(self._lib.ekf_add_landmarks(self._h, b, first, _p(xy), xy.shape[0]))

    def flags(self, b: int = 0) -> int:
        f = C.c_uint()
        self._check(self._lib.ekf_status_flags(self._h, b, C.byref(f)))
        return f.value

    def flush(self):
        """Apply the pending low-rank covariance update to P now (asynchronous)."""
        self._check(self._lib.ekf_flush(self._h))

    def sync(self):
        self._check(self._lib.ekf_sync(self._h))

    # -- the EKF ---------------------------------------------------------------------------------
    def predict(self, lin, ang):
        """Motion model + P <- G_F P G_F^T + F^T R F  (src/replay_no_ros.py:368-430)."""
        self._per_traj(lin, "lin", self._lin)
        self._per_traj(ang, "ang", self._ang)
        self._check(self._lib.ekf_predict(self._h, self._plin, self._pang))

    def update(self, idx, ranges, bearings):
        """Sequential range/bearing updates in the given order (src/replay_no_ros.py:436-480)."""
        pI, pR, pB, pm, stride = self._obs(idx, ranges, bearings)
        self._check(self._lib.ekf_update(self._h, pI, pR, pB, pm, stride))

    def step(self, lin, ang, idx, ranges, bearings):
        """predict + update in one fused pass over P.  Observations: a list per trajectory (a flat list for a single
        trajectory), or -- a whole bank with the same number of observations each -- [batch, m] arrays."""
        self._per_traj(lin, "lin", self._lin)
        self._per_traj(ang, "ang", self._ang)
        pI, pR, pB, pm, stride = self._obs(idx, ranges, bearings)
        self._check(self._lib.ekf_step(self._h, self._plin, self._pang, pI, pR, pB, pm, stride))

    def step_state(self, lin, ang, idx, ranges, bearings, b: int = 0):
        """``step`` followed by ``state(b)`` as one library call (``ekf_step_fetch``): one iteration of the reference's
        loop, which gets mean and covariance back from every call (src/replay_no_ros.py:229-237).  On the small-state
        path that is one kernel launch and one synchronisation."""
        self._per_traj(lin, "lin", self._lin)
        self._per_traj(ang, "ang", self._ang)
        pI, pR, pB, pm, stride = self._obs(idx, ranges, bearings)
        n = self.size(b)
        if n <= _SMALL_OUT_N:
            mu, P, pmu, pP = self._small_out()
            self._check(self._lib.ekf_step_fetch(self._h, self._plin, self._pang, pI, pR, pB, pm, stride, b, pmu, pP, n))
            return mu[:n].copy(), P[:n * n].reshape(n, n).copy()
        mu, P = np.empty(n), _pinned.empty(self._lib, (n, n))
        self._check(self._lib.ekf_step_fetch(self._h, self._plin, self._pang, pI, pR, pB, pm, stride, b, _p(mu), _p(P), n))
        return mu, P

    # -- device-side front end (association, gate, averaging, augmentation on the GPU) -----------------
    def set_association(self, gate_range: float = 1.5, ignore_tags: Sequence[int] = ()):
        ig = _i32(list(ignore_tags)) if len(ignore_tags) else None
        self._check(self._lib.ekf_set_association(self._h, float(gate_range), _p(ig, _ip) if ig is not None else None,
                                                  len(ignore_tags)))
        self._assoc_gate, self._assoc_ignore = float(gate_range), tuple(int(t) for t in ignore_tags)

    def _window_fits_device(self, win, index_len_hint=None) -> bool:
        """Whether one trajectory's window stays inside the device front end's limits: at most EKF_DMAX = 256 detections, at
        most EKF_AMAX = 32 distinct tags behind IGNORE_TAGS and the gate (src/replay_no_ros.py:286-289), tag ids in [0, 1024)."""
        ids, count, gate2 = [], 0, self._assoc_gate ** 2
        for _stamp, tags in win:
            for tag in tags:
                count += 1
                t = np.asarray(tag.pose_t, dtype=np.float64).ravel()
                if tag.tag_id in self._assoc_ignore or t[2] ** 2 + t[0] ** 2 > gate2:
                    continue
                if tag.tag_id not in ids:
                    ids.append(tag.tag_id)
        return count <= EKF_DMAX and len(ids) <= EKF_AMAX and all(0 <= int(t) < EKF_TAGMAX for t in ids)

    def step_detections(self, lin, ang, detections):
        """One window of raw detections per trajectory: the reference's ``[(timestamp, [tag, ...])]`` list
        (src/replay_no_ros.py:280-284), or a list of such lists for a batch.  Association, the 1.5 m gate,
        per-tag averaging, augmentation, prediction and update all run on the GPU.

        The device front end has limits the reference's dictionaries do not (:280-301): tag ids in [0, 1024), at most 256
        detections and 32 distinct tags per window (round 6; until round 5: 64 and 16 -- the reference's normal mode, a 0.7 s
        window of every camera frame, :17, did not fit).  Every window is checked against them on the host first; when a
        trajectory's window does not fit -- or its TAG_INDEX already holds an id the device table cannot -- the CALL takes
        the host association (``frontend.associate``: no limits), augmentation through ``add_landmarks`` and the update
        through ``step`` (which splits lists of more than 16 landmarks), and the device's table follows the host's
        afterwards where its ids allow.  Same results either way (tests/test_gpu_api_regressions.py).  Only the capacity
        stays fixed: a map that would outgrow ``n_max`` raises ``EkfError`` before anything is enqueued."""
        self._per_traj(lin, "lin", self._lin)
        self._per_traj(ang, "ang", self._ang)
        lin, ang = self._lin, self._ang
        if self.batch == 1 and (len(detections) == 0 or isinstance(detections[0], tuple)):
            detections = [detections]
        if len(detections) != self.batch:
            raise ValueError("detections: one window per trajectory expected")
        if self._host_index or not all(self._window_fits_device(win) for win in detections):
            self._lib.ekf_debug_note_assoc_fallback(self._h)       # (counted: ekf_debug_assoc_fallbacks / assoc_fallbacks())
            return self._step_detections_host(lin, ang, detections)
        self._host_tags.clear()
        flat = [[tag for _stamp, tags in win for tag in tags] for win in detections]
        count = np.array([len(w) for w in flat], dtype=np.int32)
        stride = max(1, int(count.max()))
        ids = np.zeros((self.batch, stride), dtype=np.int32)
        pt = np.zeros((self.batch, stride, 3))
        pe = np.zeros((self.batch, stride))
        for b, w in enumerate(flat):
            for i, tag in enumerate(w):
                ids[b, i] = tag.tag_id
                pt[b, i] = np.asarray(tag.pose_t, dtype=np.float64).ravel()[:3]
                pe[b, i] = tag.pose_err
        self._check(self._lib.ekf_step_detections(self._h, self._plin, self._pang, _p(count, _ip), _p(ids, _ip), _p(pt), _p(pe),
                                                  stride))

    def _step_detections_host(self, lin, ang, detections):
        """The same window through the host association (see step_detections): what the reference does at :280-360 with
        its unbounded dictionaries, then the device's augmentation and update."""
        indices, tags_all = [], []
        for b, win in enumerate(detections):                       # everything is checked before anything is enqueued
            index = dict(self._host_index[b]) if b in self._host_index else self.tag_index(b)
            n_old = self.size(b)
            if 3 + 2 * len(index) != n_old:
                # landmarks added without a tag (add_landmarks / set_state): give them ids no detector produces
                for j in range((n_old - 3) // 2):
                    if j not in index.values():
                        index[-1 - j] = j
            tags = associate(win, index, self.mean(b)[:3], self._assoc_gate, self._assoc_ignore)
            if 3 + 2 * len(index) > self.n_max:
                raise EkfError(f"step_detections: trajectory {b}'s map would grow to {len(index)} landmarks, beyond this "
                               f"handle's capacity n_max = {self.n_max}")
            indices.append(index)
            tags_all.append(tags)
        for b, (index, tags) in enumerate(zip(indices, tags_all)):
            n_old, n_new = self.size(b), 3 + 2 * len(index)
            new_xy = [(tags[(i - 3) // 2][0], tags[(i - 3) // 2][1]) for i in range(n_old, n_new, 2)]   # (:355-360)
            if new_xy:
                self.add_landmarks(np.array(new_xy), b)
        idx = [list(t.keys()) for t in tags_all]
        self.step(lin, ang, idx, [[t[k][4] for k in ks] for t, ks in zip(tags_all, idx)],
                  [[t[k][5] for k in ks] for t, ks in zip(tags_all, idx)])
        for b, (index, tags) in enumerate(zip(indices, tags_all)):
            self._host_tags[b] = tags
            real = {t: j for t, j in index.items() if t >= 0}
            if len(real) == len(index) and all(t < EKF_TAGMAX for t in real):
                self.set_tag_index(real, b)                        # the device table can hold it: it follows the host's
                self._host_index.pop(b, None)
            else:
                self._host_index[b] = index                        # ids beyond the device table: the host stays in charge

    def assoc_fallbacks(self) -> int:
        """Windows `step_detections` took through the host association because they did not fit the device front end."""
        return int(self._lib.ekf_debug_assoc_fallbacks(self._h))

    def tags_positions(self, b: int = 0) -> dict:
        """What EKF_pose_estimation returns as its third value for the last window (:331-337), update order."""
        if b in self._host_tags:                                   # the last window was associated on the host
            return dict(self._host_tags[b])
        m = C.c_int()
        idx, tid = np.zeros(EKF_AMAX, dtype=np.int32), np.zeros(EKF_AMAX, dtype=np.int32)
        arrs = [np.zeros(EKF_AMAX) for _ in range(5)]
        self._check(self._lib.ekf_download_tags(self._h, b, C.byref(m), _p(idx, _ip), _p(tid, _ip), *[_p(a) for a in arrs]))
        if self.flags(b) & EKF_FLAG_ASSOC:
            raise EkfError("device-side association dropped a detection (tag id outside [0, 1024), more than "
                           f"{EKF_AMAX} distinct tags in one window, or the state is full): the map no longer matches "
                           "the reference's; use a larger n_max or the host association")
        xw, yw, err, rng, brg = arrs
        return {int(idx[i]): [xw[i], yw[i], err[i], int(tid[i]), rng[i], brg[i]] for i in range(m.value)}

    def tag_index(self, b: int = 0) -> dict:
        """TAG_INDEX of trajectory b: {tag_id: landmark index} (:294-295) -- the device's table, or the host's while it
        holds ids the device table cannot (see step_detections)."""
        if b in self._host_index:
            return {t: j for t, j in self._host_index[b].items() if t >= 0}
        cap = max(1, (self.n_max - 3) // 2)
        tags = np.full(cap, -1, dtype=np.int32)
        cnt = C.c_int()
        self._check(self._lib.ekf_download_tag_index(self._h, b, _p(tags, _ip), cap, C.byref(cnt)))
        return {int(tags[i]): i for i in range(cnt.value)}

    def set_tag_index(self, tag_index: dict, b: int = 0):
        order = [t for t, _ in sorted(tag_index.items(), key=lambda kv: kv[1])]
        if sorted(tag_index.values()) != list(range(len(order))):
            raise ValueError("TAG_INDEX values must be 0..len-1")
        arr = _i32(order) if order else None
        self._check(self._lib.ekf_upload_tag_index(self._h, b, _p(arr, _ip) if arr is not None else None, len(order)))

    def stream_upload(self, lin, ang, idx, ranges, bearings, m=None) -> int:
        """Copy a whole input stream into HBM (blocking).  Returns the number of steps.

        lin, ang: [steps, batch]; idx, ranges, bearings: [steps, batch, stride]; m: [steps, batch]
        (default: every step observes ``stride`` landmarks).
        """
        lin = _f64(lin).reshape(-1, self.batch)
        steps = lin.shape[0]
        ang = _f64(ang).reshape(steps, self.batch)
        idx = _i32(idx).reshape(steps, self.batch, -1)
        stride = idx.shape[2]
        ranges = _f64(ranges).reshape(steps, self.batch, stride)
        bearings = _f64(bearings).reshape(steps, self.batch, stride)
        m = np.full((steps, self.batch), stride, dtype=np.int32) if m is None else _i32(m).reshape(steps, self.batch)
        self._check(self._lib.ekf_stream_upload(self._h, steps, _p(lin), _p(ang), _p(idx, _ip), _p(ranges),
                                                _p(bearings), _p(m, _ip), stride))
        return steps

    def stream_run(self, first: int, count: int):
        """Enqueue steps [first, first+count) of the uploaded stream (asynchronous)."""
        self._check(self._lib.ekf_stream_run(self._h, int(first), int(count)))

    def run_stream(self, lin, ang, idx, ranges, bearings, m=None):
        """stream_upload + stream_run of every step."""
        steps = self.stream_upload(lin, ang, idx, ranges, bearings, m)
        self.stream_run(0, steps)

    def predict_dense(self, F, Q, b: int = 0):
        """P <- F P F^T + Q for a general dense Jacobian (fp64 MFMA GEMMs)."""
        n = self.size(b)
        F, Q = _f64(F, (n, n)), _f64(Q, (n, n))
        self._check(self._lib.ekf_predict_dense(self._h, b, _p(F), _p(Q)))

    # -- measurement hooks ---------------------------------------------------------------------
    def timer_begin(self):
        self._check(self._lib.ekf_timer_begin(self._h))

    def timer_end(self) -> float:
        ms = C.c_double()
        self._check(self._lib.ekf_timer_end(self._h, C.byref(ms)))
        return ms.value

    def profile_enable(self, on: bool = True):
        self._check(self._lib.ekf_profile_enable(self._h, int(on)))

    def profile_read(self):
        """(total ms, number) of the launches of the covariance pass that carried an event pair since the last read."""
        ms, cnt = C.c_double(), C.c_longlong()
        self._check(self._lib.ekf_profile_read(self._h, C.byref(ms), C.byref(cnt)))
        return ms.value, cnt.value

    def profile_read_class(self, cls: int):
        """(total ms, number) of the bracketed launches of class `cls` (1 solve, 2 chain / gather, 3 panel; needs the option
        "profile_kernels"); read before `profile_read`, which resets."""
        ms, cnt = C.c_double(), C.c_longlong()
        self._check(self._lib.ekf_profile_read_class(self._h, int(cls), C.byref(ms), C.byref(cnt)))
        return ms.value, cnt.value

    def profile_passes(self) -> int:
        """All launches of the covariance pass since profiling was enabled (with "profile_stride" = k every k-th is timed)."""
        return int(self._lib.ekf_profile_passes(self._h))

    def last_pass(self) -> str:
        """Name of the kernel the last covariance pass launched (e.g. ``ekf::k_flush_rs<20, true, false, true>``: k-tiles,
        nontemporal, column-panel layout, W fragments formed from V -- as rocprofv3 prints it), '' if none yet."""
        k, t, st = C.c_int(), C.c_int(), C.c_int()
        self._check(self._lib.ekf_last_pass(self._h, C.byref(k), C.byref(t), C.byref(st)))
        if k.value < 0:
            return ""
        nt = "true" if st.value else "false"
        tiles = next(x for x in (4, 8, 12, 16, 20) if x >= t.value)
        if k.value == 2:
            wv = "true" if self._lib.ekf_debug_last_pass_wv(self._h) == 1 else "false"
            return f"ekf::k_flush_rs<{tiles}, {nt}, {'true' if self.n_max > 4096 else 'false'}, {wv}>"
        regs = {4: (4, 0), 8: (8, 0), 12: (12, 0), 16: (16, 0), 20: (15, 5)}[tiles]
        return f"ekf::k_flush<{regs[0]}, {regs[1]}, {nt}>"

    def cadence_counters(self):
        """(fused cadences launched by stream_run so far, steps of the uploaded stream they covered): which path ran."""
        a, b = C.c_long(), C.c_long()
        self._check(self._lib.ekf_debug_cadences(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def set_option(self, name: str, value: int):
        self._check(self._lib.ekf_set_option(self._h, name.encode(), int(value)))


# ------------------------------------------------------------------------------------------------
# Drop-in for the reference function
# ------------------------------------------------------------------------------------------------
class _DropInState:
    """Device state kept between EKF_pose_estimation calls so that a SMALL state need not be uploaded again when the
    caller passes back exactly what the previous call returned (the reference loop does, src/replay_no_ros.py:229-237).
    "Exactly" is checked against PRIVATE byte-for-byte records of what was returned, never against the returned arrays
    themselves: the caller owns those and may edit them in place.  Beyond 131 x 131 nothing is remembered and every call
    uploads: proving that a large covariance is unchanged means reading all of it (row and column sums of the 128 MB at
    N = 2000 cost 19 ms per call), uploading it takes 2.3 ms."""
    filt: Optional[EkfSlam] = None
    mean_obj: Optional[np.ndarray] = None      # the objects handed to the caller (identity test only)
    cov_obj: Optional[np.ndarray] = None
    mean_copy: Optional[bytes] = None          # private records of their contents (small states only)
    cov_copy: Optional[bytes] = None


_drop = _DropInState()
DROP_IN_CONFIG = EkfConfig()      # edit like the reference's module constants
DROP_IN_ALWAYS_UPLOAD = False     # True: never trust the records, upload mean and covariance every call
# Opt-in residency for LARGE states.  False (default): beyond 131 x 131 every call uploads the upper
# triangle (1.4 ms of the 3.8 ms per call at N = 2000).  True: a call that receives the very `mean` / `covariance` OBJECTS
# the previous call returned -- what the reference's loop passes back, src/replay_no_ros.py:229-237 -- skips the upload at
# any size, on the identity of the objects alone: an IN-PLACE edit of the returned arrays between two calls is then NOT
# seen (their bytes are not compared: reading 128 MB costs more than uploading them).  Small states keep the byte-for-byte
# check either way.
DROP_IN_TRUST_IDENTITY = False
DROP_IN_MIN_CAPACITY = 79         # n_max of the first handle (38 landmarks: the small-state path, P resident in LDS); grows by doubling


def EKF_pose_estimation(angular_displacement, linear_displacement, motion_model_mean, motion_model_covariance,
                        delta_t, timestamp_detectedTags_pair_list, TAG_INDEX):
    """Drop-in for ``EKF_pose_estimation`` (src/replay_no_ros.py:269-482).

    Same arguments (``delta_t`` accepted and unused, like :274), same return triple
    ``(mean, covariance, tags_positions)``; ``TAG_INDEX`` is mutated in place (:295).
    Raises ``KeyError`` like :359 when a newly indexed tag has no measurement this window.
    """
    cfg = DROP_IN_CONFIG
    mean_in = np.asarray(motion_model_mean, dtype=np.float64)
    cov_in = np.asarray(motion_model_covariance, dtype=np.float64)
    tags_positions = associate(timestamp_detectedTags_pair_list, TAG_INDEX, mean_in,
                               cfg.gate_range, cfg.ignore_tags)
    n_old = mean_in.shape[0]
    n_new = max(n_old, 3 + 2 * len(TAG_INDEX))
    new_xy = []
    for i in range(n_old, n_new, 2):                      # :355-360
        j = (i - 3) // 2
        new_xy.append((tags_positions[j][0], tags_positions[j][1]))   # KeyError like the reference

    d = _drop
    small = n_old <= _SMALL_OUT_N
    same_objects = (not DROP_IN_ALWAYS_UPLOAD and d.filt is not None and d.mean_obj is motion_model_mean
                    and d.cov_obj is motion_model_covariance and d.filt.size() == n_old and cov_in.shape == (n_old, n_old))
    resident = same_objects and ((small and d.mean_copy == mean_in.tobytes() and d.cov_copy == cov_in.tobytes())
                                 or (not small and DROP_IN_TRUST_IDENTITY))
    if d.filt is None or d.filt.n_max < n_new or d.filt.config != cfg:
        cap = max(DROP_IN_MIN_CAPACITY, n_new if d.filt is None else max(n_new, 2 * d.filt.n_max - 3))
        cap |= 1
        if d.filt is not None:
            d.filt.close()
        d.filt = EkfSlam(cap, 1, 0, dataclasses.replace(cfg))
        resident = False
    f = d.filt
    if not resident:
        f.set_state(mean_in, cov_in)
    if new_xy:
        f.add_landmarks(np.array(new_xy))
    idx = list(tags_positions.keys())
    mean, cov = f.step_state(linear_displacement, angular_displacement, idx,
                             [tags_positions[k][4] for k in idx], [tags_positions[k][5] for k in idx])
    # (q == 0 or a singular S leave NaN in the mean, like NumPy's 0/0 at :466-469 -- the sticky device flag says the same,
    #  but asking for it costs a round trip per call)
    if not np.isfinite(mean).all():
        warnings.warn("EKF_pose_estimation: non-finite state (q == 0 or singular S)", RuntimeWarning)
    d.mean_obj, d.cov_obj = mean, cov
    d.mean_copy, d.cov_copy = (mean.tobytes(), cov.tobytes()) if len(mean) <= _SMALL_OUT_N else (None, None)
    return mean, cov, tags_positions


# ------------------------------------------------------------------------------------------------
# 3-state prototype surface (src/EKF-SLAM.py:29-84)
# ------------------------------------------------------------------------------------------------
motion_noise = np.diag([0.1, 0.1, np.radians(5)])      # EKF-SLAM.py:12
observation_noise = np.diag([0.5, 0.5])                # EKF-SLAM.py:13
_proto = {}


def _proto_filter(kind: str) -> EkfSlam:
    if kind not in _proto:
        cfg = EkfConfig(meas_sigma=float(np.sqrt(observation_noise[0, 0])))
        _proto[kind] = EkfSlam(3 if kind == "predict" else 5, 1, 0, cfg)
    return _proto[kind]


def predict(state, covariance, control, dt):
    """3-state EKF prediction, ``src/EKF-SLAM.py:29-56``: arc motion for |omega| > 1e-6, theta wrapped
    with atan2(sin, cos), straight-line Jacobian F, ``covariance <- F P F^T + motion_noise`` on the GPU
    (general dense path)."""
    v, omega = control
    state = np.asarray(state, dtype=np.float64)
    theta = state[2]
    if abs(omega) > 1e-6:
        dx = -v / omega * np.sin(theta) + v / omega * np.sin(theta + omega * dt)
        dy = v / omega * np.cos(theta) - v / omega * np.cos(theta + omega * dt)
        dth = omega * dt
    else:
        dx, dy, dth = v * np.cos(theta) * dt, v * np.sin(theta) * dt, 0.0
    new_state = state + np.array([dx, dy, dth])
    new_state[2] = np.arctan2(np.sin(new_state[2]), np.cos(new_state[2]))
    F = np.array([[1.0, 0.0, -v * dt * np.sin(theta)], [0.0, 1.0, v * dt * np.cos(theta)], [0.0, 0.0, 1.0]])
    f = _proto_filter("predict")
    f.set_state(new_state, covariance)
    f.predict_dense(F, motion_noise)
    return new_state, f.covariance()


def update(state, covariance, observation, landmark_pos):
    """3-state EKF update against a known landmark, ``src/EKF-SLAM.py:59-84``.  Runs the device update
    on [x, y, theta, lx, ly] with a zero-covariance landmark block, which reduces to the 2x3 H of :67-70
    (observation noise diag(0.5, 0.5), :13)."""
    f = _proto_filter("update")
    mu = np.concatenate([np.asarray(state, dtype=np.float64)[:3], np.asarray(landmark_pos, dtype=np.float64)[:2]])
    P = np.zeros((5, 5))
    P[:3, :3] = covariance
    f.set_state(mu, P)
    f.update([0], [float(observation[0])], [float(observation[1])])
    mu2, P2 = f.state()
    return mu2[:3].copy(), P2[:3, :3].copy()

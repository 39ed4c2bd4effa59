"""CPU: host logic, the C-ABI surface (no compute calls), trajectory sharding over gloo ranks."""
import os
import re
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from oracle import ekf_oracle as orc
from tests import golden_util as gu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def sd():
    import __graft_entry__ as ge
    import slam_duckietown_amd as sd
    if not os.path.exists(sd.library_path()):
        ge.build()
    return sd


def header_functions():
    text = open(os.path.join(ROOT, "include", "ekfslam_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ekf_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(sd):
    import ctypes
    lib = sd.load_library()
    names = header_functions()
    assert len(names) >= 20
    from slam_duckietown_amd import ekf_bindings as eb
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/ekfslam_hip.h but not exported"
        assert name in eb.ABI, f"{name} has no ctypes signature"
    assert sorted(eb.ABI) == names
    # ... and the other way round: nothing named ekf_* leaves the library without a declaration (diagnostics included)
    import subprocess
    nm = subprocess.run(["nm", "-D", "--defined-only", sd.library_path()], check=True, capture_output=True, text=True).stdout
    exported = sorted({ln.split()[-1] for ln in nm.splitlines() if re.fullmatch(r"ekf_[a-z0-9_]+", ln.split()[-1])})
    assert exported == names, sorted(set(exported) ^ set(names))


def test_config_default_matches_reference_constants(sd):
    from slam_duckietown_amd import ekf_bindings as eb
    c = eb._CConfig()
    assert sd.load_library().ekf_config_default(c) == 0
    # src/replay_no_ros.py:15-19, :28, :356, :376
    assert (c.motion_sigma, c.meas_sigma, c.arc_threshold, c.landmark_init_var) == (0.1, 0.7, 1e-2, 10000.0)
    assert (c.enable_measurement_model, c.enable_circular_interpolation, c.disable_motion_model) == (1, 1, 0)
    d = sd.EkfConfig()
    assert (d.motion_sigma, d.meas_sigma, d.arc_threshold, d.landmark_init_var, d.gate_range) == (0.1, 0.7, 1e-2, 1e4, 1.5)


def test_no_gpu_means_loud_failure_not_fallback(sd):
    try:
        f = sd.EkfSlam(43)
    except sd.EkfError as e:
        assert "no CPU fallback" in str(e) or "gfx950" in str(e)
    else:                       # a GPU is present (GPU box): the handle must be real
        assert f.size() == 3
        f.close()


def test_missing_library_is_an_error(sd, monkeypatch):
    from slam_duckietown_amd import ekf_bindings as eb
    monkeypatch.setattr(eb, "_lib", None)
    monkeypatch.setattr(eb, "_LIB_NAME", "libekfslam_hip_missing.so")
    with pytest.raises(sd.EkfError):
        eb.load_library()


@pytest.mark.parametrize("case", gu.REPLAY_CASES)
def test_frontend_association_matches_reference_order(sd, case):
    """Host-side a2: tag->index map, gate, averaging, update order (vs the reference's own outputs)."""
    g = gu.load(case)
    tag_index, o_index = {}, {}
    for k in range(len(g["lin"])):
        det = gu.detections_for_step(g, k)
        n = int(g["out_size"][k - 1]) if k else 3
        pose = g["out_mean"][k - 1, :3] if k else np.zeros(3)
        tp = sd.associate(det, tag_index, pose, ignore_tags=gu.ignore_tags(g))
        op = orc.associate(det, o_index, pose, orc.EkfConfig(ignore_tags=gu.ignore_tags(g)))
        assert list(tp.keys()) == [i for i in g["out_obs_order"][k] if i >= 0]
        assert list(tp.keys()) == list(op.keys())
        for key in tp:
            assert tp[key][3] == op[key][3]
            assert np.array_equal(np.array(tp[key], dtype=float), np.array(op[key], dtype=float))
    assert sorted(tag_index.items(), key=lambda kv: kv[1]) == [tuple(r) for r in g["out_tag_index"]]
    if case == "replay_ignore_tags":                 # the ignored tags were detected, and got no landmark index
        assert set(gu.ignore_tags(g)) <= {int(t) for t in g["det_tag_id"]} and not set(gu.ignore_tags(g)) & set(tag_index)


def test_frontend_gate_and_ignore(sd):
    from types import SimpleNamespace as NS
    mk = lambda i, x, z: NS(tag_id=i, pose_R=np.eye(3), pose_t=np.array([[x], [0.0], [z]]), pose_err=0.0)
    ti = {}
    tp = sd.associate([(0, [mk(1, 0.0, 1.6), mk(2, 0.9, 1.21), mk(3, 0.9, 1.19), mk(4, 0.1, 0.5)])], ti,
                      np.zeros(3), ignore_tags=(4,))
    assert ti == {3: 0} and list(tp) == [0]          # 1 and 2 are beyond 1.5 m (:289), 4 ignored (:286)


def test_odometry_golden(sd):
    g = gu.load("odometry")
    for (a, b, c, d), dphi, disp in zip(g["ticks"], g["dphi"], g["disp"]):
        l = sd.delta_phi(int(a), int(b), int(g["resolution"]))
        r = sd.delta_phi(int(c), int(d), int(g["resolution"]))
        assert (l, r) == tuple(dphi)
        assert sd.displacement(float(g["wheel_radius"]), float(g["baseline"]), l, r) == tuple(disp)


def test_benchmark_stream_generator_equals_oracle_generator(sd):
    import slam_duckietown_amd.synthetic as syn
    for args in [(20, 40, 8, 0), (50, 25, 1, 7), (300, 12, 8, 31)]:
        for x, y in zip(syn.synthetic_stream(*args), orc.synthetic_stream(*args)):
            assert np.array_equal(x, y)


def test_synthetic_stream_stays_inside_the_gate(sd):
    import slam_duckietown_amd.synthetic as syn
    _, _, _, _, _, zr, _ = syn.synthetic_stream(500, 200, 8, 1)
    assert zr.max() < 1.5            # every observation would pass the reference's gate (:289)


def test_shard_partition_properties(sd):
    from slam_duckietown_amd.sharding import shard_trajectories
    for total, world in [(256, 8), (32, 1), (10, 4), (3, 8)]:
        parts = [shard_trajectories(total, world, r) for r in range(world)]
        flat = [t for p in parts for t in p]
        assert flat == list(range(total))
        assert max(map(len, parts)) - min(map(len, parts)) <= 1
    assert [len(shard_trajectories(256, 8, r)) for r in range(8)] == [32] * 8


def test_banks_of_a_rank(sd):
    from slam_duckietown_amd.sharding import split_banks
    for count, size in [(32, 32), (64, 32), (48, 32), (33, 32), (5, 2), (0, 32), (100, 32)]:
        ids = list(range(7, 7 + count))
        banks = split_banks(ids, size)
        assert [t for b in banks for t in b] == ids
        assert all(1 <= len(b) <= size for b in banks)
        assert len(banks) == -(-count // size)
        if banks:
            assert max(map(len, banks)) - min(map(len, banks)) <= 1
    assert [len(b) for b in split_banks(range(64))] == [32, 32]
    with pytest.raises(ValueError):
        split_banks([1, 2], 0)


RANK_SCRIPT = textwrap.dedent("""
    import json, os, sys, time
    sys.path.insert(0, {root!r})
    import slam_duckietown_amd.sharding as shard
    grp = shard.RankGroup()
    ids = shard.shard_trajectories(int(os.environ["EKF_TEST_TRAJECTORIES"]), grp.world, grp.rank)
    work = 0.05 * (grp.rank + 1)
    dt = shard.timed_region(grp, lambda: time.sleep(work), lambda: None, time.perf_counter)
    total = shard.aggregate_steps_per_second(len(ids) * 7, grp, dt) * dt
    own = shard.timed_region.last_local_seconds
    per_rank = grp.gather_over_ranks(own)
    print(json.dumps(dict(rank=grp.rank, ids=ids, dt=dt, total=total, own=own, per_rank=per_rank)), flush=True)
    grp.close()
""")


def run_ranks(tmp_path, world, trajectories):
    import json
    import socket
    script = tmp_path / "rank.py"
    script.write_text(RANK_SCRIPT.format(root=ROOT))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), EKF_TEST_TRAJECTORIES=str(trajectories), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        out, _ = p.communicate(timeout=300)
        assert p.returncode == 0
        outs.append(json.loads(out.strip().splitlines()[-1]))
    outs.sort(key=lambda o: o["rank"])
    return outs


def test_two_rank_gloo_sharding_and_timing(tmp_path):
    """world_size 2 over gloo on the CPU: disjoint shards, max-over-ranks time, summed units."""
    outs = run_ranks(tmp_path, 2, 10)
    assert outs[0]["ids"] + outs[1]["ids"] == list(range(10))
    assert abs(outs[0]["dt"] - outs[1]["dt"]) < 1e-12 and outs[0]["dt"] >= 0.1     # slower rank's time on both
    assert abs(outs[0]["total"] - 70.0) < 1e-9
    assert outs[0]["per_rank"] == outs[1]["per_rank"] == [outs[0]["own"], outs[1]["own"]]
    assert outs[0]["own"] < outs[1]["own"] == outs[0]["dt"]


def test_eight_rank_gloo_at_the_real_shape(tmp_path):
    """BASELINE config 4 as the driver's SCALE run launches it: world_size 8, 256 trajectories (seeds 1234 ..
    1489) -> 32 contiguous ids per rank, the timed region's max over ranks on every rank, every rank's own time gathered
    in rank order (what bench.py reports as rank_dt_ms), units summed over all ranks.  gloo on the CPU, no GPU."""
    outs = run_ranks(tmp_path, 8, 256)
    assert [o["rank"] for o in outs] == list(range(8))
    for r, o in enumerate(outs):
        assert o["ids"] == list(range(32 * r, 32 * r + 32))
        assert o["per_rank"] == outs[0]["per_rank"] and len(o["per_rank"]) == 8
        assert abs(o["per_rank"][r] - o["own"]) < 1e-15
        assert abs(o["dt"] - max(o["per_rank"])) < 1e-12 and o["dt"] >= 0.4        # rank 7 sleeps 0.4 s
        assert abs(o["total"] - 256 * 7) < 1e-6
    own = outs[0]["per_rank"]
    assert all(own[r] < own[r + 1] for r in range(7))                              # the skew is visible rank by rank


def test_header_is_plain_c_and_links_against_the_library(tmp_path):
    """include/ekfslam_hip.h is the drop-in boundary: it must compile as C (no C++, no torch types) and a C program
    using it must link against the shared library (no device call is made: there is no GPU here)."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "slam-duckietown_amd", "libekfslam_hip.so")
    if shutil.which("gcc") is None or not os.path.exists(lib):
        pytest.skip("gcc or the built library is not available")
    src = tmp_path / "abi.c"
    src.write_text(
        '#include "ekfslam_hip.h"\n'
        "#include <stdio.h>\n"
        "int main(void) {\n"
        "  ekf_config cfg;\n"
        "  ekf_config_default(&cfg);\n"
        '  printf("%g %g %d %d\\n", cfg.motion_sigma, cfg.meas_sigma, EKF_MMAX, (int)EKF_FLAG_NONFINITE);\n'
        "  return (void*)ekf_step == (void*)0 || (void*)ekf_flush == (void*)0;\n"
        "}\n")
    exe = tmp_path / "abi"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(root, "include"), str(src), "-o", str(exe),
                    lib, "-Wl,-rpath," + os.path.dirname(lib)], check=True, capture_output=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    assert float(out[0]) == 0.1 and float(out[1]) == 0.7 and int(out[2]) == 16 and int(out[3]) == 1


def test_every_entry_point_selects_its_device():
    """A host thread may own handles on several GPUs: every exported function that
    enqueues work on the handle's stream must call hipSetDevice(h->device) first -- itself or through the
    function it delegates to.  Source audit of csrc/ekf_api.hip."""
    text = open(os.path.join(ROOT, "slam-duckietown_amd", "csrc", "ekf_api.hip")).read()
    text = re.sub(r"//[^\n]*", "", text)
    bodies = {}
    for mt in re.finditer(r'^(?:extern "C" |static )[^\n;{]*?\b([A-Za-z_][A-Za-z0-9_]*)\s*\(([^)]*)\)\s*\{', text, flags=re.M):
        depth, i = 1, mt.end()
        while depth:
            depth += {"{": 1, "}": -1}.get(text[i], 0)
            i += 1
        bodies[mt.group(1)] = (mt.group(0).startswith("extern"), mt.group(2), text[mt.end():i])
    exported = {k for k, v in bodies.items() if v[0] and "ekf_handle* h" in v[1]}
    assert {"ekf_step", "ekf_stream_run", "ekf_download_tags", "ekf_set_option", "ekf_flush"} <= exported
    gpu = re.compile(r"\bhip(?!SetDevice)[A-Z]\w*\s*\(|\blaunch_\w+\s*\(|\bdense_propagate\s*\(")

    def callees(name):
        return [c for c in set(re.findall(r"\b([A-Za-z_]\w*)\s*\(", bodies[name][2])) if c in bodies and c != name]

    def touches(name, seen=()):
        return bool(gpu.search(bodies[name][2])) or any(touches(c, seen + (name,)) for c in callees(name) if c not in seen)

    def selects(name, seen=()):
        body = bodies[name][2]
        if "hipSetDevice(h->device)" in body:
            return True
        if gpu.search(body):
            return False
        return all(selects(c, seen + (name,)) for c in callees(name) if c not in seen and touches(c))

    missing = sorted(n for n in exported if touches(n) and not selects(n))
    assert not missing, f"entry points that use the stream without hipSetDevice(h->device): {missing}"


def test_row_slab_pass_hands_out_every_unit_exactly_once(sd):
    """The work queues of the row-slab covariance pass (k_flush_rs): whatever the batch, the number of 128-row slabs and
    the mode (0 uniform chunks, 1 pairs with the unpaired trajectory cut, 2 last trajectories dealt over the queues,
    3 the same in half slabs),
    the eight queues together hand out every (trajectory, slab) exactly once -- as one whole slab or as all of its
    chunks.  The integer functions are the ones the kernel calls (`__host__ __device__`), reached here through an
    diagnostics section of the header (`ekf_debug_pass_units`); no device needed."""
    import ctypes as C
    lib = sd.load_library()
    fn = lib.ekf_debug_pass_units
    buf = (C.c_int * 70000)()
    for batch in list(range(1, 42)) + [64, 100]:
        for nrb in (1, 2, 5, 7, 8, 9, 15, 16, 17, 32, 126):
            for nch, mode in ((1, 0), (2, 0), (3, 0), (2, 1), (1, 2), (1, 3), (3, 3), (8, 3), (16, 3)):
                total = fn(batch, nrb, nch, mode, buf, len(buf))
                assert 0 < total <= len(buf)
                units = np.frombuffer(buf, dtype=np.int32, count=total)
                assert (units >= 0).all()
                code, slab = units & 1023, units >> 10
                if mode == 3:
                    nch = 2                                     # (mode 3: `nch` carried half the chunk length; two chunks)
                traj, rb = slab // nrb, slab % nrb
                assert traj.max() == batch - 1 and traj.min() == 0
                whole = code == 1023
                # every (trajectory, slab) is covered by one whole unit or by chunks 0..nch-1, never both
                count_whole = np.bincount(slab[whole], minlength=batch * nrb)
                count_chunk = np.bincount(slab[~whole], minlength=batch * nrb)
                assert ((count_whole == 1) & (count_chunk == 0) | (count_whole == 0) & (count_chunk == nch)).all(), \
                    (batch, nrb, nch, mode)
                if (~whole).any():
                    key = slab[~whole].astype(np.int64) * 1024 + code[~whole]
                    assert len(np.unique(key)) == len(key) and code[~whole].max() == nch - 1
                if mode == 2:                                   # equal work: queue sizes differ by at most the dealt remainder
                    sizes = [fn(batch, nrb, nch, mode, None, 0)]   # (total only; per-queue balance is implied by coverage)
                    assert sizes[0] == batch * nrb


def test_host_planning_logic_under_the_sanitizers(tmp_path):
    """SURVEY.md section 5 ("-fsanitize=address,undefined host build"): the host-side planning logic the library ships
    (csrc/ekf_host_plan.h + the layout helpers of csrc/ekf_device.h: work queues and equal static shares of the row-slab
    pass, pass / cadence planning, step records and their active bound, validation of observation lists, the
    covariance's device layout) compiled with plain g++ under AddressSanitizer + UndefinedBehaviorSanitizer and run
    through tests/host_plan_check.cpp: the enumerations of the tests above plus randomised (batch, size, option,
    m-sequence) invariants of `plan_pass` / `cadence_length` / `fill_step` / `validate_obs`.  CPU only."""
    import shutil
    if shutil.which("g++") is None:
        pytest.skip("g++ is not available")
    exe = tmp_path / "host_plan_check"
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-DEKF_HOST_ONLY",
           "-Wall", "-Werror", "-I", os.path.join(ROOT, "slam-duckietown_amd", "csrc"), "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "host_plan_check.cpp"), "-o", str(exe)]
    built = subprocess.run(cmd, capture_output=True, text=True)
    assert built.returncode == 0, built.stderr
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1"))
    assert run.returncode == 0, run.stdout + run.stderr
    assert "checks passed" in run.stdout and "runtime error" not in run.stderr and "AddressSanitizer" not in run.stderr
    # the library itself is built from the same header (a stale copy would make this test vacuous)
    api = open(os.path.join(ROOT, "slam-duckietown_amd", "csrc", "ekf_api.hip")).read()
    assert '#include "ekf_host_plan.h"' in api and "struct ekf_handle : ekf::HostPlan" in api
    for fn in ("plan_pass", "plan_cadences", "cadences_possible", "fill_step", "validate_obs", "build_pass_shares", "order_pass_shares"):
        assert re.search(r"\b%s\(" % fn, api), fn                      # called from the API ...
        assert not re.search(r"^(static|inline)[^\n;]*\b%s\(" % fn, api, flags=re.M), fn   # ... and defined only in the header


def test_store_hazard_guard_is_in_the_shipped_machine_code(sd):
    """The 16-byte buffer stores of the row-slab pass (offen + SGPR soffset: the form LLVM's hazard recogniser does not
    cover): in the library that ships no instruction may overwrite a store's data registers within
    two wait states of the store -- checked on the disassembly of the gfx950 code objects inside libekfslam_hip.so, not
    on the source."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_lint
    stores, gaps, bad = isa_lint.lint(sd.library_path())
    assert stores >= 500, f"only {stores} guarded stores found: did the pass kernel change its store form?"
    assert not bad, "\n".join(bad)
    assert isa_lint.wait_states("s_nop 1") == 2 and isa_lint.wait_states("v_mov_b32_e32 v1, v2") == 1
    # the scanner itself: a VALU write of a data register between store and nop is what it exists to catch
    assert isa_lint.writes_vgpr("v_add_u32_e32 v78, v1, v2", 76, 79)
    assert isa_lint.writes_vgpr("ds_read_b128 v[76:79], v248 offset:4096", 64, 76)
    assert not isa_lint.writes_vgpr("s_add_i32 s37, s60, s79", 76, 79)
    assert not isa_lint.writes_vgpr("v_add_u32_e32 v80, v78, v2", 76, 79)
    assert not isa_lint.writes_vgpr("buffer_store_dwordx4 v[76:79], v251, s[48:51], s39 offen nt", 76, 79)
    # round 6: the fp64 DPP broadcasts of the fused cadence's panel launch (inline assembly the compiler's hazard recogniser does
    # not look into): fed from LDS reads, never within two wait states of a VALU write of their source or five of a v_cmpx
    dpps, bad_dpp = isa_lint.lint_dpp(sd.library_path())
    assert dpps >= 6000, f"only {dpps} DPP operations found: did the panel launch change its down-date?"
    assert not bad_dpp, "\n".join(bad_dpp[:20])


def test_create_rejects_sizes_beyond_the_32_bit_offsets(sd):
    """The kernels address one covariance with unsigned 32-bit byte offsets: ekf_create refuses an n_max whose padded
    allocation (rows x column panels of 4096 doubles) reaches 4 GiB, before it looks for a device (so this runs
    without one)."""
    import ctypes
    from slam_duckietown_amd import ekf_bindings as eb
    lib = sd.load_library()
    text = open(os.path.join(ROOT, "include", "ekfslam_hip.h")).read()
    limit = int(re.search(r"#define EKF_N_MAX_LIMIT (\d+)", text).group(1))
    assert limit == eb.EKF_N_MAX_LIMIT and limit % 2 == 1
    alloc = lambda rows: rows * (-(-rows // 4096)) * 4096 * 8   # rows x column panels of 4096 doubles (ekf_device.h)
    rows = (limit + 63) // 64 * 64
    assert alloc(rows) < 2 ** 32 <= alloc(rows + 64)
    h = ctypes.c_void_p()
    for n_max in (limit + 2, 3 + 2 * 12000, 3 + 2 * 50000):
        assert lib.ekf_create(0, n_max, 1, None, ctypes.byref(h)) == -1          # EKF_ERR_ARG
        msg = lib.ekf_last_error(None).decode()
        assert "EKF_N_MAX_LIMIT" in msg and str(limit) in msg and not h.value
        with pytest.raises(sd.EkfError, match="EKF_N_MAX_LIMIT"):
            sd.EkfSlam(n_max)
    # at the limit the size check passes (what fails here, without a GPU, is the device probe)
    rc = lib.ekf_create(0, limit, 1, None, ctypes.byref(h))
    if rc == 0:
        lib.ekf_destroy(h)
    else:
        assert rc == -2 and "EKF_N_MAX_LIMIT" not in lib.ekf_last_error(None).decode()


def test_row_slab_pass_equal_shares_cover_every_strip_once(sd):
    """The static partition the row-slab pass uses for a few long trajectories (N = 8000 x 1: 126 slabs for 256
    workgroups): every strip of every slab of every trajectory in exactly one piece, at most 16 pieces per share, and
    shares of equal cost (strips + 2 per piece) to within a few strips."""
    import ctypes
    lib = sd.load_library()
    for batch, n, wgs in [(1, 16003, 256), (2, 16003, 256), (1, 16003, 240), (3, 12003, 256), (7, 16003, 256),
                          (1, 21823, 256), (3, 1403, 8), (3, 1403, 5), (1, 4003, 8)]:
        out = np.zeros(wgs * 16 * 4, dtype=np.int32)
        longest = lib.ekf_debug_pass_shares(batch, n, wgs, out.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
        assert 1 <= longest <= 16, (batch, n, wgs, longest)
        nrb, s_last = (n + 127) // 128, (n - 1) >> 6
        seen = np.zeros((batch, nrb, s_last + 1), dtype=np.int32)
        costs = []
        for share in out.reshape(wgs, 16, 4):
            cost, ended = 0, False
            for b, rb, start, cnt in share:
                if cnt <= 0:
                    ended = True
                    continue
                assert not ended and 0 <= b < batch and 0 <= rb < nrb and start >= 0
                assert start + cnt <= s_last - 2 * rb + 1
                seen[b, rb, start:start + cnt] += 1
                cost += cnt + 2
            costs.append(cost)
        for rb in range(nrb):
            assert (seen[:, rb, :s_last - 2 * rb + 1] == 1).all() and (seen[:, rb, s_last - 2 * rb + 1:] == 0).all()
        assert max(costs) - min(costs) <= 4, (batch, n, wgs, min(costs), max(costs))


def test_pinned_pool_recycles_only_what_nobody_holds():
    """The binding's pool of pinned buffers behind large returned covariances (ekf_bindings._PinnedPool), driven with malloc /
    free in place of ekf_host_alloc / ekf_host_free: a buffer goes back to the pool when the LAST view of its array is gone
    (sub-views keep it), at most KEEP free buffers per size are kept, buffers of OTHER sizes stay (two handles of different size,
    a map that grows a landmark at a time: ADVICE r04) until LIMIT forces the least recently used size out, and beyond LIMIT
    the arrays are ordinary ones."""
    import ctypes as C
    import gc
    from slam_duckietown_amd import ekf_bindings as eb
    libc = C.CDLL(None)
    libc.malloc.restype, libc.malloc.argtypes, libc.free.argtypes = C.c_void_p, [C.c_size_t], [C.c_void_p]

    class Lib:
        allocated, freed = [], []

        def ekf_host_alloc(self, nbytes):
            p = libc.malloc(nbytes)
            self.allocated.append(p)
            return p

        def ekf_host_free(self, p):
            self.freed.append(p)
            libc.free(p)

    lib, pool = Lib(), eb._PinnedPool()
    a = pool.empty(lib, (150, 150))
    assert a.shape == (150, 150) and a.flags.writeable and a.flags.c_contiguous and a.ctypes.data == lib.allocated[0]
    a[:] = 3.0
    corner = a[:2, :2]
    del a
    gc.collect()
    assert pool.free == {} and pool.live == 8 * 150 * 150           # the sub-view still owns the buffer
    assert corner.sum() == 12.0
    del corner
    gc.collect()
    assert pool.free == {8 * 150 * 150: [lib.allocated[0]]}
    b = pool.empty(lib, (150, 150))
    assert b.ctypes.data == lib.allocated[0] and len(lib.allocated) == 1       # handed out again, not allocated again
    c, d, e = (pool.empty(lib, (150, 150)) for _ in range(3))
    del b, c, d, e
    gc.collect()
    assert len(pool.free[8 * 150 * 150]) == pool.KEEP and len(lib.freed) == 4 - pool.KEEP
    assert pool.live == pool.KEEP * 8 * 150 * 150
    s150, s152 = 8 * 150 * 150, 8 * 152 * 152
    f = pool.empty(lib, (152, 152))                                  # another size: the free buffers of the first one stay
    assert len(pool.free[s150]) == pool.KEEP and pool.live == pool.KEEP * s150 + s152 and len(lib.freed) == 4 - pool.KEEP
    b2 = pool.empty(lib, (150, 150))                                 # ... and are handed out again (a second handle's size)
    assert b2.ctypes.data in lib.allocated[:4] and len(lib.allocated) == 5
    del b2
    gc.collect()
    pool.LIMIT = pool.live + s152 - s150                             # (instance attribute: this pool only) room for ONE more 152 x 152
    g = pool.empty(lib, (152, 152))                                  # ... if a free buffer of the least recently used size goes
    assert g.base is not None and len(pool.free[s150]) == pool.KEEP - 1 and len(lib.freed) == 4 - pool.KEEP + 1
    assert pool.live == (pool.KEEP - 1) * s150 + 2 * s152
    pool.LIMIT = pool.live - (pool.KEEP - 1) * s150                  # nothing fits any more, whatever is evicted
    k = pool.empty(lib, (152, 152))
    assert k.shape == (152, 152) and k.base is None and f.base is not None   # k: a plain np.empty
    del f, g, k
    gc.collect()
    for size in list(pool.free):
        for p in pool.free.pop(size):
            lib.ekf_host_free(p)


def test_binding_stages_arguments_without_a_device():
    """EkfSlam's per-call staging (padded [batch, stride] arrays with cached pointers, one value per trajectory): shapes,
    padding, reuse and the errors, on an object that never touched the library."""
    import ctypes as C
    from slam_duckietown_amd import ekf_bindings as eb
    f = eb.EkfSlam.__new__(eb.EkfSlam)
    f.batch = 3
    f._lin, f._ang, f._m = np.zeros(3), np.zeros(3), np.zeros(3, dtype=np.int32)
    f._plin, f._pang, f._pm = eb._p(f._lin), eb._p(f._ang), eb._p(f._m, eb._ip)
    f._stages, f._out, f._h = {}, None, C.c_void_p()
    f._per_traj(0.25, "lin", f._lin)
    assert f._lin.tolist() == [0.25] * 3
    f._per_traj([1.0, 2.0, 3.0], "lin", f._lin)
    assert f._lin.tolist() == [1.0, 2.0, 3.0]
    f._per_traj(np.array([7.0]), "lin", f._lin)
    assert f._lin.tolist() == [7.0] * 3
    for wrong in ([1.0, 2.0], np.zeros((3, 1)), np.zeros(4)):
        with pytest.raises(ValueError):
            f._per_traj(wrong, "lin", f._lin)
    pI, pR, pB, pm, stride = f._obs([[4, 2], [], [1, 0, 3]], [[1.0, 2.0], [], [3.0, 4.0, 5.0]], [[.1, .2], [], [.3, .4, .5]])
    I, R, B = f._stages[3][:3]
    assert stride == 3 and f._m.tolist() == [2, 0, 3] and I.shape == (3, 3) and I.dtype == np.int32
    assert I[0, :2].tolist() == [4, 2] and I[2].tolist() == [1, 0, 3] and R[2].tolist() == [3.0, 4.0, 5.0] and B[0, 1] == .2
    assert C.addressof(pI.contents) == I.ctypes.data and C.addressof(pm.contents) == f._m.ctypes.data
    again = f._obs([[9], [8], [7]], [[1.0], [2.0], [3.0]], [[0.0], [0.0], [0.0]])
    assert again[4] == 1 and f._stages[1][0][:, 0].tolist() == [9, 8, 7] and f._m.tolist() == [1, 1, 1]
    assert f._obs([[4, 2], [], [1, 0, 3]], [[1.0, 2.0], [], [3.0, 4.0, 5.0]], [[.1, .2], [], [.3, .4, .5]])[0] is pI   # reused
    blk = f._obs(np.array([[1, 2], [3, 4], [5, 6]], dtype=np.int64), np.full((3, 2), 1.5), np.arange(6.0).reshape(3, 2))
    assert blk[4] == 2 and f._m.tolist() == [2, 2, 2] and f._stages[2][0].tolist() == [[1, 2], [3, 4], [5, 6]]      # a bank at once
    assert f._stages[2][2][2].tolist() == [4.0, 5.0] and f._stages[2][1][1, 0] == 1.5
    with pytest.raises(ValueError):
        f._obs(np.zeros((2, 2), dtype=np.int32), np.zeros((2, 2)), np.zeros((2, 2)))      # two rows for three trajectories
    with pytest.raises(ValueError):
        f._obs(np.zeros((3, 2), dtype=np.int32), np.zeros((3, 1)), np.zeros((3, 2)))
    with pytest.raises(ValueError):
        f._obs([[1], [2]], [[1.0], [2.0]], [[0.0], [0.0]])            # two lists for three trajectories
    with pytest.raises(ValueError):
        f._obs([[1], [2], [3]], [[1.0], [2.0, 9.0], [3.0]], [[0.0], [0.0], [0.0]])
    one = eb.EkfSlam.__new__(eb.EkfSlam)
    one.batch = 1
    one._m = np.zeros(1, dtype=np.int32)
    one._pm, one._stages, one._h = eb._p(one._m, eb._ip), {}, C.c_void_p()
    assert one._obs([5, 6], [1.0, 2.0], [0.1, 0.2])[4] == 2 and one._m[0] == 2          # a flat list is one trajectory's
    assert one._obs([], [], [])[4] == 1 and one._m[0] == 0


def test_variable_stream_is_seeded_and_well_formed():
    """synthetic.variable_stream (bench.py's `variable_m` leg: the shapes the reference's loop produces, src/replay_no_ros.py:280-301):
    seeded per trajectory, landmark counts in range, indices distinct inside a step and padded with zeros, measurements finite
    and inside the reference's 1.5 m gate (:289), same world and kinematics as synthetic_stream."""
    import slam_duckietown_amd.synthetic as syn
    a = syn.variable_stream(60, 40, 0, 8, 5)
    b = syn.variable_stream(60, 40, 0, 8, 5)
    c = syn.variable_stream(60, 40, 0, 8, 6)
    assert all(np.array_equal(x, y) for x, y in zip(a, b)) and not np.array_equal(a[7], c[7])
    mean0, diag0, lin, ang, idx, zr, zb, m = a
    assert idx.shape == (40, 8) and idx.dtype == np.int32 and m.dtype == np.int32 and m.min() >= 0 and m.max() <= 8
    assert len(set(m.tolist())) > 3                                    # the count really wanders
    for k in range(40):
        assert len(set(idx[k, :m[k]].tolist())) == m[k] and (idx[k, m[k]:] == 0).all() and (zr[k, m[k]:] == 0).all()
        assert (zr[k, :m[k]] > 0).all() and (zr[k, :m[k]] < 1.5).all() and np.isfinite(zb[k]).all()
    ref = syn.synthetic_stream(60, 40, 8, 5)
    assert np.array_equal(mean0, ref[0]) and np.array_equal(diag0, ref[1]) and np.array_equal(lin, ref[2]) and np.array_equal(ang, ref[3])

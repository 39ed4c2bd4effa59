#!/usr/bin/env python3
"""Timeline of the first units of every workgroup of k_flush_rs (diagnostic library built with -DRS_STAMPS):
  make -C slam-duckietown_amd/csrc variant TAG=stamps EXTRA=-DRS_STAMPS
  EKFSLAM_HIP_VARIANT=stamps python3 tools/rs_stamps.py"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import slam_duckietown_amd as sd
import slam_duckietown_amd.synthetic as syn

N, B, m, steps = 2000, 32, 8, 5
n = 3 + 2 * N
streams = [syn.synthetic_stream(N, steps + 5, m, t) for t in range(B)]
f = sd.EkfSlam(n, batch=B)
f.set_option("active_bound", 0)
for b, s in enumerate(streams):
    f.set_state_diag(s[0], s[1], b)
f.stream_upload(*[np.stack([s[i] for s in streams], 1) for i in (2, 3, 4, 5, 6)])
f.stream_run(0, 10)          # two passes; the stamps of the second one remain
f.sync()
words = 8 * 32 + 256 * 64 * 2
buf = np.zeros(words, dtype=np.uint32)
lib = f._lib
assert lib.ekf_debug_read(f._h, buf.ctypes.data_as(C.c_void_p), C.c_long(buf.nbytes)) == 0
st = buf[8 * 32:].view(np.uint64).reshape(256, 64).astype(np.int64)
names = ["loop top", "barrier 1", "pop+barrier 2", "scalars", "loads landed+strip stored", "tile0->image", "tile0->acc, barrier",
         "body 0", "body 1", "body 2", "body 4", "body 6", "pair loop end", "last body", "drain"]
for u in range(3):
    S = st[:, u * 20 + 15]
    ok = (st[:, u * 20 + 14] > 0) & (S > 8)
    print(f"unit #{u} of a workgroup: {ok.sum()} workgroups with S > 8, median S = {np.median(S[ok]) if ok.any() else 0}")
    if not ok.any():
        continue
    base = st[ok, u * 20 + 0]
    prev = base
    for k in range(1, 15):
        cur = st[ok, u * 20 + k]
        d = cur - prev                # s_memtime counts shader cycles on this part (tools/lat_probe: 2.39 GHz on an idle chip)
        print(f"   {names[k]:28s} +{np.median(d):9.0f} cycles   (since loop top {np.median(cur - base):9.0f})")
        prev = cur
f.close()

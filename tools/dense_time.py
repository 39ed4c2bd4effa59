import sys, time, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import slam_duckietown_amd as sd
n = 4003
f = sd.EkfSlam(n)
rng = np.random.default_rng(0)
F = np.eye(n) + 0.01 * rng.normal(size=(n, n)); Q = np.eye(n) * 0.01
f.set_state_diag(np.zeros(n), np.ones(n))
f.predict_dense(F, Q)
f.profile_enable(True)
for _ in range(3): f.predict_dense(F, Q)
ms, cnt = f.profile_read()
print(f"predict_dense n={n}: {ms/cnt:.2f} ms per propagate = {4*n**3/(ms/cnt)/1e9:.1f} TFLOP/s fp64")

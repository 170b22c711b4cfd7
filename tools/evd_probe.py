import numpy as np, sys, time
sys.path.insert(0, '/root/repo')
import isle_amd
hp = isle_amd.HotPath()
rng = np.random.default_rng(0)
n = 400
A = rng.standard_normal((n, n)).astype(np.float32); S = (A + A.T) / 2
for _ in range(3):
    t = time.perf_counter(); ev, vec = hp.eig_sym(S); print("evd ms", (time.perf_counter() - t) * 1e3)

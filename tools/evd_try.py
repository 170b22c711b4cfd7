import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from isle_amd import HotPath
hp = HotPath(0)
for n in (16, 30, 57, 200, 400, 1000, 2000):
    rng = np.random.default_rng(n)
    A = rng.standard_normal((n, n))
    S = (A @ A.T + (A + A.T)).astype(np.float32)
    t0 = time.time(); e, v = hp.eig_sym(S); t1 = time.time()
    er = np.linalg.eigvalsh(S.astype(np.float64))[::-1]
    scale = np.abs(er).max()
    print(n, "time %.1f ms" % ((t1 - t0) * 1e3), "eval err %.2e" % (np.abs(e - er).max() / scale),
          "orth %.2e" % np.abs(v.astype(np.float64).T @ v - np.eye(n)).max(),
          "resid %.2e" % (np.abs(S.astype(np.float64) @ v - v * e).max() / scale), flush=True)
# clustered spectrum like a Ritz matrix: one dominant, tight cluster (relative gaps 1e-3), tail
n = 400
rng = np.random.default_rng(1)
lam = np.concatenate([[1e4], 100 * (1 + 1e-3 * np.arange(199)), np.linspace(50, 0.0, 200)])
Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
S = ((Q * lam) @ Q.T).astype(np.float32)
e, v = hp.eig_sym(S)
er = np.linalg.eigvalsh(S.astype(np.float64))[::-1]
print("clustered", "eval err %.2e" % (np.abs(e - er).max() / 1e4), "orth %.2e" % np.abs(v.astype(np.float64).T @ v - np.eye(n)).max(),
      "resid %.2e" % (np.abs(S.astype(np.float64) @ v - v * e).max() / 1e4))
S = np.eye(64, dtype=np.float32)
e, v = hp.eig_sym(S)
print("identity", e[:3], np.abs(v.T @ v - np.eye(64)).max())

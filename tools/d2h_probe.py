#!/usr/bin/env python3
"""How long the runtime takes to bring 40 MB (the partition of 10 M documents) to the host: into a fresh pageable array (what the binding
hands the library), into a touched pageable array, into page-locked memory, and page-locked + a host copy."""
import time
import numpy as np
import torch

n = 10_000_000
t = torch.arange(n, dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
pin = torch.empty(n, dtype=torch.int32).pin_memory()
for rep in range(3):
    a = torch.empty(n, dtype=torch.int32)  # fresh pages
    t0 = time.perf_counter(); a.copy_(t); torch.cuda.synchronize(); t1 = time.perf_counter()
    a.copy_(t); torch.cuda.synchronize(); t2 = time.perf_counter()
    pin.copy_(t, non_blocking=True); torch.cuda.synchronize(); t3 = time.perf_counter()
    b = np.empty(n, np.int32); t4 = time.perf_counter()
    np.copyto(b, pin.numpy()); t5 = time.perf_counter()
    np.copyto(b, pin.numpy()); t6 = time.perf_counter()
    print("fresh pageable %.2f ms, touched pageable %.2f ms, page-locked %.2f ms, host copy into fresh %.2f ms / touched %.2f ms" %
          ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t5 - t4) * 1e3, (t6 - t5) * 1e3), flush=True)

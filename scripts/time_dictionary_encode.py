"""dictionary_encode of 30 M names, 150 K distinct: ascending dictionary vs first-seen order, best of 5 (host-side share of the call)"""
import time

import numpy as np
import pyarrow as pa

from datafusion_amd import ops
from datafusion_amd.table import DeviceTable

n, distinct = 30_000_000, 150_000
rng = np.random.default_rng(3)
names = pa.array([f"Customer#{i:09d}" for i in range(distinct)])
t = DeviceTable.from_arrow(pa.table({"name": names.take(pa.array(rng.integers(0, distinct, n)))}))
ops.sync()
for how in (True, False):
    best = None
    for _ in range(5):
        t0 = time.perf_counter()
        e = t.dictionary_encode(["name"], sorted=how)
        ops.sync()
        dt = time.perf_counter() - t0
        e.free()
        best = dt if best is None else min(best, dt)
    print(f"dictionary_encode sorted={how}: {best * 1e3:.2f} ms", flush=True)

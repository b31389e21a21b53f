import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
from datafusion_amd import _lib
_lib.init(0)
from datafusion_amd.exchange import Comm
try:
    c = Comm.rccl()
    print("RCCL2 comm ok", rank, flush=True)
    import numpy as np, pyarrow as pa
    from datafusion_amd.table import DeviceTable
    t = pa.table({"k": pa.array(np.arange(100000) + rank * 100000, pa.int64())})
    out = c.hash_exchange(DeviceTable.from_arrow(t), ["k"]).to_arrow()
    print("RCCL2 exchange rows", rank, out.num_rows, c.stats(), flush=True)
except Exception as e:
    print("RCCL2 failed", rank, repr(e)[:500], flush=True)
os._exit(0)

#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 300 python - <<'PY' 2>&1 | tail -30
import os, sys
sys.path.insert(0, os.getcwd())
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29555")
from datafusion_amd import _lib, ops
torch.cuda.set_device(0); _lib.init(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from datafusion_amd.exchange import hash_exchange
orders = ops.tpch_orders(100.0).select(["o_orderkey", "o_orderdate", "o_shippriority"])
li = ops.tpch_lineitem(100.0).select(["l_orderkey", "l_extendedprice", "l_discount"])
o = hash_exchange(orders, ["o_orderkey"], force=True)
l = hash_exchange(li, ["l_orderkey"], force=True)
print("rows", o.num_rows, l.num_rows, orders.num_rows, li.num_rows)
for mode in (0, 3):
    try:
        ht = ops.JoinHashTable(o, ["o_orderkey"], probe_mode=mode)
        i = ht.info()
        print("mode", mode, {f[0]: getattr(i, f[0]) for f in i._fields_})
        out = ht.probe(l, ["l_orderkey"], "Inner", ["o_orderdate", "o_shippriority"], ["l_orderkey", "l_extendedprice", "l_discount"])
        print("out rows", out.num_rows)
    except Exception as e:
        print("ERR", mode, e)
for i in range(3):
    v = o.column_view(i); print(v.name, v.field.type, v.field.nullable, v.validity, v.null_count)
    v = l.column_view(i); print(v.name, v.field.type, v.field.nullable, v.validity, v.null_count)
sys.stdout.flush(); os._exit(0)
PY

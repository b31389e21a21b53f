#!/bin/bash
# usage (GPU box, via gpurun): CASES=<bench_ops cases> scripts/profile_sq.sh <tag> [kernel-name filter]
#        CMD="python scripts/bench_join_shapes.py --only ... --iters 1" scripts/profile_sq.sh <tag> [filter]   (any command; it runs from /tmp: absolute paths)
# Where the waves of the per-operator kernels spend their cycles: SQ counters in separate rocprofv3 --pmc passes (counters only, as
# the pool requires), summed over each kernel's largest launch.  Writes gpurun_out/<tag>/sq.md
set -u
TAG=$1
FILTER=${2:-}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
CASES=${CASES:-agg_multikey}
CMD=${CMD:-python $R/scripts/bench_ops.py --only $CASES --iters 2}
i=0
for SET in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_BUSY_CYCLES" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  (timeout ${PASS_TIMEOUT:-300} rocprofv3 --pmc $SET -d /tmp/sq_$i -o pmc -- $CMD) > $OUT/sq_$i.log 2>&1
done
cd $R
python - <<PY > $OUT/sq.md 2>&1
import sqlite3, glob
short = lambda n: n.replace("void ", "").split("(")[0].replace("dfgpu::", "")
vals = {}
for db in sorted(glob.glob("/tmp/sq_*/pmc_results.db")):
    con = sqlite3.connect(db)
    try:
        # one row per (dispatch, counter): take each kernel's longest dispatch
        rows = con.execute("select kernel_name, counter_name, value, duration, dispatch_id from counters_collection").fetchall()
    except Exception as e:
        print("(", db, e, ")")
        continue
    finally:
        con.close()
    best = {}
    for k, c, v, d, disp in rows:
        k = short(k)
        if k not in best or d > best[k][0]:
            best[k] = (d, disp)
    for k, c, v, d, disp in rows:
        k = short(k)
        if best[k][1] == disp:
            vals.setdefault(k, {})[c] = v
            vals[k]["duration_ms"] = d / 1e6
print("# SQ / cache counters of each kernel's longest launch (rocprofv3 --pmc, one pass per group of counters)\n")
for k, d in sorted(vals.items(), key=lambda kv: -kv[1].get("duration_ms", 0)):
    if "$FILTER" and "$FILTER" not in k:
        continue
    if d.get("duration_ms", 0) < 0.2:
        continue
    print("## " + k + f" ({d['duration_ms']:.3f} ms)\n")
    for c, v in sorted(d.items()):
        if c != "duration_ms":
            print(f"- {c}: {v:.4g}")
    print()
PY
cat $OUT/sq.md | head -120

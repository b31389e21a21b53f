#!/bin/bash
# usage (on the GPU box, via gpurun): scripts/fetch_calib.sh <tag>
# FETCH_SIZE / WRITE_SIZE calibration on known byte counts per access pattern (scripts/microbench/fetch_calib.hip)
# -> gpurun_out/<tag>/fetch_calib.md
set -u
TAG=$1
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 scripts/microbench/fetch_calib.hip -o /tmp/fetch_calib || exit 1
export TMPDIR=/tmp
cd /tmp
/tmp/fetch_calib > $OUT/fetch_calib_known.txt
(timeout 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/calib_fetch -o pmc -- /tmp/fetch_calib) > $OUT/calib_fetch.log 2>&1
(timeout 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/calib_write -o pmc -- /tmp/fetch_calib) > $OUT/calib_write.log 2>&1
(timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/calib_stats -o trace -- /tmp/fetch_calib) > $OUT/calib_stats.log 2>&1
cd $R
python - <<PY > $OUT/fetch_calib.md 2>&1
import sqlite3
def q(db, sql):
    con = sqlite3.connect(db)
    try: return list(con.execute(sql))
    finally: con.close()
def short(n): return n.replace("void ", "").split("(")[0].replace(" ", "")
known = {}
for line in open("$OUT/fetch_calib_known.txt"):
    p = line.split()
    if p and p[0].startswith("KNOWN"): known["".join(p[1:-1])] = (p[0], int(p[-1]))
fetch = {short(r[0]): r[1] for r in q("$OUT/calib_fetch/pmc_results.db", "select kernel_name, avg(value) from counters_collection group by 1")}
write = {short(r[0]): r[1] for r in q("$OUT/calib_write/pmc_results.db", "select kernel_name, avg(value) from counters_collection group by 1")}
dur = {short(r[0]): r[1] for r in q("$OUT/calib_stats/trace_results.db", "select name, average from top_kernels")}
print("# FETCH_SIZE / WRITE_SIZE calibration on gfx950 (known bytes per launch / counter x 1024)\n")
print("| kernel | known bytes | kind | FETCH_SIZE KB | known / (FETCH_SIZE*1024) | WRITE_SIZE KB | known / (WRITE_SIZE*1024) | avg us | known GB/s |\n|---|---:|---|---:|---:|---:|---:|---:|---:|")
for k, (kind, b) in known.items():
    f, w, d = fetch.get(k), write.get(k), dur.get(k)
    fr = "" if not f else f"{b / (f * 1024):.3f}"
    wr = "" if not w else f"{b / (w * 1024):.3f}"
    print(f"| {k} | {b} | {'bytes' if kind == 'KNOWN' else '64-B lines touched'} | {'' if f is None else round(f)} | {fr if 'write' not in k else ''} | {'' if w is None else round(w)} | {wr if 'write' in k else ''} | {'' if d is None else round(d / 1e3, 1)} | {'' if not d else round(b / d, 1)} |")
print("\nraw (kernel, FETCH_SIZE KB, WRITE_SIZE KB, avg ns):")
for k in sorted(set(fetch) | set(write) | set(dur)): print("-", k, fetch.get(k), write.get(k), dur.get(k))
PY
cat $OUT/fetch_calib.md
rm -rf $OUT/calib_fetch $OUT/calib_write $OUT/calib_stats

#!/usr/bin/env python3
"""Per-kernel HBM traffic of one command from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as
MI355X_MICROARCH.md prescribes) -> a small markdown table (the .db files are tens of MB and stay on the GPU box).

  python scripts/summarize_pmc.py <fetch_dir> <write_dir> <out.md> [kernel-name filter ...]

gfx950: FETCH_SIZE * 1024 counts half the bytes of coalesced streaming reads (x2, profiles/r2_fetch_calib.md) and 64-byte units
touched for random gathers (x1): both readings are printed, a lookup kernel lies between them."""
import sqlite3
import sys


def load(path):
    con = sqlite3.connect(path + "/pmc_results.db")
    rows = con.execute("select kernel_name, count(*), avg(value), max(value) from counters_collection group by 1").fetchall()
    con.close()
    return {r[0].replace("void ", "").split("(")[0].replace("dfgpu::", ""): r[1:] for r in rows}


def main():
    fetch, write, out = load(sys.argv[1]), load(sys.argv[2]), sys.argv[3]
    keep = sys.argv[4:]
    names = sorted(set(fetch) | set(write), key=lambda k: -((fetch.get(k, (0, 0, 0))[2] or 0) + (write.get(k, (0, 0, 0))[2] or 0)))
    with open(out, "w") as f:
        f.write("| kernel | launches | FETCH_SIZE KB (largest launch) | WRITE_SIZE KB (largest launch) | read GB if streaming (x2) | read GB if random 64-B units (x1) | written GB |\n")
        f.write("|---|---:|---:|---:|---:|---:|---:|\n")
        for k in names:
            if keep and not any(s in k for s in keep):
                continue
            fc = fetch.get(k, (0, 0, 0))
            wc = write.get(k, (0, 0, 0))
            if (fc[2] or 0) + (wc[2] or 0) < 1024:   # below 1 MB: noise
                continue
            f.write(f"| {k} | {fc[0] or wc[0]} | {fc[2] or 0:.0f} | {wc[2] or 0:.0f} | {(fc[2] or 0) * 2048 / 1e9:.2f} | {(fc[2] or 0) * 1024 / 1e9:.2f} | {(wc[2] or 0) * 1024 / 1e9:.2f} |\n")
    print(open(out).read())


if __name__ == "__main__":
    main()

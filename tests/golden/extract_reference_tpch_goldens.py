#!/usr/bin/env python3
"""TPC-H known answers from the reference's own test data.

Reads (read-only)
  /root/reference/datafusion/sqllogictest/test_files/tpch/answers/q{N}.slt.part   pinned query results at SF 0.1
  /root/reference/datafusion/core/tests/tpch-csv/{customer,orders,lineitem}.csv   first rows of dbgen's SF 1 output
  /root/reference/datafusion/core/tests/data/tpch_{table}_small.parquet           20 rows per table of dbgen's SF 1 output (comment columns)
and writes tests/golden/tpch_answers.json: per query the result rows as the answer file prints them
(one string per row; sqllogictest trims trailing zeros of decimals) with the source line, and the CSV rows
restricted to the columns oracle/dbgen.py generates.  Runs only in the authoring container; the JSON is committed.
"""
import csv
import json
import os

REF = "/root/reference/datafusion"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tpch_answers.json")
QUERIES = list(range(1, 23))
SAMPLE_COLUMNS = {
    "customer": ["c_custkey", "c_address", "c_nationkey", "c_phone", "c_acctbal", "c_mktsegment"],
    "orders": ["o_orderkey", "o_custkey", "o_orderstatus", "o_totalprice", "o_orderdate", "o_orderpriority", "o_shippriority"],
    "lineitem": ["l_orderkey", "l_partkey", "l_suppkey", "l_linenumber", "l_quantity", "l_extendedprice", "l_discount", "l_tax", "l_returnflag",
                 "l_linestatus", "l_shipdate", "l_commitdate", "l_receiptdate", "l_shipinstruct", "l_shipmode"],
    "supplier": ["s_suppkey", "s_name", "s_address", "s_nationkey", "s_phone"],
    "nation": ["n_nationkey", "n_name", "n_regionkey"],
    "region": ["r_regionkey", "r_name"],
}


def main():
    out = {"answers": {}, "sf1_sample": {}}
    for q in QUERIES:
        rel = f"sqllogictest/test_files/tpch/answers/q{q}.slt.part"
        lines = open(os.path.join(REF, rel)).read().split("\n")
        sep = max(i for i, l in enumerate(lines) if l.strip() == "----")
        rows = []
        for l in lines[sep + 1:]:
            if not l.strip():
                break
            rows.append(l)
        out["answers"][f"q{q}"] = {"source": f"{rel}:{sep + 2}-{sep + 1 + len(rows)}", "scale_factor": 0.1, "rows": rows}
    for table, cols in SAMPLE_COLUMNS.items():
        rel = f"core/tests/tpch-csv/{table}.csv"
        with open(os.path.join(REF, rel), newline="") as f:
            recs = list(csv.DictReader(f))
        out["sf1_sample"][table] = {"source": rel, "columns": cols, "rows": [[r[c] for c in cols] for r in recs]}
    # every comment string the reference carries, with the key(s) of its row: they pin dbgen's text pool (oracle/dbgen_text.c)
    import pyarrow.parquet as pq
    out["comments"] = {}
    for table, keys, column in (("nation", ["n_nationkey"], "n_comment"), ("region", ["r_regionkey"], "r_comment"), ("supplier", ["s_suppkey"], "s_comment"),
                                ("customer", ["c_custkey"], "c_comment"), ("orders", ["o_orderkey"], "o_comment"), ("part", ["p_partkey"], "p_comment"),
                                ("partsupp", ["ps_partkey", "ps_suppkey"], "ps_comment")):
        rel = f"core/tests/data/tpch_{table}_small.parquet"
        rows = pq.read_table(os.path.join(REF, rel)).to_pylist()
        got = {tuple(r[k] for k in keys): r[column] for r in rows}
        csv_rel = f"core/tests/tpch-csv/{table}.csv"
        with open(os.path.join(REF, csv_rel), newline="") as f:
            for r in csv.DictReader(f):
                got.setdefault(tuple(int(r[k]) for k in keys), r[column])
        out["comments"][table] = {"source": f"{rel}, {csv_rel}", "keys": keys, "scale_factor": 1, "rows": [[list(k), v] for k, v in sorted(got.items())]}
    json.dump(out, open(OUT, "w"), indent=1)
    print("wrote", OUT, {k: len(v["rows"]) for k, v in out["answers"].items()})


if __name__ == "__main__":
    main()

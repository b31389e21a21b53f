#!/usr/bin/env python3
"""TPC-H known answers from the reference's own test data.

Reads (read-only)
  /root/reference/datafusion/sqllogictest/test_files/tpch/answers/q{N}.slt.part   pinned query results at SF 0.1
  /root/reference/datafusion/core/tests/tpch-csv/{customer,orders,lineitem}.csv   first rows of dbgen's SF 1 output
and writes tests/golden/tpch_answers.json: per query the result rows as the answer file prints them
(one string per row; sqllogictest trims trailing zeros of decimals) with the source line, and the CSV rows
restricted to the columns oracle/dbgen.py generates.  Runs only in the authoring container; the JSON is committed.
"""
import csv
import json
import os

REF = "/root/reference/datafusion"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tpch_answers.json")
QUERIES = [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 14, 15, 16, 17, 18, 19, 20, 21, 22]
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
    json.dump(out, open(OUT, "w"), indent=1)
    print("wrote", OUT, {k: len(v["rows"]) for k, v in out["answers"].items()})


if __name__ == "__main__":
    main()

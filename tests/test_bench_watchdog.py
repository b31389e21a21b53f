"""bench.py's watchdog (N > 1): a rank stuck in a collective cannot be rescued, but the run still leaves ONE JSON line — the measurements
that did finish, or value null — with a `watchdog` object, and every rank exits.  No GPU: the timer thread and the line, with a main
thread that simply never comes back."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PROG = """
import argparse, sys, time
sys.path.insert(0, {root!r})
import bench
args = argparse.Namespace(watchdog_s=1.0, workload="join", steps=5, warmup=2, sf=100.0)
bench.start_watchdog(args, {rank}, 2)
bench.checkpoint_phase("repartition")
{measured}
bench.checkpoint_phase("pruned")
time.sleep(60)      # "inside a collective that never returns"
print("not reached")
"""


def _run(rank, measured):
    prog = PROG.format(root=ROOT, rank=rank, measured='bench.checkpoint_phase("repartition_stream", {"metric": "m", "value": 1.5, "exchanges": {"repartition": {}}})' if measured else "")
    return subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, timeout=40)


def test_watchdog_prints_the_measured_line_and_exits():
    p = _run(0, True)
    assert p.returncode == 0 and "not reached" not in p.stdout
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["value"] == 1.5 and line["watchdog"]["fired_in_phase"] == "pruned" and line["watchdog"]["measured_before_it_fired"] is True


def test_watchdog_with_nothing_measured_prints_a_null_line_and_fails():
    p = _run(0, False)
    assert p.returncode == 3
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["value"] is None and line["n_gpus"] == 2 and line["watchdog"]["measured_before_it_fired"] is False


def test_other_ranks_leave_quietly_after_rank_0():
    p = _run(1, True)
    assert p.returncode == 3 and p.stdout.strip() == ""

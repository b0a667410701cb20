"""CPU suite: the default `bench.py` run on one GPU is an orchestrator that stays off the GPU and runs two children (bench.with_human_like_leg): the measurement itself
and, if the time budget allows, the human-like leg.  Whatever the second child does, the first one's JSON line is what the driver gets -- exactly one line."""
import argparse, io, json, os, sys, types
import common

sys.path.insert(0, common.ROOT)
import bench


class _Run:
    def __init__(self, rc, out):
        self.returncode, self.stdout, self.stderr = rc, out.encode(), b""


def _args(budget):
    return argparse.Namespace(human_like_budget=budget, steps=2, warmup=1, pairs=1000, batches=2, inflight=2, mis=5, cache="/tmp/x")


def _line(value, model="planted"):
    return json.dumps({"metric": "m", "value": value, "unit": "M reads/s", "ms_per_step": 1.0, "steps": 2, "warmup": 1, "config": {"workload": model},
                       "kernels_ms": {}, "kernels_ms_one_batch_in_flight": {}, "roofline": {"kernel": "k_seed", "traffic": 1, "stages": {}}, "counters_per_launch": {"steps": 1}})


def _orchestrate(monkeypatch, capsys, first, second, budget=1000.0):
    calls = []
    def fake_run(cmd, **kw):
        calls.append(cmd)
        r = first if len(calls) == 1 else second
        if isinstance(r, Exception):
            raise r
        return r
    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    rc = bench.with_human_like_leg(_args(budget), ["--steps", "2"])
    out = capsys.readouterr().out.strip().splitlines()
    return rc, out, calls


def test_both_children_one_line(monkeypatch, capsys):
    rc, out, calls = _orchestrate(monkeypatch, capsys, _Run(0, "noise\n" + _line(900.0) + "\n"), _Run(0, _line(390.0, "human") + "\n"))
    assert rc == 0 and len(out) == 1
    d = json.loads(out[0])
    assert d["value"] == 900.0 and d["value_human_like"] == 390.0 and d["human_like"]["workload"] == "human"
    assert calls[0][-1] == "--as-child" and "--as-child" in calls[1] and calls[1][calls[1].index("--genome-model") + 1] == "human"
    assert "--no-secondary" in calls[1] and "--no-cpu-baseline" in calls[1]


def test_budget_spent_or_second_child_failing_keeps_the_first_line(monkeypatch, capsys):
    rc, out, calls = _orchestrate(monkeypatch, capsys, _Run(0, _line(900.0)), _Run(0, _line(1.0)), budget=-1.0)     # (already over budget)
    d = json.loads(out[0])
    assert rc == 0 and len(out) == 1 and len(calls) == 1 and d["value"] == 900.0 and "budget" in d["human_like"]["skipped"]
    for second in (_Run(1, ""), _Run(0, "not json"), RuntimeError("timeout")):
        rc, out, calls = _orchestrate(monkeypatch, capsys, _Run(0, _line(901.0)), second)
        d = json.loads(out[0])
        assert rc == 0 and len(out) == 1 and d["value"] == 901.0 and "value_human_like" not in d and "failed" in d["human_like"]["skipped"]


def test_first_child_failing_is_the_run_failing(monkeypatch, capsys):
    rc, out, calls = _orchestrate(monkeypatch, capsys, _Run(3, ""), _Run(0, _line(1.0)))
    assert rc == 3 and out == [] and len(calls) == 1

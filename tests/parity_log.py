"""Parity bookkeeping for the GPU tests: every bound that is asserted is also WRITTEN, so a review reads numbers instead of dots.

``check(test, metric, measured, bound)`` appends one JSON line {test, metric, measured, bound, ok} to
``$MMSIM_PARITY_LOG`` (default ``gpurun_out/parity_r03.jsonl`` under the repo root) and then asserts ``measured < bound``.
The copy that is judged is committed under ``profiles/`` (tools/collect_parity.py turns the lines into one JSON document).
"""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.environ.get("MMSIM_PARITY_LOG") or os.path.join(ROOT, "gpurun_out", "parity_r03.jsonl")


def record(test, metric, measured, bound, ok=None):
    measured = float(measured.detach()) if hasattr(measured, "detach") else float(measured)
    bound = float(bound)
    ok = bool(measured < bound) if ok is None else bool(ok)
    try:
        os.makedirs(os.path.dirname(PATH), exist_ok=True)
        with open(PATH, "a") as fh:
            fh.write(json.dumps({"test": test, "metric": metric, "measured": measured, "bound": bound, "ok": ok}) + "\n")
    except OSError:
        pass          # a read-only tree must not turn a parity check into an I/O failure
    return ok


def check(test, metric, measured, bound):
    ok = record(test, metric, measured, bound)
    assert ok, f"{test}: {metric} = {float(measured):.6g} exceeds the bound {float(bound):.6g}"

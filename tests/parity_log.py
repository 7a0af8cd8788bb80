"""Parity bookkeeping for the GPU tests: every bound that is asserted is also WRITTEN, so a review reads numbers instead of dots.

``check(test, metric, measured, bound)`` appends one JSON line {test, metric, measured, bound, ok} to
``$MMSIM_PARITY_LOG`` (default ``gpurun_out/parity_r04.jsonl`` under the repo root) and then asserts ``measured < bound``.
The copy that is judged is committed under ``profiles/`` (tools/collect_parity.py turns the lines into one JSON document).
"""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.environ.get("MMSIM_PARITY_LOG") or os.path.join(ROOT, "gpurun_out", "parity_r04.jsonl")


_STAMP = None


def build_stamp():
    """What the rows of one session were measured on: the digest of csrc/ the library was built from (build.py's stamp file) and
    the library file's own hash -- collect_parity.py refuses a log that mixes two builds."""
    global _STAMP
    if _STAMP is None:
        import hashlib
        pkg = os.path.join(ROOT, "multimodalsimilar_amd")
        lib = os.environ.get("MMSIM_LIB") or os.path.join(pkg, "libmmsim_hip.so")
        try:
            src = open(os.path.join(pkg, ".libmmsim_hip.stamp")).read().strip()[:16]
        except OSError:
            src = "unknown"
        try:
            h = hashlib.sha256()
            with open(lib, "rb") as fh:
                for blk in iter(lambda: fh.read(1 << 20), b""):
                    h.update(blk)
            so = h.hexdigest()[:16]
        except OSError:
            so = "missing"
        _STAMP = {"csrc": src, "lib": so}
    return _STAMP


def start_session():
    """Called once per pytest session (tests/conftest.py): the log is truncated, so a file never mixes sessions."""
    try:
        os.makedirs(os.path.dirname(PATH), exist_ok=True)
        open(PATH, "w").close()
    except OSError:
        pass


def record(test, metric, measured, bound, ok=None):
    measured = float(measured.detach()) if hasattr(measured, "detach") else float(measured)
    bound = float(bound)
    ok = bool(measured < bound) if ok is None else bool(ok)
    try:
        os.makedirs(os.path.dirname(PATH), exist_ok=True)
        with open(PATH, "a") as fh:
            fh.write(json.dumps({"test": test, "metric": metric, "measured": measured, "bound": bound, "ok": ok, **build_stamp()}) + "\n")
    except OSError:
        pass          # a read-only tree must not turn a parity check into an I/O failure
    return ok


def check(test, metric, measured, bound):
    ok = record(test, metric, measured, bound)
    assert ok, f"{test}: {metric} = {float(measured):.6g} exceeds the bound {float(bound):.6g}"

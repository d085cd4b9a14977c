"""Measured parity margins, written next to the pass / fail result.

Every parity test that computes a worst-case error calls ``record(...)`` with the numbers it asserted on; at the end of the
pytest session ``tests/conftest.py`` writes them all to ``gpurun_out/parity_margins.json`` (``MSH_PARITY_MARGINS`` overrides
the path), which ``tools/gpu_final.sh`` copies to ``profiles/`` -- so the distance to each tolerance is tracked from round to
round instead of living in a log nobody keeps.  Test infrastructure only.
"""
import os

RECORDS: dict[str, dict] = {}


def record(label: str | None = None, **values) -> None:
    """Attach measured numbers to the running test (key = pytest node id, plus `label` when a test records more than once)."""
    node = os.environ.get("PYTEST_CURRENT_TEST", "unknown").split(" (")[0]
    key = node if label is None else f"{node}#{label}"
    clean = {}
    for k, v in values.items():
        if hasattr(v, "item"):
            v = v.item()
        clean[k] = round(v, 8) if isinstance(v, float) else v
    RECORDS.setdefault(key, {}).update(clean)


def dump(path: str) -> None:
    import json

    if not RECORDS:
        return
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    old = {}
    if os.path.exists(path) and os.environ.get("MSH_PARITY_MARGINS_APPEND") == "1":
        try:
            old = json.load(open(path)).get("tests", {})
        except Exception:
            old = {}
    old.update(RECORDS)
    doc = {"what": "worst-case figures the parity tests measured in this session, next to the tolerance they were held to "
                   "(tests/margins.py); tolerances: logits max-abs 5e-2, encoder rel-RMS 1e-2 / max-abs 6e-2, ids identical "
                   "where the oracle's top-1 margin exceeds 0.1",
           "tests": dict(sorted(old.items()))}
    with open(path, "w") as f:
        json.dump(doc, f, indent=1)

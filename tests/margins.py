"""Measured error next to its bar, for every parity assertion that goes through bound(): the whole -m gpu run leaves
gpurun_out/parity_margins.json = {test id: {tensor: [largest measured error, bar]}} behind (tests/conftest.py writes it at the
end of the session), so that a bar can be held at <= 10x what the kernels deliver and a regression in accuracy that still
passes is visible in the diff of that file (profiles/rNN_parity_margins.json)."""
import os

MARGINS = {}


def bound(err, bar, what):
    """assert err <= bar, and remember the largest err seen under this (test, what)."""
    test = os.environ.get("PYTEST_CURRENT_TEST", "?").split(" (")[0]
    slot = MARGINS.setdefault(test, {})
    key = str(what)
    prev = slot.get(key)
    if prev is None or not (err <= prev[0]):          # (NaN replaces anything)
        slot[key] = [float(err), float(bar)]
    assert err <= bar, f"{what}: {err:.3g} > {bar:.3g}"
    return err


def dump(path):
    import json
    if not MARGINS:
        return
    os.makedirs(os.path.dirname(path), exist_ok=True)
    worst = {t: max((v[0] / v[1] if v[1] > 0 else 0.0) for v in d.values()) for t, d in MARGINS.items()}
    with open(path, "w") as fh:
        json.dump({"convention": "max|gpu - oracle| / max|oracle| per tensor; value = [largest measured, bar]",
                   "tightest_fraction_of_bar": max(worst.values()), "tests": MARGINS}, fh, indent=0, sort_keys=True)

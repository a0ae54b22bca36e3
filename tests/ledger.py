"""Parity ledger (test infrastructure): every end-to-end gradient comparison whose gate is widened by a measured-sensitivity rule, and
every same-record comparison next to it, is appended here as one entry

    {test, scene, rollout, step, e2e_err, sensitivity, gate, same_record_adopt, same_record_forced, note}

`sensitivity` is measured on the fp64 ORACLE alone (its own gradient under a float32 rounding of its own x_new, or with its PD loop run
one iteration past its stopping rule: tests/records.py) — never on the engine. With DC_LEDGER=1 the entries of the process are merged into
a JSON file at exit (DC_LEDGER_PATH, default gpurun_out/parity_ledger.json — the only directory that travels back from the GPU box);
one full run is committed as profiles/r06_parity_ledger.json. tests/test_gpu_zz_parity_ledger.py (collected last) asserts the rules
over the entries of the run."""
import atexit
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENTRIES = []


def add(test, scene, rollout, e2e_err, sensitivity=None, gate=None, same_record_adopt=None, same_record_forced=None, step=None, note=None):
    def num(v):
        return None if v is None else float(v)
    ent = dict(test=test, scene=scene, rollout=int(rollout), step=None if step is None else int(step), e2e_err=num(e2e_err),
               sensitivity=num(sensitivity), gate=num(gate), same_record_adopt=num(same_record_adopt),
               same_record_forced=num(same_record_forced), note=note)
    ENTRIES.append(ent)
    return ent


def path():
    return os.environ.get("DC_LEDGER_PATH", os.path.join(ROOT, "gpurun_out", "parity_ledger.json"))


def _key(e):
    return (e["test"], e["scene"], e["rollout"], e["step"])


def flush():
    if os.environ.get("DC_LEDGER") != "1" or not ENTRIES:
        return
    p = path()
    merged = {}
    if os.path.exists(p):
        try:
            with open(p) as fh:
                for e in json.load(fh)["entries"]:
                    merged[_key(e)] = e
        except Exception:
            merged = {}
    for e in ENTRIES:
        merged[_key(e)] = e
    os.makedirs(os.path.dirname(p), exist_ok=True)
    ents = sorted(merged.values(), key=lambda e: (e["test"], e["scene"], e["rollout"], -1 if e["step"] is None else e["step"]))
    with open(p, "w") as fh:
        json.dump({"rule": "same-record errors <= 1e-4 flat; end to end <= max(1e-4, 1.0 x sensitivity) on the strict scenes, <= gate elsewhere; "
                           "sensitivity = the fp64 oracle's own gradient change under a float32 rounding of its x_new (or one more PD iteration)",
                   "entries": ents}, fh, indent=1)


atexit.register(flush)

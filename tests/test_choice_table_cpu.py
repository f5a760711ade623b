"""The persisted kernel-choice table (omnihd_amd/ops/policy.py::_ChoiceTable): a lookup that misses the process's
own table falls back to the committed file before anything is measured; misses are counted; the file round-trips."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]


def test_lookup_falls_back_to_the_persisted_table_and_counts_misses(tmp_path, monkeypatch):
    from omnihd_amd import ops
    path = tmp_path / "choices.json"
    key = ("fwd", (1, 1024, 160, 240), 1024, 3, 1)
    doc = {"conv": {json.dumps(list(map(lambda v: list(v) if isinstance(v, tuple) else v, key))): "hip"}, "wgrad": {}, "split": {}}
    path.write_text(json.dumps(doc))
    monkeypatch.setenv("OMNIHD_CHOICE_TABLE", str(path))
    monkeypatch.setattr(ops.policy, "_PERSISTED", {})
    monkeypatch.setattr(ops.policy, "_CHOICE_INFO", {"path": None, "sha256": None, "entries": 0, "misses": 0, "loaded": False})
    table = ops._ChoiceTable("conv")
    assert table.get(key + (3,)) == "hip" and table[key + (3,)] == "hip"              # any device index: the file has none
    assert table.get(("fwd", (1, 8, 8, 8), 8, 3, 1, 0)) is None
    info = ops.choice_table_info()
    assert info["entries"] == 1 and info["misses"] == 0 and len(info["sha256"]) == 64
    assert table.measured(("fwd", (1, 8, 8, 8), 8, 3, 1, 0), "miopen") == "miopen"
    assert ops.choice_table_info()["misses"] == 1
    # round trip: persisted + measured entries, device index dropped
    monkeypatch.setattr(ops.policy, "_CONV_CHOICE", table)
    out = tmp_path / "merged.json"
    n = ops.save_choice_table(str(out))
    merged = json.loads(out.read_text())
    assert n >= 2 and len(merged["conv"]) == 2 and set(merged["conv"].values()) == {"hip", "miopen"}
    assert all(len(json.loads(k)) == 5 for k in merged["conv"])


def test_committed_table_loads_when_present():
    from omnihd_amd import ops
    path = os.path.join(ROOT, "omnihd-scenes_amd", "kernel_choices", "gfx950.json")
    if os.path.exists(path):
        doc = json.load(open(path))
        assert set(doc) >= {"conv", "wgrad", "split"}
        for name in ("conv", "wgrad", "split"):
            for k, v in doc[name].items():
                assert isinstance(json.loads(k), list) and v in ("hip", "hip128x256", "miopen", "split")
        if os.environ.get("OMNIHD_CHOICE_TABLE") is None:
            # ... and the package FINDS it (the path is computed from the module's location: it moved when ops.py became a package)
            info = ops.choice_table_info()
            assert info["path"] == os.path.join("omnihd-scenes_amd", "kernel_choices", "gfx950.json"), info
            assert info["entries"] == sum(len(doc[n]) for n in ("conv", "wgrad", "split")), info

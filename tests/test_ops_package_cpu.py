"""omnihd_amd.ops is a package of parts by concern (round 6: ops.py had grown to 3 100 lines).  What keeps that safe without a GPU:
every global a part's code loads resolves in that part (a name left behind in another part would only fail on the GPU box, inside the
code path that uses it), parts import only EARLIER parts, and `ops.X` is the very object the owning part holds."""
import builtins
import dis
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]


def _global_loads(path):
    code = compile(open(path).read(), path, "exec")
    seen, stack = set(), [code]
    while stack:
        c = stack.pop()
        for ins in dis.get_instructions(c):
            if ins.opname in ("LOAD_GLOBAL", "LOAD_NAME"):
                seen.add(ins.argval)
        stack += [k for k in c.co_consts if hasattr(k, "co_code")]
    return seen


def test_every_global_a_part_loads_resolves_in_that_part():
    from omnihd_amd import ops
    assert len(ops.PARTS) == 12
    for part in ops.PARTS:
        missing = sorted(n for n in _global_loads(part.__file__) if not hasattr(part, n) and not hasattr(builtins, n))
        assert not missing, (part.__name__, missing)


def test_parts_import_only_earlier_parts_and_the_package_re_exports_their_objects():
    from omnihd_amd import ops
    names = [p.__name__.rsplit(".", 1)[1] for p in ops.PARTS]
    for k, part in enumerate(ops.PARTS):
        src = open(part.__file__).read()
        for later in names[k:]:
            assert f"from .{later} import" not in src, (part.__name__, later)
        assert src.count("\n") < 500, part.__name__                       # a part stays a readable file
    owners = {}
    for part in ops.PARTS:
        for n, v in vars(part).items():
            if getattr(v, "__module__", None) == part.__name__ and not n.startswith("__"):
                owners[n] = part
    assert len(owners) > 150
    for n, part in owners.items():
        assert getattr(ops, n) is getattr(part, n), n


def test_the_committed_choice_table_is_found_from_the_parts_location():
    from omnihd_amd import ops
    assert os.path.isdir(os.path.join(ops.policy._PKG_ROOT, "kernel_choices"))
    assert os.path.basename(ops.policy._PKG_ROOT) == "omnihd-scenes_amd"

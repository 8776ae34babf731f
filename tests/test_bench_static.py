"""bench.py cannot run without a GPU; its names can be checked without one: every name a function of bench.py reads and does
not bind itself must exist at module level (or be a builtin).  Catches the NameError that would otherwise cost a side run
-- or the line -- on the GPU box."""
import builtins
import os
import symtable

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_globals(table, found):
    for sym in table.get_symbols():
        if table.get_type() != "module" and sym.is_global() and sym.is_referenced():
            found.add((sym.get_name(), table.get_name(), table.get_lineno()))
    for child in table.get_children():
        _free_globals(child, found)


def _module_names(table):
    return {s.get_name() for s in table.get_symbols() if s.is_assigned() or s.is_imported() or s.is_namespace()}


def _check(path):
    src = open(path).read()
    top = symtable.symtable(src, path, "exec")
    defined = _module_names(top) | set(dir(builtins)) | {"__file__", "__name__"}
    used = set()
    _free_globals(top, used)
    return sorted((n, fn, ln) for n, fn, ln in used if n not in defined)


def test_every_global_name_bench_reads_is_defined():
    assert _check(os.path.join(ROOT, "bench.py")) == []
    for part in ("common", "stream", "counters", "side", "parity"):   # bench.py's parts (benchlib/)
        assert _check(os.path.join(ROOT, "benchlib", part + ".py")) == [], part


def test_the_same_for_the_entry_points_and_the_host_modules():
    for rel in ("__graft_entry__.py", "dftpav_amd/capi.py", "dftpav_amd/distributed.py", "dftpav_amd/optimizer.py", "dftpav_amd/scenarios.py",
                "scripts/fuzz_reference_order.py", "scripts/profile_phases.py", "scripts/ref_order_time.py"):
        p = os.path.join(ROOT, rel)
        if os.path.exists(p):
            assert _check(p) == [], rel

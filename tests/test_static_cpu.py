"""Cheap static checks (CPU): every python module of the product / oracle parses and has no
name that is loaded but never bound (catches NameErrors on rarely taken multi-rank paths)."""
import ast
import builtins
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _undefined(path):
    t = ast.parse(open(path).read())
    bound = set(dir(builtins)) | {"__file__", "__name__", "__doc__"}
    for n in ast.walk(t):
        if isinstance(n, (ast.Import, ast.ImportFrom)):
            for a in n.names:
                bound.add((a.asname or a.name).split(".")[0])
        elif isinstance(n, (ast.FunctionDef, ast.ClassDef, ast.AsyncFunctionDef)):
            bound.add(n.name)
        elif isinstance(n, ast.Name) and isinstance(n.ctx, (ast.Store, ast.Del)):
            bound.add(n.id)
        elif isinstance(n, ast.ExceptHandler) and n.name:
            bound.add(n.name)
        elif isinstance(n, ast.arg):
            bound.add(n.arg)
    return [(n.lineno, n.id) for n in ast.walk(t) if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load) and n.id not in bound]


def test_no_unbound_names():
    bad = {}
    for top in ("u2pl_amd", "oracle", "tools"):
        for d, _, fs in os.walk(os.path.join(ROOT, top)):
            for f in fs:
                if f.endswith(".py"):
                    u = _undefined(os.path.join(d, f))
                    if u:
                        bad[os.path.join(d, f)] = u
    for f in ("bench.py", "train_semi.py", "train_sup.py", "__graft_entry__.py"):
        u = _undefined(os.path.join(ROOT, f))
        if u:
            bad[f] = u
    assert not bad, bad

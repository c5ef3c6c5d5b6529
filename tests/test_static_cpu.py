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


def test_call_sites_match_the_c_header():
    """every `call("u2pl_...", args)` / `query("u2pl_...", args)` in the product passes exactly the number of
    arguments include/u2pl_hip.h declares (call() appends the stream itself) -- a signature change in the C ABI
    that misses a Python call site would otherwise only fail on the GPU."""
    import sys
    sys.path.insert(0, ROOT)
    from u2pl_amd._lib import parse_header

    decls = parse_header()
    bad = []
    files = [os.path.join(d, f) for top in ("u2pl_amd", "tools") for d, _, fs in os.walk(os.path.join(ROOT, top))
             for f in fs if f.endswith(".py")] + [os.path.join(ROOT, "eval.py")]
    seen = set()
    for path in files:
        tree = ast.parse(open(path).read())
        for n in ast.walk(tree):
            if not (isinstance(n, ast.Call) and n.args and isinstance(n.args[0], ast.Constant)
                    and isinstance(n.args[0].value, str) and n.args[0].value.startswith("u2pl_")):
                continue
            fn = n.func.attr if isinstance(n.func, ast.Attribute) else getattr(n.func, "id", "")
            if fn not in ("call", "query"):
                continue
            name = n.args[0].value
            if name not in decls:
                bad.append((path, n.lineno, name, "not declared in the header"))
                continue
            seen.add(name)
            if any(isinstance(a, ast.Starred) for a in n.args):
                continue   # *prob_strides etc.: length known only at run time
            want = len(decls[name][1]) - (1 if fn == "call" else 0)   # call() appends hipStream_t
            has_stream = decls[name][2] and decls[name][2][-1] == "stream"
            if fn == "query" and has_stream:
                want -= 0
            got = len(n.args) - 1
            if got != want:
                bad.append((os.path.relpath(path, ROOT), n.lineno, name, f"passes {got} args, header wants {want}"))
    assert not bad, bad
    assert len(seen) > 40      # the walk really found the call sites

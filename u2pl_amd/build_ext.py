"""Build libu2pl_hip.so (gfx950) in-tree with hipcc.  No torch headers: the
library is a plain C-ABI shared object (include/u2pl_hip.h)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib", "libu2pl_hip.so")
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
    "-fvisibility=hidden", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-Wall", "-Wno-unused-function",
]


# per-file switches.  igemm_ws.hip: no SLP vectorisation -- it fuses the split's subtractions into v_pk_add_f32 (+ a
# hazard nop each), which costs more beside matrix instructions than the two plain adds it replaces
EXTRA = {"igemm_ws.hip": ["-fno-slp-vectorize"]}


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True, variant=None, defs=()):
    """variant / defs: a second build of the same sources with extra -D switches (debug instrumentation), written to
    lib/variants/libu2pl_hip_<variant>.so and loaded through U2PL_LIB_PATH; the product library has neither."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(os.path.join(HERE, "lib"), exist_ok=True)
    objdir = os.path.join(HERE, "lib", "obj" + ("_" + variant if variant else ""))
    LIB = globals()["LIB"]
    if variant:
        os.makedirs(os.path.join(HERE, "lib", "variants"), exist_ok=True)
        LIB = os.path.join(HERE, "lib", "variants", "libu2pl_hip_%s.so" % variant)
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(ROOT, "include", "u2pl_hip.h"))
    objs, procs = [], []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            cmd = [hipcc] + FLAGS + EXTRA.get(os.path.basename(src), []) + ["-D" + d for d in defs] + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on " + src)
    if procs or force or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    # python -m u2pl_amd.build_ext [--force] [--variant NAME -DX -DY ...]
    var = sys.argv[sys.argv.index("--variant") + 1] if "--variant" in sys.argv else None
    print(build(force="--force" in sys.argv, variant=var, defs=[a[2:] for a in sys.argv if a.startswith("-D")]))

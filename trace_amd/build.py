"""Build libtrace_amd.so (gfx950 only) in-tree with hipcc.

    python -m trace_amd.build [--force]

Each csrc/*.hip|*.cpp is compiled to an object (skipped when up to date) and
linked into trace_amd/lib/libtrace_amd.so.  Files listed in EXACT_F32 are built
with -ffp-contract=off so that their f32 arithmetic reproduces the reference's
CPU rounding sequence.
"""
import concurrent.futures
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libtrace_amd.so")

EXACT_F32 = {"roi_ops.hip", "nms.hip", "box_ops.hip"}
COMMON = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
          "-Wno-unused-variable"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def _newest_header():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(HERE, "..", "include", "trace_amd.h"))
    return max(os.path.getmtime(h) for h in hs)


def _compile(src, force):
    obj = os.path.join(OBJ, src + ".o")
    path = os.path.join(CSRC, src)
    if (not force and os.path.exists(obj) and os.path.getmtime(obj) >= os.path.getmtime(path)
            and os.path.getmtime(obj) >= _newest_header()):
        return obj, False
    cmd = [_hipcc()] + COMMON
    if src in EXACT_F32:
        cmd.append("-ffp-contract=off")
    if src.endswith(".hip"):
        cmd += ["-x", "hip"]
    cmd += ["-c", path, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, " ".join(cmd), r.stderr[-8000:]))
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj, True


def build_lib(force=False, jobs=8):
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    srcs = _sources()
    with concurrent.futures.ThreadPoolExecutor(max_workers=jobs) as ex:
        res = list(ex.map(lambda s: _compile(s, force), srcs))
    objs = [o for o, _ in res]
    changed = any(c for _, c in res)
    if changed or not os.path.exists(LIB):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s" % r.stderr[-8000:])
    return LIB


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv))

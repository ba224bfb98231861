#!/usr/bin/env python3
"""Diagnostic builds of libtrace_amd.so: the same sources with extra -D flags on igemm.hip, linked to
trace_amd/lib/libtrace_amd_<name>.so (load it with TRACE_AMD_LIB=<path>).

    python tools/build_variant.py settle8 -DTRACE_SETTLE_SLEEP=8
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from trace_amd import build as b  # noqa: E402


def main():
    name, flags = sys.argv[1], sys.argv[2:]
    b.build_lib()
    obj = os.path.join(b.OBJ, "igemm.hip.%s.o" % name)
    cmd = [b._hipcc()] + b.COMMON + flags + ["-x", "hip", "-c", os.path.join(b.CSRC, "igemm.hip"), "-o", obj]
    subprocess.run(cmd, check=True)
    objs = [os.path.join(b.OBJ, s + ".o") for s in b._sources() if s != "igemm.hip"] + [obj]
    out = os.path.join(b.LIBDIR, "libtrace_amd_%s.so" % name)
    subprocess.run([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs, check=True)
    print(out)


if __name__ == "__main__":
    main()

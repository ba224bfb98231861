#!/usr/bin/env python3
"""Scan gfx950 assembly (hipcc --save-temps) for LDS reads still in flight when their wave arrives at an s_barrier:
a ds_read issued before the barrier whose `s_waitcnt lgkmcnt` only comes after it.  Behind such a barrier another wave
may already overwrite the bytes (LDS-DMA refill of the ring stage) while the read has not executed yet.

    python tools/scan_barrier_hazard.py file.s [...]
Linear scan per function (branches are not followed): a report is a place to look at, not a proof."""
import re
import sys


def scan(path):
    fn = None
    pending = 0          # ds_read instructions issued since the last lgkmcnt(0)-equivalent wait
    hits = {}
    for ln, line in enumerate(open(path), 1):
        s = line.strip()
        m = re.match(r"^([A-Za-z_][\w$.]*):", s)
        if m and not s.startswith(".L"):
            fn = m.group(1)
            pending = 0
            continue
        if s.startswith(".LBB"):
            continue
        op = s.split()[0] if s else ""
        if op.startswith("ds_read") or op.startswith("ds_load"):
            pending += 1
        elif op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", s)
            if m:
                pending = min(pending, int(m.group(1)))
        elif op == "s_barrier":
            if pending:
                hits.setdefault(fn, []).append((ln, pending))
        elif op in ("s_endpgm",):
            pending = 0
    return hits


if __name__ == "__main__":
    bad = 0
    for p in sys.argv[1:]:
        for fn, lst in scan(p).items():
            bad += 1
            print("%s: %s: %d barrier(s) with LDS reads in flight, first at line %d (%d pending)" % (p, fn, len(lst), lst[0][0], lst[0][1]))
    print("%d function(s) flagged" % bad)
    sys.exit(1 if bad else 0)

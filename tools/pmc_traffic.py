#!/usr/bin/env python3
"""HBM traffic of the igemm kernel family from two rocprofv3 --pmc passes of bench.py
(FETCH_SIZE and WRITE_SIZE, one pass each: they do not fit the 4 TCC slots together).

    python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>

Corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): both counters are in KiB; on gfx950
FETCH_SIZE tallies the 128-B requests of wide coalesced reads (16 B per lane: every load of these kernels,
LDS-DMA included) at 64 B, so it is doubled; WRITE_SIZE is exact for 16-B-per-lane stores."""
import csv
import json
import sys


FAMILY = ("igemm2_bf16_kernel", "igemm3_bf16_kernel", "tconv3_bf16_kernel", "stem16_bf16_kernel", "wsgemm_bf16_kernel", "sconv9_bf16_kernel")


def per_launch(path, counter):
    tot, disp = 0.0, set()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and any(k in r["Kernel_Name"] for k in FAMILY):
            tot += float(r["Counter_Value"])
            disp.add(r["Dispatch_Id"])
    return tot / max(len(disp), 1), len(disp)


fetch_kib, n1 = per_launch(sys.argv[1], "FETCH_SIZE")
write_kib, n2 = per_launch(sys.argv[2], "WRITE_SIZE")
out = dict(kernel="trace::{igemm2,igemm3,tconv3,sconv9,stem16}_bf16_kernel<*>", launches=[n1, n2], fetch_size_kib_per_launch=fetch_kib,
           write_size_kib_per_launch=write_kib, fetch_correction=2.0,
           traffic_bytes_per_launch=(2.0 * fetch_kib + write_kib) * 1024.0,
           source="rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-trace`")
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out))

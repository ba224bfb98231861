#!/usr/bin/env python3
"""Per-step kernel breakdown of the STEADY state of bench.py from a rocprofv3 kernel trace.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/steady -o steady -- \
        python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-trace
    python tools/steady_breakdown.py gpurun_out/steady/steady_kernel_trace.csv profiles/rNN_steady_step_breakdown.json

A timed step starts with one `upload_slots_kernel` launch (the step's host inputs); the window is the last `nsteps`
whole steps before the final one, so that warm-up, capture passes and the parameter upload are outside it.
"""
import collections
import csv
import json
import re
import sys


def short(name):
    name = re.sub(r"\(.*", "", name)
    return name[:96]


def main(path, out, nsteps=7):
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if "upload_slots_kernel" in r[2]]
    if len(marks) < nsteps + 2:
        raise SystemExit("only %d step markers in the trace" % len(marks))
    lo, hi = marks[-nsteps - 1], marks[-1]
    win = rows[lo:hi]
    per = collections.OrderedDict()
    for s, e, k in win:
        d = per.setdefault(short(k), [0, 0.0])
        d[0] += 1
        d[1] += (e - s) / 1e3
    busy = sum(v[1] for v in per.values())
    span = (rows[hi][0] - rows[lo][0]) / 1e3
    copies = sum(v[0] for k, v in per.items() if "copyBuffer" in k or "fillBuffer" in k)
    torchk = sum(v[0] for k, v in per.items() if "at::native" in k or k.strip() == "void")
    res = dict(window="%d consecutive timed steps (between upload_slots_kernel launches) of `rocprofv3 --kernel-trace -- "
                      "python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-trace`" % nsteps,
               kernels_per_step=len(win) / nsteps, busy_us_per_step=busy / nsteps, span_us_per_step=span / nsteps,
               copy_or_fill_nodes_per_step=copies / nsteps, torch_kernels_per_step=torchk / nsteps,
               per_kernel=[dict(kernel=k, launches_per_step=v[0] / nsteps, us_per_step=v[1] / nsteps)
                           for k, v in sorted(per.items(), key=lambda kv: -kv[1][1])])
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: v for k, v in res.items() if k != "per_kernel"}))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 7)

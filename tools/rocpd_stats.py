#!/usr/bin/env python3
"""Per-kernel statistics of a rocprofv3 results database (rocpd sqlite): `python tools/rocpd_stats.py file.db [last_n_steps marker]`
prints name, calls, total us, average us, share - the --stats summary as CSV (template arguments kept)."""
import sqlite3
import sys


def stats(path, where=""):
    db = sqlite3.connect(path)
    rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels %s "
                      "group by name order by sum(duration) desc" % where).fetchall()
    tot = sum(r[2] for r in rows) or 1
    return [(r[0], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / tot) for r in rows]


if __name__ == "__main__":
    print("name,calls,total_us,avg_us,min_us,max_us,percent")
    for r in stats(sys.argv[1]):
        print('"%s",%d,%.1f,%.2f,%.2f,%.2f,%.2f' % r)

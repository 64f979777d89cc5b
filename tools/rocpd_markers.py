"""Summarise the roctx ranges of a rocprofv3 --marker-trace run (rocpd SQLite): count and total duration per range name.
usage: python tools/rocpd_markers.py gpurun_out/prof/mark_results.db      (run the workload with TE_ROCTX=1)"""
import collections
import re
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    tables = [r[0] for r in db.execute("select name from sqlite_master where type in ('table', 'view')")]
    cand = [t for t in tables if re.search(r'region|marker|roctx', t, re.I)]
    agg = collections.defaultdict(lambda: [0, 0])
    used = None
    for t in cand:
        cols = [r[1] for r in db.execute(f'pragma table_info({t})')]
        name = next((c for c in cols if c in ('name', 'message', 'msg', 'region_name')), None)
        if not name or 'start' not in cols or 'end' not in cols:
            continue
        rows = db.execute(f'select {name}, start, end from {t}').fetchall()
        rows = [r for r in rows if r[0] and str(r[0]).startswith('te:')]
        if not rows:
            continue
        used = t
        for n, s, e in rows:
            op = str(n).split(' ')[0]
            agg[op][0] += 1
            agg[op][1] += (e - s)
        break
    if used is None:
        print('no te:* roctx ranges found; tables:', ', '.join(tables))
        return
    print(f'roctx ranges from table {used} (host-side duration of the wrapper call, i.e. launch cost, not kernel time)')
    print(f'{"range":32s} {"count":>8s} {"total_ms":>10s} {"avg_us":>9s}')
    for op, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f'{op:32s} {c:8d} {t / 1e6:10.3f} {t / c / 1e3:9.1f}')


if __name__ == '__main__':
    main(sys.argv[1])

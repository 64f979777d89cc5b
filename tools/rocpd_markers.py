"""Summarise the roctx ranges of a rocprofv3 --marker-trace run (rocpd SQLite): count and total duration per range name.
usage: python tools/rocpd_markers.py gpurun_out/prof/mark_results.db      (run the workload with TE_ROCTX=1)"""
import collections
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    tables = [r[0] for r in db.execute("select name from sqlite_master where type in ('table', 'view')") if '_0000' not in r[0]]
    agg = collections.defaultdict(lambda: [0, 0])
    used = None
    # rocprofiler-sdk layout: view `regions`, category MARKER_CORE_RANGE_API, the roctx message inside the `extdata` JSON
    if 'regions' in tables and 'extdata' in [r[1] for r in db.execute('pragma table_info(regions)')]:
        import json
        for ext, s_, e_ in db.execute("select extdata, start, end from regions where category like 'MARKER%'"):
            try:
                msg = json.loads(ext).get('message', '')
            except (ValueError, TypeError):
                continue
            if msg.startswith('te:'):
                used = 'regions.extdata'
                op = msg.split(' ')[0]
                agg[op][0] += 1
                agg[op][1] += (e_ - s_)
    for t in (() if used else ('regions', 'regions_and_samples', 'rocpd_region') + tuple(tables)):
        if t not in tables:
            continue
        cols = [r[1] for r in db.execute(f'pragma table_info({t})')]
        if 'start' not in cols or 'end' not in cols:
            continue
        for name in [c for c in cols if c in ('name', 'message', 'msg', 'region_name', 'extdata', 'args')]:
            try:
                rows = db.execute(f"select {name}, start, end from {t} where {name} like 'te:%'").fetchall()
            except sqlite3.Error:
                continue
            if rows:
                used = f'{t}.{name}'
                for n, s, e in rows:
                    op = str(n).split(' ')[0]
                    agg[op][0] += 1
                    agg[op][1] += (e - s)
                break
        if used:
            break
    if used is None:
        print('no te:* roctx ranges found')
        for t in ('regions', 'region_args', 'rocpd_region', 'rocpd_string'):
            if t in tables:
                cols = [r[1] for r in db.execute(f'pragma table_info({t})')]
                print(t, cols)
                for r in db.execute(f'select * from {t} limit 5'):
                    print('   ', r)
        return
    print(f'roctx ranges from {used} (host-side duration of the wrapper call, i.e. launch cost, not kernel time)')
    print(f'{"range":32s} {"count":>8s} {"total_ms":>10s} {"avg_us":>9s}')
    for op, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f'{op:32s} {c:8d} {t / 1e6:10.3f} {t / c / 1e3:9.1f}')


if __name__ == '__main__':
    main(sys.argv[1])

"""Summarise a rocprofv3 (rocpd SQLite) kernel trace as a per-kernel stats table (like --stats CSV).
usage: python tools/rocpd_stats.py gpurun_out/prof/bench_results.db > profiles/<name>.txt"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    return name if len(name) <= 110 else name[:107] + '...'


def main(path):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    name_col = 'name' if 'name' in cols else [c for c in cols if 'name' in c][0]
    rows = db.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                      f"from kernels group by {name_col} order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    print(f'{"kernel":112s} {"calls":>6s} {"total_ms":>10s} {"avg_us":>10s} {"min_us":>9s} {"max_us":>9s} {"%":>6s}')
    for n, c, t, a, mn, mx in rows[:60]:
        print(f'{short(n):112s} {c:6d} {t / 1e6:10.3f} {a / 1e3:10.1f} {mn / 1e3:9.1f} {mx / 1e3:9.1f} {100 * t / total:6.2f}')
    print(f'TOTAL kernel time {total / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches, {len(rows)} distinct kernels')


if __name__ == '__main__':
    main(sys.argv[1])

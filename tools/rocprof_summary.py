"""Summarise a rocprofv3 results database (sqlite, the only output format on this image).

    python tools/rocprof_summary.py <dir or *_results.db> [--pmc]

kernel trace : calls / avg / min / max / share per kernel   (rocprofv3 --kernel-trace --stats)
--pmc        : per-kernel average of every collected counter per dispatch (rocprofv3 --pmc ...)
--timeline   : start / end of every dispatch (ns from the first one), ascending
"""
import glob
import os
import sqlite3
import sys


def find_db(path):
    if os.path.isfile(path):
        return path
    c = sorted(glob.glob(os.path.join(path, '**', '*_results.db'), recursive=True))
    if not c:
        raise SystemExit('no *_results.db under %s' % path)
    return c[-1]


def names(con, kind):
    return [r[0] for r in con.execute("select name from sqlite_master where type=?", (kind,))]


def first(cands, prefix):
    for n in cands:
        if n.startswith(prefix):
            return n
    return None


def kernel_stats(con):
    tabs = names(con, 'table') + names(con, 'view')
    disp, sym = first(tabs, 'rocpd_kernel_dispatch'), first(tabs, 'rocpd_info_kernel_symbol')
    if not disp or not sym:
        raise SystemExit('unexpected schema: %s' % tabs)
    rows = con.execute('select s.kernel_name, count(*), avg(d.end - d.start), min(d.end - d.start), max(d.end - d.start), '
                       'sum(d.end - d.start) from %s d join %s s on d.kernel_id = s.id group by s.kernel_name '
                       'order by 6 desc' % (disp, sym)).fetchall()
    tot = sum(r[5] for r in rows) or 1
    for n, c, a, lo, hi, s in rows:
        print('%-62s calls %6d avg %9.2f us min %8.2f max %8.2f %6.1f%%' % (n[:62], c, a / 1e3, lo / 1e3, hi / 1e3,
                                                                             100.0 * s / tot))


def pmc(con):
    views = names(con, 'view') + names(con, 'table')
    v = first(views, 'counters_collection')
    if not v:
        raise SystemExit('no counters_collection view: %s' % views)
    cols = [r[1] for r in con.execute('pragma table_info(%s)' % v)]
    kcol = 'kernel_name' if 'kernel_name' in cols else 'name'
    ccol = 'counter_name' if 'counter_name' in cols else 'counter'
    vcol = 'value' if 'value' in cols else 'counter_value'
    dcol = 'dispatch_id' if 'dispatch_id' in cols else 'id'
    rows = con.execute('select %s, %s, sum(%s), count(distinct %s) from %s group by 1, 2 order by 1, 2'
                       % (kcol, ccol, vcol, dcol, v)).fetchall()
    out = {}
    for k, c, s, n in rows:
        out.setdefault(k, []).append('%s=%.4g' % (c, s / max(n, 1)))
    for k, items in out.items():
        print('%-46s %s' % (k[:46], '  '.join(items)))


def timeline(con):
    """--timeline: every kernel dispatch as 'start_ns end_ns name' (ascending start), for gap analysis."""
    tabs = names(con, 'table') + names(con, 'view')
    disp, sym = first(tabs, 'rocpd_kernel_dispatch'), first(tabs, 'rocpd_info_kernel_symbol')
    rows = con.execute('select d.start, d.end, s.kernel_name from %s d join %s s on d.kernel_id = s.id order by d.start'
                       % (disp, sym)).fetchall()
    t0 = rows[0][0] if rows else 0
    for a, b, n in rows:
        print('%d %d %s' % (a - t0, b - t0, n.split('(')[0][:40]))


def main():
    if len(sys.argv) < 2:
        raise SystemExit(__doc__)
    con = sqlite3.connect(find_db(sys.argv[1]))
    if '--pmc' in sys.argv:
        pmc(con)
    elif '--timeline' in sys.argv:
        timeline(con)
    else:
        kernel_stats(con)


if __name__ == '__main__':
    main()

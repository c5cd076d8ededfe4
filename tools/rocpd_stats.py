"""Per-kernel summary of a rocprofv3 (ROCm 7.2) run that wrote its default rocpd SQLite database:

    python tools/rocpd_stats.py <results.db> [name filter] > profiles/<name>_kernel_stats.csv

Columns: kernel, calls, total us, average us, min us, max us, percent -- the same figures `rocprofv3 --kernel-trace --stats` prints
as kernel_stats.csv when asked for CSV output (demangled, template arguments kept, parameter lists cut)."""
import re
import sqlite3
import subprocess
import sys


def demangle(names):
    try:
        out = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True, check=True).stdout.split('\n')
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}


def main():
    db = sqlite3.connect(sys.argv[1])
    flt = sys.argv[2] if len(sys.argv) > 2 else ''
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
    ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
    rows = db.execute('select s.kernel_name, count(*), sum(d.end - d.start) / 1e3, avg(d.end - d.start) / 1e3, min(d.end - d.start) / 1e3, '
                      'max(d.end - d.start) / 1e3 from {} d join {} s on d.kernel_id = s.id group by s.kernel_name order by 3 desc'.format(kd, ks)).fetchall()
    dm = demangle([r[0].replace('.kd', '') for r in rows])
    total = sum(r[2] for r in rows)
    print('"Name","Calls","TotalDurationUs","AverageUs","MinUs","MaxUs","Percentage"')
    for r in rows:
        name = dm[r[0].replace('.kd', '')]
        name = re.sub(r'^void ', '', name)
        name = re.sub(r'\(anonymous namespace\)::', '', name)
        cut = name.find('(')
        name = name[:cut] if cut > 0 else name
        if flt and flt not in name:
            continue
        print('"{}",{},{:.1f},{:.2f},{:.2f},{:.2f},{:.2f}'.format(name, r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / total))


if __name__ == '__main__':
    main()

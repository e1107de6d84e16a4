#!/usr/bin/env python3
"""Export the judged summaries from a rocprofv3 run database (rocpd sqlite, the default output of ROCm 7.2's rocprofv3).

usage: rocpd_export.py stats  RUN.db OUT.csv     # per-kernel calls / total / average (the --stats summary)
       rocpd_export.py pmc    RUN.db OUT.csv     # one row per dispatch: Kernel_Name, Counter_Name, Counter_Value
"""

import csv
import sqlite3
import sys


def main():
    mode, db, out = sys.argv[1:4]
    con = sqlite3.connect(db)
    with open(out, 'w', newline='') as f:
        w = csv.writer(f)
        if mode == 'stats':
            w.writerow(['Name', 'Calls', 'TotalDurationUs', 'AverageUs', 'Percentage'])
            for row in con.execute('select name, total_calls, total_duration, average, percentage from top_kernels order by total_duration desc'):
                w.writerow(row)
        elif mode == 'pmc':
            w.writerow(['Dispatch_Id', 'Kernel_Name', 'Counter_Name', 'Counter_Value'])
            for row in con.execute('select dispatch_id, kernel_name, counter_name, value from counters_collection order by dispatch_id'):
                w.writerow(row)
        else:
            raise SystemExit(__doc__)


if __name__ == '__main__':
    main()

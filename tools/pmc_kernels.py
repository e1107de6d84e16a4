#!/usr/bin/env python3
"""Per-kernel sums of the counters of one `rocprofv3 --pmc ... --kernel-trace` run (rocpd sqlite): one line per kernel family.
usage: pmc_kernels.py RUN.db [substring ...]"""
import collections
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
want = sys.argv[2:]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for k, c, v in con.execute('select kernel_name, counter_name, value from counters_collection'):
    name = k.split('(')[0].replace('void ', '')
    if want and not any(w in name for w in want):
        continue
    acc[name][c] += v
for k, d in con.execute('select name, duration from kernels'):
    name = k.split('(')[0].replace('void ', '')
    if name in acc:
        cnt[name] += 1
        acc[name]['duration_ns'] += d
for name, cs in sorted(acc.items(), key=lambda kv: -kv[1].get('duration_ns', 0)):
    n = max(cnt[name], 1)
    print(f'{name} launches={cnt[name]} avg_us={cs.get("duration_ns", 0) / n / 1e3:.1f}: ' + ', '.join(f'{c}={v / n:.4g}' for c, v in sorted(cs.items()) if c != 'duration_ns'))

#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs of the same command) into an HBM-traffic summary.

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md §HBM: both counters are in KiB; on gfx950 FETCH_SIZE reports
half of the bytes of wide (16 B/lane) coalesced streaming reads, so it is doubled; WRITE_SIZE is exact for 16 B/lane stores.
The conv kernels read 16-byte units per lane (buffer_load_dwordx4 / global_load_lds_dwordx4) but in 544-byte row runs, i.e. not a
pure stream, so the doubled figure is an UPPER bound for them; the raw figure is kept alongside.

usage: hbm_traffic.py FETCH.csv WRITE.csv FORWARDS OUT.json
"""

import collections
import csv
import json
import sys


def load(path):
    tot, cnt = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(path)):
        name = r['Kernel_Name']
        fam = 'rsa::conv_kernel*' if 'conv_kernel' in name else name.split('(')[0][:60]
        tot[fam] += float(r['Counter_Value'])
        cnt[fam] += 1
    return tot, cnt


def main():
    fetch_csv, write_csv, forwards, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    f, fc = load(fetch_csv)
    w, wc = load(write_csv)
    fam = 'rsa::conv_kernel*'
    launches = fc[fam] / forwards
    res = {
        'kernel': fam,
        'forwards_profiled': forwards,
        'launches_per_forward': launches,
        'fetch_raw_GB_per_forward': f[fam] * 1024 / 1e9 / forwards,
        'fetch_x2_GB_per_forward': 2 * f[fam] * 1024 / 1e9 / forwards,
        'write_GB_per_forward': w[fam] * 1024 / 1e9 / forwards,
    }
    res['traffic_GB_per_forward'] = res['fetch_x2_GB_per_forward'] + res['write_GB_per_forward']
    res['traffic_GB_per_launch'] = res['traffic_GB_per_forward'] / launches
    res['note'] = 'FETCH_SIZE doubled per the gfx950 correction (upper bound for the 544-byte row runs of the halo fill); WRITE_SIZE exact'
    json.dump(res, open(out, 'w'), indent=1)
    print(json.dumps(res))


if __name__ == '__main__':
    main()

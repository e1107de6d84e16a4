#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs of the same command) into an HBM-traffic summary.

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md §HBM: both counters are in KiB; on gfx950 FETCH_SIZE reports
half of the bytes of wide (16 B/lane) coalesced streaming reads, so it is doubled; WRITE_SIZE is exact for 16 B/lane stores.
The conv kernels read 16-byte units per lane (LDS-DMA / buffer loads) in 544-byte row runs, i.e. not a pure stream, so the doubled
figure is an UPPER bound for them; the raw figure is kept alongside.  The counters sit on the L2's memory side: Infinity-Cache hits
are counted as traffic.

The summary is stamped with the source hash of the library it was collected on (libresselt_amd.so.srchash); bench.py reports
`roofline.traffic` only while that hash matches the library it has loaded.

usage: hbm_traffic.py FETCH.csv WRITE.csv FORWARDS OUT.json [ALGORITHMIC_GB_PER_FORWARD]
"""

import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def family(name: str) -> str:
    name = name.replace('void ', '')
    if 'conv_ring<' in name:  # conv_ring<SHAPE, UP, OUTK, HM, FMT, PROD>
        import re

        a = [int(v) for v in re.findall(r'-?\d+', name.split('conv_ring<')[1].split('>')[0])]
        a += [0] * (6 - len(a))
        shape, fmt, prod = a[0], a[4], (a[5] or 3)
        cout = {2: 'Cout<=32', 3: 'Cout 33..48', 1: 'Cout 49..64'}.get(shape, '?')
        return f'rsa::conv_ring<{shape},...,{"f16" if fmt else "bf16"},{prod}> ({cout}, {prod} product{"s" if prod > 1 else ""})'
    if 'conv_kernel_pp' in name:
        return 'rsa::conv_kernel_pp'
    if 'conv_kernel' in name:
        return 'rsa::conv_kernel'
    return name.split('(')[0][:60]


def load(path):
    tot, cnt = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(path)):
        fam = family(r['Kernel_Name'])
        tot[fam] += float(r['Counter_Value'])
        cnt[fam] += 1
    return tot, cnt


def main():
    fetch_csv, write_csv, forwards, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    algo = float(sys.argv[5]) if len(sys.argv) > 5 else 258.5  # layer-wise bf16 model of SURVEY.md 8d for the C2 frame
    f, fc = load(fetch_csv)
    w, _ = load(write_csv)
    fams = {}
    for fam in f:
        if 'conv' not in fam:
            continue
        launches = fc[fam] / forwards
        fx2 = 2 * f[fam] * 1024 / forwards
        wr = w.get(fam, 0.0) * 1024 / forwards
        fams[fam] = {'kernel': fam, 'launches_per_forward': launches, 'fetch_raw_GB_per_forward': f[fam] * 1024 / 1e9 / forwards,
                     'fetch_x2_GB_per_forward': fx2 / 1e9, 'write_GB_per_forward': wr / 1e9, 'traffic_bytes_per_launch': (fx2 + wr) / launches}  # fmt: skip
    total = sum(v['fetch_x2_GB_per_forward'] + v['write_GB_per_forward'] for v in fams.values())
    stamp = os.path.join(ROOT, 'resselt_amd', 'libresselt_amd.so.srchash')
    res = {
        'srchash': open(stamp).read().strip() if os.path.exists(stamp) else None,
        'forwards_profiled': forwards,
        'kernels': sorted(fams.values(), key=lambda v: -(v['fetch_x2_GB_per_forward'] + v['write_GB_per_forward'])),
        'traffic_GB_per_forward': total,
        'algorithmic_GB_per_forward': algo,
        'traffic_over_algorithmic': total / algo,
        'note': 'FETCH_SIZE doubled per the gfx950 correction (upper bound for the 544-byte row runs of the halo fill); WRITE_SIZE exact; conv kernels only',
    }
    res['dominant'] = res['kernels'][0] if res['kernels'] else {}
    json.dump(res, open(out, 'w'), indent=1)
    print(json.dumps(res)[:1500])


if __name__ == '__main__':
    main()

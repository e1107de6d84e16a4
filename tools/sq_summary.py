#!/usr/bin/env python3
"""Summarise two rocprofv3 --pmc passes (SQ counters; GRBM_GUI_ACTIVE) per conv / Swin-block kernel family: clock and MFMA busy %.

usage: sq_summary.py PASS1.db PASS2.db OUT.txt
"""
import collections
import sqlite3
import sys


def agg(db):
    con = sqlite3.connect(db)
    out = collections.defaultdict(lambda: collections.defaultdict(float))
    n, dur = collections.Counter(), collections.defaultdict(float)
    for k, c, v in con.execute('select kernel_name, counter_name, value from counters_collection'):
        if 'conv_' in k or 'swin_' in k:
            out[k.split('(')[0].replace('void rsa::', '')][c] += v
    for k, d in con.execute('select name, duration from kernels'):
        if 'conv_' in k or 'swin_' in k:
            fam = k.split('(')[0].replace('void rsa::', '')
            n[fam] += 1
            dur[fam] += d
    return out, n, dur


def main():
    o1, n1, d1 = agg(sys.argv[1])
    o2, _, d2 = agg(sys.argv[2])
    lines = ['# rocprofv3 --pmc, two separate passes over `python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline` (3 forwards), FINAL build, MI355X',
             '# per kernel family, summed over its launches.  clock = GRBM_GUI_ACTIVE / 8 XCDs / duration (pass 2);',
             '# MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (duration * clock) (pass 1)']  # fmt: skip
    for fam in sorted(o1):
        c, g = o1[fam], o2[fam].get('GRBM_GUI_ACTIVE', 0.0)
        dur1, dur2 = d1[fam] / 1e3, d2[fam] / 1e3
        clock = g / 8 / (dur2 * 1e-6) / 1e9 if dur2 else 0.0
        mfma = c['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / (dur1 * 1e-6 * clock * 1e9) if clock else 0.0
        lines.append(f'{fam} launches={n1[fam]} total_us={dur1:.0f}: ' + ', '.join(f'{k}={v:.4g}' for k, v in sorted(c.items()))
                     + f', GRBM_GUI_ACTIVE={g:.4g} (pass 2, {dur2:.0f} us) -> clock {clock:.2f} GHz, MFMA busy {100 * mfma:.1f}%')  # fmt: skip
    open(sys.argv[3], 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(lines))


if __name__ == '__main__':
    main()

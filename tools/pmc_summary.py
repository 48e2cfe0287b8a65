#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs: per kernel name and counter -> dispatches, mean, max.

    python tools/pmc_summary.py out.csv <counter_collection.csv> [<counter_collection.csv> ...]

Used for profiles/*_pmc_*.csv (FETCH_SIZE and WRITE_SIZE are collected in separate passes, as MI355X_MICROARCH.md prescribes)."""
import collections
import csv
import sys


def main():
    out, files = sys.argv[1], sys.argv[2:]
    acc = collections.defaultdict(list)
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                acc[(r['Kernel_Name'], r['Counter_Name'])].append(float(r['Counter_Value']))
    rows = sorted(acc.items(), key=lambda kv: -sum(kv[1]))
    with open(out, 'w') as fh:
        fh.write('# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras\n')
        fh.write('# units: KiB per dispatch as reported; gfx950: FETCH_SIZE under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md HBM section)\n')
        w = csv.writer(fh)
        w.writerow(['kernel', 'counter', 'dispatches', 'mean_KiB', 'max_KiB'])
        for (k, c), v in rows[:400]:          # (round 5 kept 60 rows: the grid sweep's 10 MB WRITE_SIZE row fell off the end)
            w.writerow([k[:120], c, len(v), f'{sum(v) / len(v):.1f}', f'{max(v):.1f}'])


if __name__ == '__main__':
    main()

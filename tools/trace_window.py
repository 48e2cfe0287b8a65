#!/usr/bin/env python3
"""Per-iteration kernel summary from a rocprofv3 kernel trace.

    python tools/trace_window.py <kernel_trace.csv> <out.csv> [--anchor lbs_bwd_kernel] [--steps 20]

bench.py's process also runs the pre-fit of the SDF network and the warm-up steps, so the whole-run `kernel_stats.csv` mixes them
with the timed iterations.  This takes the LAST `steps` periods of a once-per-iteration anchor kernel (start-to-start) and reports,
per kernel name, launches and busy microseconds per iteration, plus the period (wall) and the sum of kernel durations (busy).
"""
import argparse
import csv
import collections


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('trace')
    ap.add_argument('out')
    ap.add_argument('--anchor', default='lbs_bwd_kernel')
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--detail', default=None, help='print every launch of the last iteration whose name contains this')
    ap.add_argument('--timeline', default=None, help='write the last iteration as a timeline (offset, duration, idle gap before) here')
    a = ap.parse_args()
    rows = []
    with open(a.trace) as f:
        for r in csv.DictReader(f):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], int(r.get('Grid_Size_X', r.get('Grid_Size', 0)) or 0)))
    rows.sort()
    anchors = [s for s, e, n, g in rows if a.anchor in n]
    if len(anchors) < a.steps + 1:
        raise SystemExit(f'only {len(anchors)} launches of the anchor {a.anchor}')
    t0, t1 = anchors[-a.steps - 1], anchors[-1]
    busy = collections.defaultdict(lambda: [0, 0])
    for s, e, n, g in rows:
        if t0 <= s < t1:
            busy[n][0] += 1
            busy[n][1] += e - s
    tot = sum(v[1] for v in busy.values())
    with open(a.out, 'w') as f:
        w = csv.writer(f)
        w.writerow(['Name', 'LaunchesPerIter', 'BusyUsPerIter', 'AvgUs', 'PercentOfBusy'])
        for n, (c, d) in sorted(busy.items(), key=lambda kv: -kv[1][1]):
            w.writerow([n, f'{c / a.steps:.2f}', f'{d / a.steps / 1e3:.1f}', f'{d / c / 1e3:.1f}', f'{100.0 * d / tot:.2f}'])
        w.writerow(['# period_us_per_iter', f'{(t1 - t0) / a.steps / 1e3:.1f}', 'busy_us_per_iter', f'{tot / a.steps / 1e3:.1f}',
                    f'launches_per_iter {sum(v[0] for v in busy.values()) / a.steps:.0f}'])
    if a.timeline is not None:
        # the last period, in start order: offset from the anchor, duration, and the time the whole GPU sat idle before this launch
        ts = anchors[-2]
        last_end = None
        with open(a.timeline, 'w') as f:
            f.write('offset_us,dur_us,idle_before_us,kernel\n')
            for s, e, n, g in rows:
                if ts <= s < t1:
                    gap = 0.0 if last_end is None else max(0.0, (s - last_end) / 1e3)
                    f.write(f'{(s - ts) / 1e3:.1f},{(e - s) / 1e3:.1f},{gap:.1f},"{n[:70]}"\n')
                    last_end = e if last_end is None else max(last_end, e)
    if a.detail is not None:
        # every launch of the last period whose name contains the pattern: start offset, duration, grid size
        ts = anchors[-2]
        for s, e, n, g in rows:
            if ts <= s < t1 and a.detail in n:
                print(f'{(s - ts) / 1e3:10.1f} us  dur {(e - s) / 1e3:8.1f} us  grid {g:10d}  {n[:90]}')


if __name__ == '__main__':
    main()

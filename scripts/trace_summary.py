#!/usr/bin/env python
"""Steady-state per-step kernel table from a rocprofv3 kernel trace.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python3 bench.py ...
    python scripts/trace_summary.py gpurun_out/trace --steps 4 --out profiles/rNN/....csv

The window is the last ``--steps`` periods of a kernel that runs exactly once per step
(default: gram16_fused_kernel, the B <= 32 loss forward), so warm-up, MIOpen's find step and the
CPU-baseline leg are excluded.  Reports, per kernel: launches per step, mean duration and
milliseconds per step; plus the window's wall time and GPU-busy time (union of kernel
intervals) per step.
"""
import argparse
import csv
import glob
import os
import sys


def find_trace(path):
    if os.path.isfile(path):
        return path
    hits = glob.glob(os.path.join(path, '**', '*kernel_trace.csv'), recursive=True)
    if not hits:
        sys.exit('no *kernel_trace.csv under %s' % path)
    return max(hits, key=os.path.getsize)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('trace')
    ap.add_argument('--steps', type=int, default=4)
    ap.add_argument('--marker', default='gram16_fused_kernel')
    ap.add_argument('--out', default=None)
    ap.add_argument('--note', default='')
    ap.add_argument('--gaps', type=int, default=0,
                    help='also print the N largest classes of idle gaps (GPU idle time before a '
                         'kernel, summed by the kernel that ends the gap), ms per step')
    args = ap.parse_args()
    rows = []
    with open(find_trace(args.trace)) as f:
        for r in csv.DictReader(f):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
    rows.sort()
    marks = [s for s, _, n in rows if args.marker in n]
    if len(marks) < args.steps + 1:
        sys.exit('marker %r seen %d times; need %d' % (args.marker, len(marks), args.steps + 1))
    t0, t1 = marks[-args.steps - 1], marks[-1]
    win = [(s, e, n) for s, e, n in rows if t0 <= s < t1]
    busy, cur_s, cur_e = 0, None, None
    for s, e, _ in win:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        busy += cur_e - cur_s
    acc = {}
    for s, e, n in win:
        c, t = acc.get(n, (0, 0))
        acc[n] = (c + 1, t + e - s)
    k = float(args.steps)
    wall_ms = (t1 - t0) / 1e6 / k
    busy_ms = busy / 1e6 / k
    out = open(args.out, 'w', newline='') if args.out else sys.stdout
    w = csv.writer(out)
    w.writerow(['kernel', 'launches_per_step', 'avg_us', 'ms_per_step'])
    w.writerow(['# window = last %d steps; wall %.3f ms/step; GPU busy %.3f ms/step; '
                'kernel sum %.3f ms/step%s'
                % (args.steps, wall_ms, busy_ms, sum(t for _, t in acc.values()) / 1e6 / k,
                   ('; ' + args.note) if args.note else ''), '', '', ''])
    for n, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        w.writerow([n, round(c / k, 2), round(t / c / 1e3, 2), round(t / 1e6 / k, 4)])
    if args.gaps:
        gaps, last_e = {}, None
        for s_, e_, n in win:
            if last_e is not None and s_ > last_e:
                c, t = gaps.get(n, (0, 0))
                gaps[n] = (c + 1, t + s_ - last_e)
            last_e = e_ if last_e is None else max(last_e, e_)
        print('idle before kernel: count/step, mean us, ms/step')
        for n, (c, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:args.gaps]:
            print('  %6.1f %8.1f %8.4f  %s' % (c / k, t / c / 1e3, t / 1e6 / k, n[:90]))
    if args.out:
        out.close()
        print('wall %.3f ms/step, busy %.3f ms/step -> %s' % (wall_ms, busy_ms, args.out))


if __name__ == '__main__':
    main()

#!/usr/bin/env python
"""Per-kernel means of rocprofv3 --pmc passes (one pass per counter set, never combined with
tracing: MI355X_MICROARCH.md HBM / rocprofv3 section).

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc/fetch -- python3 scripts/microbench.py --what netvlad --iters 3
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc/write -- python3 scripts/microbench.py ...
    python scripts/pmc_summary.py gpurun_out/pmc --out profiles/rNN/pmc_....csv [--json profiles/pmc_traffic.json]

HBM bytes per launch: read = 2 * FETCH_SIZE * 1024 (gfx950 reports half of wide coalesced
reads; the factor is validated on a kernel with a known stream, see profiles/r01/README.md),
write = WRITE_SIZE * 1024.
"""
import argparse
import csv
import glob
import json
import os
import re
import sys


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    return re.sub(r'\(.*$', '', name)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('root')
    ap.add_argument('--out', default=None)
    ap.add_argument('--only', default='kernel', help='substring a kernel name must contain')
    ap.add_argument('--merge-templates', action='store_true',
                    help='drop template arguments: one row per kernel family (mean over all '
                         'its dispatches)')
    args = ap.parse_args()
    acc = {}
    for path in glob.glob(os.path.join(args.root, '**', '*counter_collection.csv'), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                k, c = short(r['Kernel_Name']), r['Counter_Name']
                if args.only not in k:
                    continue
                if args.merge_templates:
                    k = re.sub(r'<.*$', '', k)
                s, n = acc.get((k, c), (0.0, 0))
                acc[(k, c)] = (s + float(r['Counter_Value']), n + 1)
    if not acc:
        sys.exit('no counter rows under %s' % args.root)
    kernels = sorted({k for k, _ in acc})
    counters = sorted({c for _, c in acc})
    out = open(args.out, 'w', newline='') if args.out else sys.stdout
    w = csv.writer(out)
    w.writerow(['kernel', 'dispatches'] + counters + ['hbm_read_MB_corrected', 'hbm_write_MB'])
    for k in kernels:
        vals = {c: acc[(k, c)][0] / acc[(k, c)][1] for c in counters if (k, c) in acc}
        n = max(acc[(k, c)][1] for c in counters if (k, c) in acc)
        rd = 2 * vals['FETCH_SIZE'] * 1024 / 1e6 if 'FETCH_SIZE' in vals else ''
        wr = vals['WRITE_SIZE'] * 1024 / 1e6 if 'WRITE_SIZE' in vals else ''
        w.writerow([k, n] + [round(vals.get(c, float('nan')), 1) for c in counters] +
                   [round(rd, 2) if rd != '' else '', round(wr, 2) if wr != '' else ''])
    if args.out:
        out.close()
        print('wrote', args.out)


if __name__ == '__main__':
    main()

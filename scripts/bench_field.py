"""Print chosen fields of bench.py's JSON line read from stdin: `bench.py ... | python scripts/bench_field.py ms_per_step comm.finish_wait_us_median`."""
import json
import sys

line = [l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]
d = json.loads(line)
out = []
for path in sys.argv[1:]:
    v = d
    for k in path.split('.'):
        v = v.get(k) if isinstance(v, dict) else None
    out.append('%s=%s' % (path, json.dumps(v)))
print(' '.join(out))

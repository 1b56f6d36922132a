#!/usr/bin/env python3
"""One-line digest of bench.py JSON files."""
import json, sys
for path in sys.argv[1:]:
    # the JSON line is the last line that starts with a brace (a gloo rehearsal prints its own lines first)
    d = json.loads([line for line in open(path) if line.startswith('{')][-1])
    r = d.get('roofline') or {}
    c = d['config']
    print('%s: %.0f %s  %.2f us/step  kernel %.2f us frac %s  in_flight %s serial %s enqueue %s' % (
        path, d['value'], d['unit'], 1e3 * d['ms_per_step'], 1e3 * (r.get('kernel_ms') or 0), r.get('frac'),
        c.get('batches_in_flight'), c.get('serial_ms_per_step'), c.get('host_enqueue_ms_per_step')))

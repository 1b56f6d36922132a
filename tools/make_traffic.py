#!/usr/bin/env python3
"""profiles/traffic.json from the PMC summaries tools/profile.sh wrote (tools/pmc_summary.py format).
usage: make_traffic.py <workload>=<pmc.txt>[:<bench arguments, for the record>] ...
HBM bytes per launch of a kernel = FETCH_SIZE [KiB] x 1024 x 2 (on gfx950 FETCH_SIZE tallies 128-byte
requests at 64 bytes) + WRITE_SIZE [KiB] x 1024; TCC_MISS_sum x 128 bytes is kept as the cross-check."""
import json, os, re, sys
KERNELS = {'k_detect_rows': 'yh::k_detect_rows', 'k_stream_wave': 'yh::k_stream_wave', 'k_stream': 'yh::k_stream<',
           'k_loss_main': 'yh::k_loss_main', 'k_loss_grad_pos': 'yh::k_loss_grad_pos', 'k_sort_nms': 'yh::k_sort_nms', 'k_post': 'yh::k_post'}
out_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles', 'traffic.json')
try:
    out = json.load(open(out_path))
    out = {k: v for k, v in out.items() if isinstance(v, dict)}
except (OSError, ValueError):
    out = {}
fresh = {}
for arg in sys.argv[1:]:
    workload, path = arg.split('=', 1)
    path, _, cmd = path.partition(':')
    vals = {}
    for line in open(path):
        if line.startswith('#') or not line.strip():
            continue
        name, rest = line.split(None, 1)
        for key, pat in KERNELS.items():
            if name.startswith(pat):
                d = vals.setdefault(key + '|' + name, {})
                for m in re.finditer(r'(\w+)=([-+.\deE]+)\(n=(\d+)\)', rest):
                    d[m.group(1)] = (float(m.group(2)), int(m.group(3)))
    for kn, d in vals.items():
        key, name = kn.split('|', 1)
        if 'FETCH_SIZE' not in d:
            continue
        fetch, n = d['FETCH_SIZE']
        write = d.get('WRITE_SIZE', (0.0, 0))[0]
        rec = dict(bytes_per_launch=int(fetch * 1024 * 2 + write * 1024), fetch_size_kib=fetch, write_size_kib=write,
                   tcc_miss_x128=int(d['TCC_MISS_sum'][0] * 128) if 'TCC_MISS_sum' in d else None,
                   tcc_hit=d.get('TCC_HIT_sum', (None, 0))[0], dispatches=n, kernel=name,
                   source='profiles/%s (rocprofv3 --pmc FETCH_SIZE TCC_MISS_sum / WRITE_SIZE TCC_HIT_sum, separate passes, '
                          'mean over %d dispatches; bench.py %s)' % (os.path.basename(path), n, cmd))
        # one workload may hold several template instances of a kernel: keep the one with most dispatches
        if fresh.get((workload, key), -1) < n:
            fresh[(workload, key)] = n
            out.setdefault(workload, {})[key] = rec
json.dump(out, open(out_path, 'w'), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))

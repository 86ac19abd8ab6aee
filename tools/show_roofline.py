#!/usr/bin/env python3
"""Print the per-entry-point table of a tools/config_bench.py --roofline json.  usage: show_roofline.py file [cfg]"""
import json
import sys
d = json.load(open(sys.argv[1]))
for k, v in d.items():
    if len(sys.argv) > 2 and k not in sys.argv[2:]:
        continue
    print(k, {a: b for a, b in v.items() if a not in ('top_kernels', 'abi_calls')})
    for r in v.get('abi_calls', []):
        print('  %-24s %-38s x%d %7.1fus %6.0fGB/s %6.1fTF hbm %.2f mfma %.2f' % (
            r['entry'][5:], r['shape'], r['calls'], r['us'], r['GB_per_s'], r['TFLOP_per_s'], r['frac_hbm'], r['frac_mfma']))

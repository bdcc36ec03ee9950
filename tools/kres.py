#!/usr/bin/env python3
"""Summarise hipcc -Rpass-analysis=kernel-resource-usage remarks: tools/kres.py build.log"""
import re, sys
for path in sys.argv[1:]:
    txt = open(path).read()
    for b in re.split(r'remark: [^\n]*Function Name: ', txt)[1:]:
        name = b.split('\n')[0].strip()
        def g(k):
            m = re.search(k + r': (\d+)', b)
            return m.group(1) if m else '?'
        short = re.sub(r'^_ZN3dmm', '', name)[:78]
        print("%-80s vgpr=%4s agpr=%3s sgpr=%3s scratch=%4s occ=%2s lds=%s" % (
            short, g('VGPRs'), g('AGPRs'), g('SGPRs'), g(r'ScratchSize \[bytes/lane\]'),
            g(r'Occupancy \[waves/SIMD\]'), g(r'LDS Size \[bytes/block\]')))

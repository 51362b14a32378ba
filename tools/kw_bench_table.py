#!/usr/bin/env python3
"""compact table of a tools/kw_bench log: one line per shape -- the round-4 schedule, GM_TILE where planned, GM_KW per tile height (us per launch)"""
import re
import sys
cur = None
def flush():
    if cur:
        print('%-14s ref %7.2f | ' % (cur[0], cur[1]) + '  '.join(cur[2]))
for l in open(sys.argv[1]):
    m = re.match(r'(\S+\s+\S+( L)?)\s+M=.*mode (\d)\):\s+([\d.]+) us', l)
    if m:
        flush(); cur = [m.group(1), float(m.group(4)), []]; continue
    m = re.match(r'\s+GM_KW mt=(\d)( \(planner\))? :\s+([\d.]+) us.*?(bit-identical|MISMATCH)', l)
    if m and cur: cur[2].append('kw%s %7.2f%s' % ('*' if m.group(2) else m.group(1), float(m.group(3)), '' if m.group(4) == 'bit-identical' else '!!'))
    m = re.match(r'\s+GM_TILE.*:\s+([\d.]+) us.*?(bit-identical|MISMATCH)', l)
    if m and cur: cur[2].append('TILE %7.2f%s' % (float(m.group(1)), '' if m.group(2) == 'bit-identical' else '!!'))
flush()

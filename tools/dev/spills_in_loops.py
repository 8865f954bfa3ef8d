#!/usr/bin/env python3
"""Developer check on a hipcc -S listing: scratch (spill) instructions per kernel, split by the loop depth of their basic block."""
import re, sys, collections
name, depth = None, 0
cnt = collections.defaultdict(lambda: collections.Counter())
for line in open(sys.argv[1]):
    m = re.match(r'^(_Z\S+):\s', line)
    if m:
        name, depth = m.group(1), 0
        continue
    if line.startswith('; %bb.') or re.match(r'^\.LBB\d+_\d+:', line):
        d = re.search(r'Depth=(\d+)', line)
        depth = int(d.group(1)) if d else 0
    elif 'Loop Header: Depth=' in line:
        depth = int(re.search(r'Depth=(\d+)', line).group(1))
    if name and 'scratch_' in line:
        cnt[name][depth] += 1
for n, c in cnt.items():
    print(dict(c), n[:100])

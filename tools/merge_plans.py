#!/usr/bin/env python3
"""Merge re-tuned plan lines over the shipped table: usage merge_plans.py <shipped> <new...> > merged  (a key = the first 10 ints;
the LAST line of a key in the new files wins, keys missing from them keep their shipped line)."""
import sys
head, plans = [], {}
for i, fn in enumerate(sys.argv[1:]):
    for line in open(fn):
        if line.startswith("#"):
            if i == 0: head.append(line)
            continue
        f = line.split()
        if len(f) < 13: continue
        plans[tuple(int(x) for x in f[:10])] = line
sys.stdout.write("".join(head))
for k in sorted(plans):
    sys.stdout.write(plans[k])

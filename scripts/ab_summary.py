"""Summarise a scripts/ab_run.sh log: TB/s per config and variant."""
import re, sys, collections
d = collections.OrderedDict()
for ln in open(sys.argv[1]):
    m = re.match(r'(\S+)\s+(\d+) x\s+(\d+)\s+(\d+)-bit m(\d) (\S+):\s+([\d.]+) us/sweep\s+([\d.]+) TB/s', ln)
    if m:
        d.setdefault((m.group(2), m.group(3), m.group(4), m.group(5), m.group(6)), collections.OrderedDict()).setdefault(m.group(1), []).append(float(m.group(8)))
for k, v in d.items():
    print("%s x %s, %s-bit, metric %s, %s" % k)
    for n, t in v.items():
        print('   %-10s %s' % (n, ' '.join('%.2f' % x for x in t)))

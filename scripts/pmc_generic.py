"""Per-kernel averages of arbitrary rocprofv3 --pmc counters: python scripts/pmc_generic.py DIR [DIR ...]"""
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for d in sys.argv[1:]:
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for row in csv.DictReader(open(f)):
            name = row['Kernel_Name'].split('(')[0].replace('void ', '').replace('lzmi::', '')
            v = acc[name][row['Counter_Name']]
            v[0] += float(row['Counter_Value']); v[1] += 1
for k in sorted(acc):
    if not k.startswith(('enc_', 'dec_')):
        continue
    print(k, ' '.join(f"{c}={v[0]/v[1]:.4g}" for c, v in sorted(acc[k].items())))

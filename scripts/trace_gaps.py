"""Reads a rocprofv3 --kernel-trace csv and prints, for the last encode and decode bursts, wall span vs summed kernel time and the gaps."""
import csv, sys, glob, os
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ', '').replace('lzmi::', '')) for r in csv.DictReader(open(f))]
rows.sort()
# split into bursts separated by > 300 us of idle
bursts, cur = [], [rows[0]]
end = rows[0][1]
for r in rows[1:]:
    if r[0] - end > 300000:
        bursts.append(cur); cur = []
    cur.append(r); end = max(end, r[1])
bursts.append(cur)
for b in bursts[-int(os.environ.get('BURSTS', '4')):]:   # BURSTS=n: the last n bursts
    t0 = b[0][0]; t1 = max(r[1] for r in b)
    busy = 0; e = t0
    for s_, e_, _ in b:
        if e_ > e:
            busy += e_ - max(s_, e); e = e_
    kinds = 'enc' if b[0][2].startswith('enc') else 'dec'
    print(f"{kinds} burst: {len(b)} kernels, span {(t1-t0)/1e6:.3f} ms, device busy {busy/1e6:.3f} ms, first {b[0][2]}, sum of kernel durations {sum(r[1]-r[0] for r in b)/1e6:.3f} ms")
    if len(sys.argv) > 2:
        for s_, e_, n in b:
            print(f"   {(s_-t0)/1e6:8.3f} -> {(e_-t0)/1e6:8.3f}  {n}")

"""Aggregates rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE csv output into per-kernel HBM bytes per launch.
FETCH_SIZE / WRITE_SIZE are in KiB (MI355X_MICROARCH.md, HBM section); on gfx950 FETCH_SIZE reads one half
of the bytes of wide (16 B/lane) coalesced streaming reads - both the raw and the x2-corrected read
figure are kept, the corrected one only applies to streaming kernels."""
import csv, glob, json, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def collect(d, counter):
    out = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get('Counter_Name') != counter:
                continue
            name = row['Kernel_Name'].split('(')[0].replace('void ', '').replace('lzmi::', '')
            v = out[name]
            v[0] += float(row['Counter_Value']); v[1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in out.items()}

fetch = collect(sys.argv[1], 'FETCH_SIZE')
write = collect(sys.argv[2], 'WRITE_SIZE')
res = {}
for k in sorted(set(fetch) | set(write)):
    f = fetch.get(k, (0, 0)); w = write.get(k, (0, 0))
    res[k] = {"launches": max(f[1], w[1]), "fetch_bytes_raw": f[0] * 1024, "fetch_bytes_x2": f[0] * 2048,
              "write_bytes": w[0] * 1024, "hbm_bytes_raw": (f[0] + w[0]) * 1024, "hbm_bytes": (2 * f[0] + w[0]) * 1024}
import bench
# stamped with the identity of the kernel sources it was taken on: bench.py reports roofline.traffic only for these
calls = int(sys.argv[4]) if len(sys.argv) > 4 else 0   # encode (= decode) batch calls of the profiled command: verify + warm-up + steps
json.dump({"source_sha": bench.kernel_source_sha(), "command": "python bench.py --steps 2 --warmup 1 (default workload: snappy x 256)",
           "encode_calls": calls,
           "units": "bytes per launch, averaged over the launches of the run; FETCH_SIZE / WRITE_SIZE KiB -> bytes; hbm_bytes = "
                    "2 x fetch + write (gfx950: FETCH_SIZE tallies 128-byte read requests at 64 bytes, MI355X_MICROARCH.md HBM section)",
           "kernels": res}, open(sys.argv[3], 'w'), indent=1)
for k, v in res.items():
    if k.startswith(('enc_', 'dec_')):
        print(f"{k:28s} launches {v['launches']:4d} fetch(raw) {v['fetch_bytes_raw']/1e6:9.1f} MB write {v['write_bytes']/1e6:9.1f} MB  hbm = 2 x fetch + write {v['hbm_bytes']/1e6:9.1f} MB")

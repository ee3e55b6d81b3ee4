"""Summary of scripts/stall_probe.py ... trace: timed samples (index >= 0) whose encode call took more than 1.6 x the file's median,
with the host-phase lines of the lanes of that call."""
import sys, re, statistics
cur, blocks = None, []
for l in open(sys.argv[1]):
    l = l.strip()
    if l.startswith("sample "):
        _, name, k = l.split(); cur = {"name": name, "k": int(k), "lanes": [], "wall": None}; blocks.append(cur)
    elif l.startswith("enc_host") and cur is not None:
        cur["lanes"].append(l)
    elif l.startswith("wall ") and cur is not None:
        cur["wall"] = float(l.split()[4])
by = {}
for b in blocks:
    if b["k"] >= 0 and b["wall"] is not None: by.setdefault(b["name"], []).append(b)
n_slow = 0
for name, bs in by.items():
    med = statistics.median(b["wall"] for b in bs)
    for b in bs:
        if b["wall"] > 1.6 * med:
            n_slow += 1
            print(f"{name} sample {b['k']}: wall {b['wall']:.3f} ms (median {med:.3f})")
            for l in b["lanes"]: print("   ", l)
print(f"{sum(len(v) for v in by.values())} timed samples, {n_slow} slow")

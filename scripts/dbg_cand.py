import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
import numpy as np
from oracle_py import Oracle
import lzfse_rust_amd as m
from test_gpu_encode import gpu_candidates, synth_cases
o = Oracle(); ctx = m.Context(0)
for name, raw in synth_cases().items():
    mi, fl = o.candidates(raw)
    prev, rec = gpu_candidates(ctx, raw)
    a = np.frombuffer(raw, dtype=np.uint8)
    n = len(raw) - 3
    v = (a[:n].astype(np.uint32) | (a[1:n+1].astype(np.uint32) << 8) | (a[2:n+2].astype(np.uint32) << 16) | (a[3:n+3].astype(np.uint32) << 24))
    key = ((v.astype(np.uint64) * 0x9E3779B1) & 0xFFFFFFFF) >> 18
    order = np.lexsort((np.arange(n), key))
    tp = np.full(n, 0xFFFFFFFF, dtype=np.uint32)
    same = key[order][1:] == key[order][:-1]
    tp[order[1:][same]] = order[:-1][same]
    badp = np.nonzero(tp != prev)[0]
    fwd = rec[:, 1]; capped = (rec[:, 0] >> 31) != 0
    bad = np.nonzero((fwd != fl) & ~capped)[0]
    print(name, 'n', n, 'prev mismatches', len(badp), 'fwd mismatches', len(bad))
    for i in badp[:5]:
        print('  prev @', i, 'true', tp[i], 'gpu', prev[i], 'key', key[i])
    for i in bad[:5]:
        print('  fwd @', i, 'oracle', mi[i], fl[i], 'gpu dist', rec[i,0] & 0x3FFFF, 'fwd', fwd[i], 'prev', prev[i])

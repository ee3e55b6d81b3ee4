import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
bad=[(r["file"], r["gpu_encode_ms_median_max"], r["gpu_decode_ms_median_max"]) for r in d["rows"] if r["gpu_encode_ms_median_max"][1] > 1.6*r["gpu_encode_ms_median_max"][0] or r["gpu_decode_ms_median_max"][1] > 1.6*r["gpu_decode_ms_median_max"][0]]
print(sys.argv[1], "slow rows:", bad)

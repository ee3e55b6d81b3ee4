import json,sys
a=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); b=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
for x,y in zip(a["rows"],b["rows"]):
    print(x["file"][:18].ljust(18), "raw MB x R", round(x["raw_bytes"]*x["copies"]/1e6), "enc 1 lane", x["gpu_encode_GBps"], "2 lanes", y["gpu_encode_GBps"])

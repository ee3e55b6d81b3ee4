for v in 1 4 5; do
  LZFSE_MI_LZ_VARIANT=$v timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant $v', d['value'], d['encode_MBps'], d['decode_MBps'], d['kernel_ms_per_step']['dec_lz'], d['kernel_ms_per_step']['dec_fse'])" || exit 1
done

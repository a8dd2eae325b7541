#!/bin/bash
# one GPU call: HC parity tests, then stamps + env-knob matrix on cfg4 (diagnostic)
mkdir -p gpurun_out/$1
o=gpurun_out/$1/matrix.txt
for c in 1 4; do
  ZLZ4_HC_CANDS=$c python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -x -q -m gpu -k "hc" > gpurun_out/$1/t_hc_$c.log 2>&1 || { tail -30 gpurun_out/$1/t_hc_$c.log; exit 1; }
  tail -1 gpurun_out/$1/t_hc_$c.log
done
for c in 1 2 4; do echo "== stamps cands=$c" >> $o; ZLZ4_HC_CANDS=$c ZLZ4_AMD_LIB=zig-lz4_amd/libzlz4_amd_stamps.so python tools/hc_seg_stamps.py 9 text 2>&1 | grep -v amdgpu.ids >> $o; done
for cfg in "1 32 2" "2 32 2" "4 32 2" "4 16 4" "4 64 1" "4 32 1" "4 16 2"; do set -- $cfg
  echo "== cands=$1 seg=$2 lps=$3" >> $o
  ZLZ4_HC_CANDS=$1 ZLZ4_HC_SEG=$2 ZLZ4_HC_LPS=$3 python bench.py --workload cfg4 --steps 3 --warmup 1 --no-cpu 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('compress_ms', d['compress_ms'])" >> $o 2>&1
done
cat $o

#!/bin/bash
# one GPU call: HC parity tests, then stamps + env-knob matrix on cfg4 (diagnostic)
mkdir -p gpurun_out/$1
o=gpurun_out/$1/matrix.txt
python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -x -q -m gpu -k "hc or golden or kats or understated" > gpurun_out/$1/t_hc.log 2>&1 || { tail -30 gpurun_out/$1/t_hc.log; exit 1; }
tail -1 gpurun_out/$1/t_hc.log
for c in 2 4; do echo "== stamps cands=$c" >> $o; ZLZ4_HC_CANDS=$c ZLZ4_AMD_LIB=zig-lz4_amd/libzlz4_amd_stamps.so python tools/hc_seg_stamps.py 9 text 2>&1 | grep -v amdgpu.ids >> $o; done
for cfg in "4 32 2" "2 32 2" "4 64 1"; do set -- $cfg
  echo "== cands=$1 seg=$2 lps=$3" >> $o
  ZLZ4_AMD_LIB=zig-lz4_amd/libzlz4_amd_tuning.so ZLZ4_HC_CANDS=$1 ZLZ4_HC_SEG=$2 ZLZ4_HC_LPS=$3 python bench.py --workload cfg4 --steps 3 --warmup 1 --no-cpu 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('compress_ms', d['compress_ms'])" >> $o 2>&1
done
cat $o
tools/prof_kernels.sh $1 --workload cfg4 --steps 3 --warmup 1 --no-cpu | grep -v "^{"

#!/bin/bash
# HC parity tests + cfg4 kernel profile in one GPU call: tools/gpu_hc_check.sh <tag> [extra bench args]
tag=$1; shift
mkdir -p gpurun_out/$tag
python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -x -q -m gpu -k "hc" > gpurun_out/$tag/t_hc.log 2>&1
rc=$?
tail -4 gpurun_out/$tag/t_hc.log
[ $rc -ne 0 ] && exit $rc
tools/prof_kernels.sh $tag --workload cfg4 --steps 3 --warmup 1 --no-cpu "$@"

set -e
mkdir -p gpurun_out/r3l
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_shipped_paths.py tests/test_gpu_frame.py tests/test_gpu_large.py tests/test_gpu_sharded_frame.py -m gpu -x -q -k "hc or golden or optimal or max_in_len or graph or frame or verify" > gpurun_out/r3l/parity.txt 2>&1 || { tail -30 gpurun_out/r3l/parity.txt; exit 1; }
tail -3 gpurun_out/r3l/parity.txt
export ZLZ4_AMD_LIB=$PWD/zig-lz4_amd/libzlz4_amd_tuning.so
for rep in 1 2; do
python bench.py --workload cfg4 --steps 5 --warmup 2 --no-cpu 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('beside  L9 comp %.2f ms %.2f GiB/s'%(d['compress_ms'],d['compress_gibs_per_gpu']))"
ZLZ4_HC_LINKS_INLINE=1 python bench.py --workload cfg4 --steps 5 --warmup 2 --no-cpu 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('inline  L9 comp %.2f ms %.2f GiB/s'%(d['compress_ms'],d['compress_gibs_per_gpu']))"
done
python bench.py --workload cfg4 --level 5 --steps 5 --warmup 2 --no-cpu 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('beside  L5 comp %.2f ms %.2f GiB/s'%(d['compress_ms'],d['compress_gibs_per_gpu']))"
ZLZ4_HC_LINKS_INLINE=1 python bench.py --workload cfg4 --level 5 --steps 5 --warmup 2 --no-cpu 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('inline  L5 comp %.2f ms %.2f GiB/s'%(d['compress_ms'],d['compress_gibs_per_gpu']))"

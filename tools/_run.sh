set -e
mkdir -p gpurun_out/r3l
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_shipped_paths.py tests/test_gpu_frame.py tests/test_gpu_large.py -m gpu -x -q -k "hc or golden or optimal or max_in_len or graph or frame" > gpurun_out/r3l/parity.txt 2>&1 || { tail -30 gpurun_out/r3l/parity.txt; exit 1; }
tail -3 gpurun_out/r3l/parity.txt
for d in text reptext zero mixed; do
python bench.py --workload cfg4 --dist $d --steps 5 --warmup 2 --no-cpu 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$d L9 comp %.2f ms %.2f GiB/s'%(d['compress_ms'],d['compress_gibs_per_gpu']))"
done
python bench.py --workload cfg4 --level 6 --steps 3 --warmup 1 --no-cpu 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('text L6 comp %.2f ms %.2f GiB/s'%(d['compress_ms'],d['compress_gibs_per_gpu']))"

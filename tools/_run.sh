set -e
mkdir -p gpurun_out/r3g
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_shipped_paths.py -m gpu -x -q -k "hc or golden or optimal or max_in_len" > gpurun_out/r3g/parity.txt 2>&1 || { tail -30 gpurun_out/r3g/parity.txt; exit 1; }
tail -3 gpurun_out/r3g/parity.txt
for l in 12 10; do
  python bench.py --workload cfg4 --level $l --steps 2 --warmup 1 --no-cpu > gpurun_out/r3g/l$l.json 2> gpurun_out/r3g/l$l.err || { tail gpurun_out/r3g/l$l.err; exit 1; }
  python -c "import json;d=json.load(open('gpurun_out/r3g/l$l.json'));print($l, d['compress_ms'], d['compress_gibs_per_gpu'])"
done
tools/prof_kernels.sh r3g/prof_l12 --workload cfg4 --level 12 --steps 2 --warmup 1 --no-cpu | grep -v "^{"

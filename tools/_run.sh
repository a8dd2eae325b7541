set -e
mkdir -p gpurun_out/r3c
python -m pytest tests/test_gpu_shipped_paths.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r3c/tests.txt 2>&1 || { tail -40 gpurun_out/r3c/tests.txt; exit 1; }
tail -3 gpurun_out/r3c/tests.txt
ZLZ4_BENCH_REHEARSE=1 python3 bench.py --gpus 2 --steps 2 --warmup 1 --blocks 8192 > gpurun_out/r3c/rehearse2.json 2> gpurun_out/r3c/rehearse2.err || { tail -20 gpurun_out/r3c/rehearse2.err; exit 1; }
cat gpurun_out/r3c/rehearse2.json | cut -c1-400
ZLZ4_BENCH_REHEARSE=1 python3 bench.py --gpus 2 --workload cfg5 --steps 2 --warmup 1 --blocks 128 > gpurun_out/r3c/rehearse2_cfg5.json 2> gpurun_out/r3c/rehearse2_cfg5.err || { tail -20 gpurun_out/r3c/rehearse2_cfg5.err; exit 1; }
cat gpurun_out/r3c/rehearse2_cfg5.json | cut -c1-400

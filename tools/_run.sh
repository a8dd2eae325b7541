set -e
mkdir -p gpurun_out/r3n
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_fuzz.py tests/test_gpu_frame.py tests/test_gpu_large.py -m gpu -x -q -k "default or fast or golden or batch_of or too_small or max_in_len or single_buffer or fuzz or frame" > gpurun_out/r3n/parity.txt 2>&1 || { tail -30 gpurun_out/r3n/parity.txt; exit 1; }
tail -3 gpurun_out/r3n/parity.txt
export ZLZ4_AMD_LIB=$PWD/zig-lz4_amd/libzlz4_amd_tuning.so
for d in text reptext mixed; do python tools/time_compress.py $d 65536 2>/dev/null | sed "s/^.*tuning.so//"; done
python tools/time_compress.py text 1024 4194304 2>/dev/null | sed "s/^.*tuning.so//"

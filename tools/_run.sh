set -e
mkdir -p gpurun_out/r3h
timeout -k 10 1100 python -m pytest tests -m gpu -x -q -k "decompress or decoder or batch_of or single_buffer or interop or fuzz or frame or golden or verify or large" > gpurun_out/r3h/parity.txt 2>&1 || { tail -30 gpurun_out/r3h/parity.txt; exit 1; }
tail -3 gpurun_out/r3h/parity.txt
export ZLZ4_AMD_LIB=$PWD/zig-lz4_amd/libzlz4_amd_tuning.so
rm -f gpurun_out/r3h/ab.txt
for rep in 1 2; do for ph in 1 0; do for d in text reptext; do
  ZLZ4_DECOMP_PHASES=$ph python tools/time_decompress.py $d 65536 2>/dev/null | sed "s/^/phases=$ph /" >> gpurun_out/r3h/ab.txt
done; done; done
for d in mixed zero ramp random; do python tools/time_decompress.py $d 65536 2>/dev/null >> gpurun_out/r3h/ab.txt; done
cat gpurun_out/r3h/ab.txt

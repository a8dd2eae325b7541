set -e
mkdir -p gpurun_out/r3k
rm -f gpurun_out/r3k/ab.txt
for rep in 1 2; do
  ZLZ4_AMD_LIB=$PWD/zig-lz4_amd/libzlz4_amd_r02.so python tools/time_decompress.py text 65536 2>/dev/null | sed "s/^/r02 /" >> gpurun_out/r3k/ab.txt
  export ZLZ4_AMD_LIB=$PWD/zig-lz4_amd/libzlz4_amd_tuning.so
  python tools/time_decompress.py text 65536 2>/dev/null | sed "s/^/cur /" >> gpurun_out/r3k/ab.txt
  ZLZ4_DECOMP_PHASE_MIN=64 python tools/time_decompress.py text 65536 2>/dev/null | sed "s/^/cur-min64 /" >> gpurun_out/r3k/ab.txt
  ZLZ4_DECOMP_PHASES=0 python tools/time_decompress.py text 65536 2>/dev/null | sed "s/^/cur-nophase /" >> gpurun_out/r3k/ab.txt
done
cat gpurun_out/r3k/ab.txt

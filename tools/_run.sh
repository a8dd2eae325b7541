set -e
mkdir -p gpurun_out/r3j
rm -f gpurun_out/r3j/ab.txt
for rep in 1 2; do
for v in tuning_f0 tuning tuning_f512 tuning_f192; do
  export ZLZ4_AMD_LIB=$PWD/zig-lz4_amd/libzlz4_amd_$v.so
  python tools/time_compress.py text 65536 2>/dev/null | sed "s/^.*tuning/tuning/" >> gpurun_out/r3j/ab.txt
  python tools/time_compress.py text 1024 4194304 2>/dev/null | sed "s/^.*tuning/tuning/" >> gpurun_out/r3j/ab.txt
done; done
cat gpurun_out/r3j/ab.txt

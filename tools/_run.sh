set -e
mkdir -p gpurun_out/r3e
export ZLZ4_AMD_LIB=$PWD/zig-lz4_amd/libzlz4_amd_stamps.so
echo "== grid 4MiB" > gpurun_out/r3e/stamps2.txt
ZLZ4_TUNE_GRID=1 python tools/stamp_profile.py text 1024 4194304 >> gpurun_out/r3e/stamps2.txt 2>&1
echo "== anchored 4MiB" >> gpurun_out/r3e/stamps2.txt
ZLZ4_TUNE_GRID=0 python tools/stamp_profile.py text 1024 4194304 >> gpurun_out/r3e/stamps2.txt 2>&1
echo "== grid 64K" >> gpurun_out/r3e/stamps2.txt
ZLZ4_TUNE_GRID=1 python tools/stamp_profile.py text 65536 >> gpurun_out/r3e/stamps2.txt 2>&1
grep -v amdgpu.ids gpurun_out/r3e/stamps2.txt

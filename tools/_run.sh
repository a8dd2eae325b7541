set -e
mkdir -p gpurun_out/r3m
export ZLZ4_AMD_LIB=$PWD/zig-lz4_amd/libzlz4_amd_stamps.so
python tools/hc_seg_stamps.py 2>&1 | grep -v amdgpu > gpurun_out/r3m/hcstamps.txt
cat gpurun_out/r3m/hcstamps.txt

set -e
mkdir -p gpurun_out/r3r
ZLZ4_AMD_LIB=$PWD/zig-lz4_amd/libzlz4_amd_stamps.so python tools/stamp_profile.py text 65536 2>&1 | grep -v amdgpu > gpurun_out/r3r/st.txt
echo "=== no flush stores (timing only, output wrong)" >> gpurun_out/r3r/st.txt
ZLZ4_AMD_LIB=$PWD/zig-lz4_amd/libzlz4_amd_stamps_ns.so python tools/stamp_profile.py text 65536 2>&1 | grep -v amdgpu >> gpurun_out/r3r/st.txt
cat gpurun_out/r3r/st.txt

set -e
mkdir -p gpurun_out/r3b
export ZLZ4_AMD_LIB=$PWD/zig-lz4_amd/libzlz4_amd_stamps.so
for tag in 0 1; do
  echo "== tag $tag" >> gpurun_out/r3b/stamps.txt
  ZLZ4_TUNE_TAG=$tag python tools/stamp_profile.py text 65536 >> gpurun_out/r3b/stamps.txt 2>&1
done
echo "== tag 0 pad (13 waves)" >> gpurun_out/r3b/stamps.txt
ZLZ4_TUNE_TAG=0 ZLZ4_TUNE_LDS_PAD=4096 python tools/stamp_profile.py text 65536 >> gpurun_out/r3b/stamps.txt 2>&1
cat gpurun_out/r3b/stamps.txt

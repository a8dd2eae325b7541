set -e
mkdir -p gpurun_out/r3q
export ZLZ4_AMD_LIB=$PWD/zig-lz4_amd/libzlz4_amd_tuning.so ZLZ4_DECOMP_SHORT=32
timeout -k 10 1100 python tools/fuzz_parity.py --check tests/_fuzz_cache/r3a.npz > gpurun_out/r3q/fuzz_a_phases.txt 2>&1 || { tail -20 gpurun_out/r3q/fuzz_a_phases.txt; exit 1; }
tail -2 gpurun_out/r3q/fuzz_a_phases.txt

set -e
mkdir -p gpurun_out/r3q
timeout -k 10 1100 python tools/fuzz_parity.py --check tests/_fuzz_cache/r3b.npz > gpurun_out/r3q/fuzz_b.txt 2>&1 || { tail -20 gpurun_out/r3q/fuzz_b.txt; exit 1; }
tail -3 gpurun_out/r3q/fuzz_b.txt

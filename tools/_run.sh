set -e
tools/prof_kernels.sh r3l/prof_l9 --workload cfg4 --steps 3 --warmup 1 --no-cpu | grep -v "^{"

#!/bin/bash
# Run on the GPU box from the repo root: regenerates everything profiles/ holds for one round.
#   tools/refresh_profiles.sh <tag>      -> gpurun_out/<tag>/...   (copy into profiles/ afterwards, see tools/collect_profiles.py)
# Per BASELINE.json config: the bench line (with cpu_baseline where it is cheap), rocprofv3 --kernel-trace --stats of the
# same command, and FETCH_SIZE / WRITE_SIZE PMC passes (separate runs, kernel-trace only) for roofline.traffic.
#   tools/refresh_profiles.sh <tag> bench|prof|tests   -> one part only (a whole refresh does not fit one 20-minute call;
#   WL="cfg2 cfg3" LV="" tools/refresh_profiles.sh <tag> prof   -> a subset of the profiled workloads / HC levels)
tag=$1
part=${2:-all}
out=gpurun_out/$tag
mkdir -p $out
if [ "$part" = all ] || [ "$part" = bench ]; then
python bench.py --steps 20 --warmup 3 > $out/cfg2_bench.json 2> $out/cfg2_bench.err || tail -5 $out/cfg2_bench.err
python bench.py --workload cfg3 --steps 5 --warmup 1 > $out/cfg3_bench.json 2> $out/cfg3_bench.err || tail -5 $out/cfg3_bench.err
python bench.py --workload cfg4 --steps 10 --warmup 2 > $out/cfg4_bench.json 2> $out/cfg4_bench.err || tail -5 $out/cfg4_bench.err
python bench.py --workload cfg5 --steps 5 --warmup 2 > $out/cfg5_bench.json 2> $out/cfg5_bench.err || tail -5 $out/cfg5_bench.err
for l in 2 12; do python bench.py --workload cfg4 --level $l --steps 2 --warmup 1 --no-cpu > $out/cfg4_level${l}_bench.json 2> $out/cfg4_level${l}_bench.err; done
for d in reptext mixed random zero ramp; do python bench.py --dist $d --steps 3 --warmup 1 --no-cpu > $out/cfg2_${d}_bench.json 2>/dev/null; done
for d in zero mixed reptext; do python bench.py --workload cfg4 --dist $d --steps 3 --warmup 1 --no-cpu > $out/cfg4_${d}_bench.json 2>/dev/null; done
echo "bench lines done"
fi
if [ "$part" = all ] || [ "$part" = prof ]; then
for w in ${WL:-cfg2 cfg3 cfg4 cfg5}; do
  tools/prof_kernels.sh $tag/prof_$w --workload $w --steps 3 --warmup 1 --no-cpu | grep -v "^{"
  tools/pmc_run.sh $out/pmc_$w "fetch write" --workload $w
  python3 tools/pmc_summarize.py $out/pmc_$w > $out/pmc_$w/summary.txt
  echo "$w profiled"
done
for l in ${LV:-2 12}; do     # the other two HC strategies (own kernels, own traffic)
  tools/prof_kernels.sh $tag/prof_cfg4_level$l --workload cfg4 --level $l --steps 2 --warmup 1 --no-cpu | grep -v "^{"
  tools/pmc_run.sh $out/pmc_cfg4_level$l "fetch write" --workload cfg4 --level $l
  python3 tools/pmc_summarize.py $out/pmc_cfg4_level$l > $out/pmc_cfg4_level$l/summary.txt
  echo "cfg4 level $l profiled"
done
fi
if [ "$part" = all ] || [ "$part" = tests ]; then
python -m pytest tests -q -m gpu > $out/gpu_tests.txt 2>&1; tail -2 $out/gpu_tests.txt
fi

#!/bin/bash
# usage: tools/prof_kernels.sh <tag> <bench.py args...>   (run on the GPU box, from the repo root)
# rocprofv3 kernel-trace + stats of one bench.py command; prints the zlz4 kernels' average durations and leaves the
# stats CSV under gpurun_out/<tag>/ (copy the summary into profiles/ to have it judged).
set -e
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
cd "$root"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof" -o run -- python3 bench.py "$@" > "$out/bench.json" 2> "$out/bench.err" || { tail -20 "$out/bench.err"; exit 1; }
python3 - "$out" <<'PY'
import csv, glob, sys
out = sys.argv[1]
f = glob.glob(out + "/prof/**/*kernel_stats.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "zlz4" in r["Name"]]
with open(out + "/kernel_stats_zlz4.csv", "w") as w:
    w.write("kernel,calls,avg_ms,total_ms,min_ms,max_ms\n")
    for r in rows:
        name = r["Name"].split("(")[0].replace("void ", "")
        line = "%s,%s,%.4f,%.3f,%.4f,%.4f" % (name, r["Calls"], float(r["AverageNs"]) / 1e6, float(r["TotalDurationNs"]) / 1e6,
                                            float(r["MinNs"]) / 1e6, float(r["MaxNs"]) / 1e6)
        w.write(line + "\n")
        print(line)
PY
cut -c1-400 "$out/bench.json"

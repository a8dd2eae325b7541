#!/usr/bin/env python3
"""Copy one round's GPU-box outputs (tools/refresh_profiles.sh <tag>) from gpurun_out/<tag>/ into profiles/ and rebuild
profiles/traffic.json (FETCH_SIZE + WRITE_SIZE per launch of the dominant kernels, from the PMC passes).
usage: tools/collect_profiles.py <tag> <round-prefix, e.g. r02>"""
import glob, json, os, re, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, pre = sys.argv[1], sys.argv[2]
src, dst = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles")
for f in glob.glob(os.path.join(src, "*_bench.json")):
    if os.path.getsize(f):
        shutil.copy(f, os.path.join(dst, "%s_%s" % (pre, os.path.basename(f))))
for w in ("cfg2", "cfg3", "cfg4", "cfg5"):
    k = os.path.join(src, "prof_" + w, "kernel_stats_zlz4.csv")
    if os.path.exists(k):
        shutil.copy(k, os.path.join(dst, "%s_%s_text_kernel_stats.csv" % (pre, w)))
    p = os.path.join(src, "pmc_" + w, "summary.txt")
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, "%s_%s_text_pmc_traffic.txt" % (pre, w)))
if os.path.exists(os.path.join(src, "gpu_tests.txt")):
    shutil.copy(os.path.join(src, "gpu_tests.txt"), os.path.join(dst, pre + "_gpu_tests.txt"))


def pmc(w):
    """kernel -> {counter: mean per launch, in bytes (rocprofv3 reports KiB)}"""
    out, cur = {}, None
    p = os.path.join(src, "pmc_" + w, "summary.txt")
    if not os.path.exists(p):
        return out
    for ln in open(p):
        if not ln.startswith(" "):
            cur = ln.strip(); out[cur] = {}
        else:
            m = re.match(r"\s+(\w+)\s+n=(\d+)\s+mean\s+([0-9.e+]+)", ln)
            if m:
                out[cur][m.group(1)] = (float(m.group(3)) * 1024.0, int(m.group(2)))
    return out


def total(d, pred, launches_per_step=1):
    t = 0.0
    for k, v in d.items():
        if pred(k) and "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            t += (v["FETCH_SIZE"][0] + v["WRITE_SIZE"][0]) * launches_per_step(k, v) if callable(launches_per_step) else (v["FETCH_SIZE"][0] + v["WRITE_SIZE"][0])
    return t or None


tj = {}
note = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of `bench.py --no-cpu --steps 2 --warmup 1` (tools/pmc_run.sh), "
        "(FETCH_SIZE + WRITE_SIZE) * 1024 per launch; FETCH_SIZE not doubled (16-byte gathers, not wide coalesced reads); "
        "FETCH_SIZE counts Infinity-Cache hits too (MI355X_MICROARCH.md), so this is fabric traffic, an upper bound of HBM traffic")
for w, nb in (("cfg2", 65536), ("cfg3", 1 << 20), ("cfg4", 16384), ("cfg5", 1024)):
    d = pmc(w)
    if not d:
        continue
    dec = total(d, lambda k: "k_decompress_safe" in k)
    if w == "cfg4":
        # the HC pipeline runs its three kernels once per round; n / (steps = 3 incl. warm-up and the untimed pass ... ) is
        # not needed: sum of per-launch means times launches per compress call (8 rounds of 2048 blocks)
        comp = 0.0
        for k, v in d.items():
            if "k_hc_" in k and "FETCH_SIZE" in v and "WRITE_SIZE" in v:
                comp += (v["FETCH_SIZE"][0] + v["WRITE_SIZE"][0]) * 8
        comp = comp or None
    else:
        comp = total(d, lambda k: "k_compress_fast" in k)
    tj["%s/text/%d" % (w, nb)] = {"compress": comp, "decompress": dec,
                                  "source": "profiles/%s_%s_text_pmc_traffic.txt: %s" % (pre, w, note)}
json.dump(tj, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
print(json.dumps(tj, indent=1)[:1500])

#!/usr/bin/env python3
"""Copy one round's GPU-box outputs (tools/refresh_profiles.sh <tag>) from gpurun_out/<tag>/ into profiles/ and rebuild
profiles/traffic.json (fabric bytes per compress / decompress CALL of the dominant kernels, from the PMC passes).
usage: tools/collect_profiles.py <tag> <round-prefix, e.g. r03>

Traffic per call = sum over the call's kernels of (2 x FETCH_SIZE + WRITE_SIZE) x 1024 x launches per call:
  * FETCH_SIZE is doubled for every kernel: profiles/r03_gather_calibration.md shows that on gfx950 an L2 miss is one
    128-byte fabric request tallied at 64 bytes, for 16-byte gathers exactly as for wide coalesced reads;
  * launches per call come from the PMC pass itself (dispatch count of the kernel / dispatch count of the decoder, which
    runs once per bench step) -- the HC pipeline launches its kernels once per round, and the number of rounds is a
    tuning choice that has changed between rounds;
  * keys are workload/distribution/blocks, with /L<level> appended for cfg4 (the HC levels run different kernels)."""
import glob, json, os, re, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, pre = sys.argv[1], sys.argv[2]
src, dst = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles")
for f in glob.glob(os.path.join(src, "*_bench.json")):
    if os.path.getsize(f):
        shutil.copy(f, os.path.join(dst, "%s_%s" % (pre, os.path.basename(f))))
RUNS = (("cfg2", 65536, ""), ("cfg3", 1 << 20, ""), ("cfg4", 16384, "/L9"), ("cfg5", 1024, ""),
        ("cfg4_level2", 16384, "/L2"), ("cfg4_level12", 16384, "/L12"))
for w, _, _ in RUNS:
    k = os.path.join(src, "prof_" + w, "kernel_stats_zlz4.csv")
    if os.path.exists(k):
        shutil.copy(k, os.path.join(dst, "%s_%s_text_kernel_stats.csv" % (pre, w)))
    p = os.path.join(src, "pmc_" + w, "summary.txt")
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, "%s_%s_text_pmc_traffic.txt" % (pre, w)))
if os.path.exists(os.path.join(src, "gpu_tests.txt")):
    shutil.copy(os.path.join(src, "gpu_tests.txt"), os.path.join(dst, pre + "_gpu_tests.txt"))


def pmc(w):
    """kernel -> {counter: (mean bytes per launch, dispatches)}   (rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB)"""
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


def per_call(d, pred, calls):
    """(fabric bytes per call, {kernel: launches per call}) over the kernels that satisfy pred"""
    t, lp = 0.0, {}
    for k, v in d.items():
        if pred(k) and "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            n = v["FETCH_SIZE"][1] / float(calls)
            lp[k] = n
            t += (2.0 * v["FETCH_SIZE"][0] + v["WRITE_SIZE"][0]) * n
    return (t or None), lp


tj = {}
note = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of `bench.py --no-cpu --steps 2 --warmup 1` (tools/pmc_run.sh); "
        "(2 x FETCH_SIZE + WRITE_SIZE) x 1024 x launches per call, FETCH_SIZE doubled per profiles/r03_gather_calibration.md; "
        "FETCH_SIZE counts Infinity-Cache hits too (MI355X_MICROARCH.md), so this is fabric traffic, an upper bound of HBM traffic")
for w, nb, suffix in RUNS:
    d = pmc(w)
    if not d:
        continue
    deck = [k for k in d if "k_decompress_safe" in k and "FETCH_SIZE" in d[k]]
    calls = d[deck[0]]["FETCH_SIZE"][1] if deck else 4
    dec, _ = per_call(d, lambda k: "k_decompress_safe" in k, calls)
    if w.startswith("cfg4"):
        comp, lp = per_call(d, lambda k: "k_hc_" in k or "fillBuffer" in k, calls)
    elif w == "cfg3":
        comp, lp = None, {}                       # decompress only (the untimed pre-compression is not a bench step)
    else:
        comp, lp = per_call(d, lambda k: "k_compress_fast" in k, calls)
    tj["%s/text/%d%s" % (w.split("_")[0], nb, suffix)] = {
        "compress": comp, "decompress": dec, "launches_per_compress_call": lp,
        "source": "profiles/%s_%s_text_pmc_traffic.txt: %s" % (pre, w, note)}
json.dump(tj, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
print(json.dumps(tj, indent=1)[:3000])

#!/usr/bin/env python3
"""Copy what `profiles/final_r02.sh` left under gpurun_out/ into the tracked files under profiles/ (names per round) and
rebuild profiles/ed_traffic.json (what bench.py reports as roofline.traffic) from the PMC passes of the same run."""
import json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
pairs = {
    "prof_r02/bench_f1_kernel_stats.csv": "r02_bench_kernel_stats.csv",
    "prof_r02/bench_f3_kernel_stats.csv": "r02_bench_inflight3_kernel_stats.csv",
    "prof_r02/bench_dtw_f1_kernel_stats.csv": "r02_bench_cnsm_dtw_kernel_stats.csv",
    "prof_r02/ed_trace_kernel_stats.csv": "r02_ed_kernel_stats.csv",
    "prof_r02/dtw_trace_kernel_stats.csv": "r02_dtw_kernel_stats.csv",
    "prof_r02/summary.json": "r02_rocprof_summary.json",
    "prof_r02/bench_f1.json": "r02_bench_under_rocprof_inflight1.json",
    "prof_r02/bench_f3.json": "r02_bench_under_rocprof_inflight3.json",
    "prof_r02/bench_dtw_f1.json": "r02_bench_cnsm_dtw_under_rocprof.json",
    "final_bench_rsm_ed.json": "r02_bench_n1.json",
    "final_bench_cnsm_ed.json": "r02_bench_cnsm_ed.json",
    "final_bench_cnsm_dtw.json": "r02_bench_cnsm_dtw.json",
    "final_bench_mixed.json": "r02_bench_mixed.json",
}
for src, dst in pairs.items():
    a, b = os.path.join(G, src), os.path.join(P, dst)
    if not os.path.exists(a):
        print("missing", src); continue
    if src.endswith(".json") and not src.endswith("summary.json"):
        lines = [l for l in open(a) if l.startswith("{")]
        open(b, "w").write(lines[-1])
    else:
        shutil.copyfile(a, b)
    print(dst)
s = json.load(open(os.path.join(P, "r02_rocprof_summary.json")))
key = lambda d: next(k for k in d if "kvm_ed_dot_kernel<false, false>" in k)
fetch = s["bench_fetch"][key(s["bench_fetch"])]["FETCH_SIZE"]
write = s["bench_write"][key(s["bench_write"])]["WRITE_SIZE"]
cal = s["calib_fetch"]
ck = next(k for k in cal if "ed_wave_kernel" in k)
corr = 400_031_744 / (cal[ck]["FETCH_SIZE"] * 1024.0)  # calib mode of prof_workloads.py: that many bytes must come from HBM
old = json.load(open(os.path.join(P, "ed_traffic.json")))
print("fetch correction from this run's calibration pass: %.4f (stored %.4f)" % (corr, old["fetch_correction"]))
old["source"] = ("profiles/prof_r02.sh (round 2, last rerun: after the launcher's launch model was re-fitted): rocprofv3 --pmc FETCH_SIZE and "
                 "--pmc WRITE_SIZE, separate passes over `bench.py --in-flight 1 --steps 3 --warmup 1 --extras 0 --cpu-baseline 0`, "
                 "averaged over the launches of the kernel; collected into this file by profiles/collect_r02.py")
old.update(fetch_size_kb=fetch, write_size_kb=write, fetch_correction=corr,
           hbm_bytes_per_launch=fetch * 1024.0 * corr + write * 1024.0)
json.dump(old, open(os.path.join(P, "ed_traffic.json"), "w"), indent=1)
print("ed_traffic.json: %.4f GB per launch" % (old["hbm_bytes_per_launch"] / 1e9))

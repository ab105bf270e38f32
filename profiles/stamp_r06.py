#!/usr/bin/env python3
"""Files what profiles/prof_r06.sh measured under profiles/ and prints every figure DESIGN.md section 0 quotes.
    python3 profiles/stamp_r06.py gpurun_out/prof_r06
Writes  profiles/ed_traffic.json   HBM bytes per launch of the ED pruner, RSM and cNSM (FETCH_SIZE / WRITE_SIZE passes), stamped with
                                   the hash of kvm_ed_dot.hip: bench.py quotes them as `roofline.traffic` only while the hash matches
        profiles/dtw_pmc.json      LDS bank conflicts, occupancy and VALU counters of the banded DP on the 66 762-DP slice, stamped
                                   with the hashes of kvm_dtw.hip / kvm_dtw_dp.h
        profiles/r06_*             the bench lines, kernel-trace summaries and logs themselves"""
import glob, hashlib, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "prof_r06")
sha = lambda f: hashlib.sha256(open(os.path.join(ROOT, "kv-match_amd", "csrc", f), "rb").read()).hexdigest()[:16]  # noqa: E731
FETCH_CORRECTION = 1.9914887868430706   # measured in round 2 on this kernel family's own access shape (profiles/prof_workloads.py calib:
                                        # isolated windows, 400 031 744 B must come from HBM); reproduces the guide's gfx950 x2 under-count


def load(name):
    p = os.path.join(src, name)
    return json.load(open(p)) if os.path.exists(p) and os.path.getsize(p) > 2 else {}


def counters(pmc, group, needle):
    """counter -> mean per launch, summed over the kernels whose name holds `needle`"""
    out = {}
    for k, c in (pmc.get(group) or {}).items():
        if needle in k:
            for a, b in c.items():
                out[a] = out.get(a, 0.0) + b
    return out


def traffic(pmc, needles, label):
    f = sum(counters(pmc, "fetch", nd).get("FETCH_SIZE", 0.0) for nd in needles)
    w = sum(counters(pmc, "write", nd).get("WRITE_SIZE", 0.0) for nd in needles)
    if not f:
        return None
    return {"hbm_bytes_per_launch": f * 1024.0 * FETCH_CORRECTION + w * 1024.0, "kernel": label, "fetch_size_kb": f, "write_size_kb": w,
            "fetch_correction": FETCH_CORRECTION, "kvm_ed_dot_sha16": sha("kvm_ed_dot.hip"),
            "source": "profiles/prof_r06.sh (round 6): rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in passes of their own over `bench.py --steps 2 "
                      "--warmup 1 --queries-per-step 20 --extras 0 --cpu-baseline 0 --in-flight 1 --sub-workloads 0 --workload <w>`, averaged over the "
                      "launches of the kernel(s); FETCH_SIZE x the correction factor measured in round 2 on this kernel family's own access shape"}


rsm, cnsm, dtw = load("pmc_rsm_ed.json"), load("pmc_cnsm_ed.json"), load("pmc_dtw.json")
out = {}
t = traffic(rsm, ["kvm_ed_dot_kernel<false, false>"], "kvm_ed_dot_kernel<false,false>")
if t:
    out.update(t)
t = traffic(cnsm, ["kvm_ed_dot_kernel<true, false>", "kvm_retile_cnsm_kernel"], "kvm_ed_dot_kernel<true,false> + kvm_retile_cnsm_kernel")
if t:
    out["cnsm_ed"] = t
if out:
    json.dump(out, open(os.path.join(ROOT, "profiles", "ed_traffic.json"), "w"), indent=1)
    print("ed_traffic.json: RSM %.4g B per launch, cNSM %s" % (out.get("hbm_bytes_per_launch", 0), (out.get("cnsm_ed") or {}).get("hbm_bytes_per_launch")))
for tag, pmc in (("rsm_ed", rsm), ("cnsm_ed", cnsm)):
    for g in ("sq", "lds"):
        c = counters(pmc, g, "kvm_ed_dot_kernel")
        if c:
            print("pmc", tag, g, {a: round(b) for a, b in c.items()})
d = {}
for g in ("sq", "lds"):
    c = counters(dtw, g, "kvm_dtw_dp")
    if c:
        d[g] = c
if d:
    lds, sq = d.get("lds", {}), d.get("sq", {})
    d["derived"] = {
        "lds_bank_conflict_share_of_idx_active": lds.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, lds.get("SQ_LDS_IDX_ACTIVE", 0)),
        "valu_insts_per_dp": sq.get("SQ_INSTS_VALU", 0) / 66762.0 if sq.get("SQ_INSTS_VALU") else None,
        "waves_per_simd": sq.get("SQ_WAVES", 0) / 1024.0 if sq else None,   # (persistent waves: each pops DPs from the list until it is drained)
        "valu_active_share_of_wave_cycles": sq.get("SQ_ACTIVE_INST_VALU", 0) / max(1.0, sq.get("SQ_WAVE_CYCLES", 0)) if sq else None,
        "wait_inst_any_share_of_wave_cycles": sq.get("SQ_WAIT_INST_ANY", 0) / max(1.0, sq.get("SQ_WAVE_CYCLES", 0)) if sq else None}
    d["kvm_dtw_sha16"] = sha("kvm_dtw.hip")
    d["kvm_dtw_dp_h_sha16"] = sha("kvm_dtw_dp.h")
    d["workload"] = "profiles/dp_slice.py: cNSM-DTW m = 4096, rho = 204, the 66 762 candidates of bench.py's roofline_dtw.full_dp slice, cascade and abandoning off"
    d["source"] = "profiles/prof_r06.sh: one --pmc group per pass, never together with tracing; counters are per launch of kvm_dtw_dp_reg_kernel"
    ks = os.path.join(src, "trace_dtw_kernel_stats.csv")
    if os.path.exists(ks):
        import csv
        d["kernel_trace"] = [{"kernel": r["Name"][:60], "calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6}
                             for r in csv.DictReader(open(ks)) if "dtw" in r["Name"]]
    json.dump(d, open(os.path.join(ROOT, "profiles", "dtw_pmc.json"), "w"), indent=1)
    print("dtw_pmc.json:", json.dumps(d["derived"]))
for f in glob.glob(os.path.join(src, "*.json")) + glob.glob(os.path.join(src, "*.log")) + glob.glob(os.path.join(src, "*kernel_stats*.csv")):
    shutil.copy(f, os.path.join(ROOT, "profiles", "r06_" + os.path.basename(f)))
# ---- the figures of DESIGN.md section 0
try:
    line = json.loads(open(os.path.join(src, "bench_default.json")).read().strip().splitlines()[-1])
    r = line["roofline"]
    print("headline %.3g cand/s; kernel %.4f ms frac %.3f (executed share %s); early abandon %.3g; group world 1 %s" % (
        line["value"], r["kernel_ms"], r["frac"], r.get("executed_fma_share"), line["value_early_abandon"],
        {k: (round(v, 3) if isinstance(v, float) else v) for k, v in (line.get("group_world1") or {}).items() if k != "note"}))
    for w, c in (line.get("configs") or {}).items():
        rr = c.get("roofline") or {}
        print(" config %s: %.3g cand/s (abandon %.3g), kernel %.4f ms frac %.3f, traffic %s, cpu_baseline %s" % (
            w, c.get("value") or 0, c.get("value_early_abandon") or 0, rr.get("kernel_ms") or 0, rr.get("frac") or 0, rr.get("traffic"),
            {k: v for k, v in (c.get("cpu_baseline") or {}).items() if k in ("value", "cores", "value_early_abandon")}))
    fd = (line.get("roofline_dtw") or {}).get("full_dp") or {}
    print(" full DP slice: %.3f ms frac %.3f" % (fd.get("kernel_ms") or 0, fd.get("frac") or 0))
    print(" device rates:", line.get("device_rates_measured"))
except Exception as e:  # noqa: BLE001
    print("no bench line:", repr(e))

import csv, glob, collections, sys, json, os
base = sys.argv[1]
out = {}
for d in sorted(glob.glob(os.path.join(base, "*"))):
    if not os.path.isdir(d): continue
    tag = os.path.basename(d)
    st = glob.glob(d + "/*/*_kernel_stats.csv")
    cc = glob.glob(d + "/*/*_counter_collection.csv")
    if st:
        rows = list(csv.DictReader(open(st[0])))
        out[tag] = [{ "kernel": r["Name"][:70], "calls": int(r["Calls"]), "avg_us": float(r["AverageNs"])/1e3, "pct": float(r["Percentage"])} for r in rows if float(r["Percentage"]) > 0.5]
    if cc:
        rows = list(csv.DictReader(open(cc[0])))
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in rows:
            agg[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        out[tag] = {k: {c: sum(v)/len(v) for c, v in d2.items()} for k, d2 in agg.items() if not k.startswith("__amd")}
print(json.dumps(out, indent=1))

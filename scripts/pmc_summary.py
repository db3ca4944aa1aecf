import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for r in rows:
    k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ao::", "")[:28]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k, v in sorted(agg.items()):
    if k.startswith("k_"):
        print(f"{k:28s} n={len(n[k]):4d} " + " ".join(f"{c}={x/len(n[k]):.4g}" for c, x in sorted(v.items())))

"""developer tool: fold rocprofv3 --pmc CSVs (one row per dispatch and counter) into per-kernel means"""
import csv, glob, os, re, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "*", "*", "*counter_collection.csv")) + glob.glob(os.path.join(root, "pmc_*", "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"<.*>", "", r["Kernel_Name"].split("(")[0]).replace("void ", "").strip()   # template instantiations fold into one row
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for k in acc.values() for c in k})
print("kernel," + ",".join(names))
for k, d in sorted(acc.items()):
    if not k.startswith("k_"):
        continue
    print(k + "," + ",".join(f"{sum(d[c])/len(d[c]):.4g}" if c in d else "" for c in names))

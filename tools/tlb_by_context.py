"""developer tool: per-dispatch duration and UTCL1 counters of one kernel from a rocprofv3 --kernel-trace --pmc run of
tools/kbench.py (several contexts taking turns) -- do slow contexts miss more in the TLB?
usage: python3 tools/tlb_by_context.py <dir> <kernel substring>"""
import csv, glob, os, sys, collections
root, pat = sys.argv[1], sys.argv[2]
cc = glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)[0]
kt = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)[0]
dur = {}
for r in csv.DictReader(open(kt)):
    if pat in r["Kernel_Name"]:
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
cnt = collections.defaultdict(dict)
for r in csv.DictReader(open(cc)):
    if pat in r["Kernel_Name"]:
        cnt[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
names = sorted({c for d in cnt.values() for c in d})
print("dispatch  ms   " + " ".join(n.replace("TCP_UTCL1_", "")[:18].rjust(18) for n in names))
for d in sorted(cnt, key=int):
    print(f"{d:>8s} {dur.get(d, 0):6.3f} " + " ".join(f"{cnt[d].get(n, 0):18.4g}" for n in names))

import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
ks=["GRBM_GUI_ACTIVE","TCP_UTCL1_REQUEST_sum","TCP_UTCL1_TRANSLATION_HIT_sum","TCP_UTCL1_TRANSLATION_MISS_sum","TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum","TCP_UTCL1_STALL_LFIFO_NO_RES_sum","TCP_UTCL1_STALL_INFLIGHT_MAX_sum"]
print("kernel".ljust(22), " ".join(k.replace("TCP_UTCL1_","").replace("_sum","")[:12].rjust(12) for k in ks))
for r in rows:
    if any(s in r["kernel"] for s in sys.argv[2].split(",")):
        print(r["kernel"][:22].ljust(22), " ".join((r.get(k) or "-").rjust(12) for k in ks))

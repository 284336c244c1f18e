import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(sys.argv[1] + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
rows = sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_LDS_IDX_ACTIVE", 0))
for k, c in rows[:18]:
    a, b = c.get("SQ_LDS_IDX_ACTIVE", 0), c.get("SQ_LDS_BANK_CONFLICT", 0)
    print(f"{k:72s} lds_active {a:10.3e} conflict {b:10.3e} {100 * b / a if a else 0:5.1f} %  wait_lds {c.get('SQ_WAIT_INST_LDS', 0):.3e}")

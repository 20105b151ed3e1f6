"""Summarise rocprofv3 --pmc CSVs per kernel: mean counter value per dispatch."""
import csv, glob, sys, collections, json
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][-48:]
        a = acc[k][row["Counter_Name"]]
        a[0] += float(row["Counter_Value"]); a[1] += 1
res = {}
for k, d in acc.items():
    if "swk" not in k: continue
    res[k] = {c: (v[0] / v[1], v[1]) for c, v in d.items()}
for k, d in sorted(res.items()):
    print(k)
    for c, (m, n) in sorted(d.items()):
        print("   %-28s mean/dispatch %16.1f   dispatches %d" % (c, m, n))
if len(sys.argv) > 2:
    json.dump({k: {c: v[0] for c, v in d.items()} for k, d in res.items()}, open(sys.argv[2], "w"), indent=1)

"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel: mean counter value per dispatch."""
import csv, glob, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].split("(")[0].replace("void ", "")
        if "bogp" not in name: continue
        acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print("   %-34s n=%4d mean=%.6g" % (c, len(v), sum(v) / len(v)))

"""Sum PMC counters per kernel from rocprofv3 --pmc CSV output dirs: python tools/pmc_kernels.py DIR [substr]"""
import collections, csv, glob, sys
rows = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0][-40:]
        if len(sys.argv) > 2 and sys.argv[2] not in r["Kernel_Name"]:
            continue
        rows[n][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(n, r["Counter_Name"])] += 1
for n, cs in rows.items():
    print(n)
    for c, v in sorted(cs.items()):
        k = cnt[(n, c)]
        print(f"   {c:32s} {v / k:16.1f} per dispatch  ({k} dispatches)")

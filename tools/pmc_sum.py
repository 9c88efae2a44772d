"""Sum rocprofv3 --pmc counters per kernel: python tools/pmc_sum.py DIR [name-filter] [n_sequences]"""
import csv, glob, sys, collections
d = sys.argv[1]; filt = sys.argv[2] if len(sys.argv) > 2 else "decompress"; nseq = float(sys.argv[3]) if len(sys.argv) > 3 else 0
tot = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if filt not in k: continue
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[(k, r["Counter_Name"])] += 1
for k, cs in tot.items():
    print(k[:90])
    for c, v in sorted(cs.items()):
        n = calls[(k, c)]
        s = f"  {c:28s} {v:16.0f}  ({n} dispatches)"
        if nseq: s += f"   per sequence {v / n / nseq:8.3f}"
        print(s)

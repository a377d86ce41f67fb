"""Average the counters of tools/pmc_stalls.sh per kernel.  Usage: python tools/pmc_table.py <tag>"""
import csv, glob, os, sys
from collections import defaultdict
tag = sys.argv[1]
repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
acc = defaultdict(lambda: defaultdict(list))
for path in glob.glob(os.path.join(repo, "gpurun_out", tag + "_*", "**", "*_counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"]
        if "frontend_kernel" not in k:
            continue
        k = "nfft1024" if "true, 8" in k.replace("(bool)1", "true") or "Lb1E" in k else "nfft512"
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, c in acc.items():
    print(k)
    for name in sorted(c):
        print(f"  {name:32s} {sum(c[name]) / len(c[name]):16.1f}  (n={len(c[name])})")

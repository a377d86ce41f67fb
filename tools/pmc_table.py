"""Average the counters of tools/pmc_stalls.sh per kernel.  Usage: python tools/pmc_table.py <tag> [kernel substring]"""
import csv, glob, os, sys
from collections import defaultdict
tag = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else "frontend_kernel"
repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
acc = defaultdict(lambda: defaultdict(list))
for path in glob.glob(os.path.join(repo, "gpurun_out", tag + "_*", "**", "*_counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"]
        if want not in k:
            continue
        if want == "frontend_kernel":
            k = "nfft1024" if "true, 8" in k.replace("(bool)1", "true") or "Lb1E" in k else "nfft512"
        else:
            k = k[:60] + " grid=" + row["Grid_Size"]
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, c in acc.items():
    print(k)
    for name in sorted(c):
        print(f"  {name:32s} {sum(c[name]) / len(c[name]):16.1f}  (n={len(c[name])})")

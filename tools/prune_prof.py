#!/usr/bin/env python3
"""Shrink rocprofv3 output under gpurun_out/ before it travels back (64 MiB limit): drop the per-dispatch kernel
traces (the *_kernel_stats.csv summaries stay) and keep only the rows of libsvk kernels in the counter files."""
import csv
import glob
import os
import sys

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
KEEP = ("c3d2_", "frontend_kernel", "cosine", "cmvn", "vad_", "cube_", "draw_crops", "inv_norm", "decimate",
        "resample_kernel", "spectrum_", "mel_features", "roc", "l2_dist", "fc5_")
for path in glob.glob(os.path.join(root, "**", "*_kernel_trace.csv"), recursive=True):
    os.remove(path)
for path in glob.glob(os.path.join(root, "**", "*_counter_collection.csv"), recursive=True):
    rows = list(csv.reader(open(path)))
    if not rows:
        continue
    col = rows[0].index("Kernel_Name")
    kept = [rows[0]] + [r for r in rows[1:] if any(k in r[col] for k in KEEP)]
    with open(path, "w", newline="") as fh:
        csv.writer(fh).writerows(kept)

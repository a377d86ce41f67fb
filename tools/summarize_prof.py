#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (gpurun_out/...) into the small files kept under profiles/.

    python tools/summarize_prof.py <round-tag>      e.g.  r01

Reads  gpurun_out/prof_bench/**/_kernel_stats.csv          (rocprofv3 --kernel-trace --stats -- bench.py)
       gpurun_out/pmc_{fetch,write,sq}/**/_counter_collection.csv   (one --pmc pass each, bench.py --frontend-only)
Writes profiles/<tag>_bench_kernel_stats.csv   (verbatim top rows of the stats table)
       profiles/<tag>_frontend_pmc.json        (per-launch averages for our kernels + derived HBM traffic)
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 64 B per 128 B request for wide
coalesced reads, so read bytes = 2 x FETCH_SIZE x 1024 (MI355X_MICROARCH.md, HBM section).
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OURS = ("frontend_kernel", "cmvn_kernel", "vad_kernel", "vad_small_kernel", "vad_flags_kernel", "vad_walk_kernel", "vad_copy_kernel",
        "cube_gather_kernel", "cosine_kernel", "cosine_tiled_kernel",
        "inv_norm_kernel", "draw_crops_kernel", "decimate_kernel", "resample_kernel",
        "c3d2_stage1h_kernel", "c3d2_conv21h_kernel", "c3d2_conv22h_kernel", "c3d2_conv31h_kernel", "c3d2_conv32h_kernel", "c3d2_conv41h_kernel",
        "cmvnw_kernel", "spectrum_pow2_kernel",
        "spectrum_dft_kernel", "spectrum_fft_kernel", "mel_features_kernel", "c3d2_tail_kernel", "fc5_reduce_kernel", "fc5_kernel")


def short(name):
    for k in OURS:
        if k in name:
            extra = ""
            if k == "c3d2_tail_kernel":
                extra = "<Conv42>"
            if "frontend_kernel" in name:
                extra = "<int16,nfft1024>" if "<short, true" in name else "<int16,nfft512>" if "<short, false" in name \
                    else "<f32,nfft1024>" if "<float, true" in name else "<f32,nfft512>"
            return k + extra
    return None


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    out_dir = os.path.join(REPO, "profiles")
    os.makedirs(out_dir, exist_ok=True)
    stats = sorted(glob.glob(os.path.join(REPO, "gpurun_out", "prof_bench", "**", "*_kernel_stats.csv"), recursive=True),
                   key=os.path.getmtime)
    if stats:
        rows = list(csv.reader(open(stats[-1])))          # newest run
        keep = [rows[0]] + [r for r in rows[1:] if float(r[4]) >= 0.2 or short(r[0])]
        for r in keep[1:]:
            if len(r[0]) > 160:
                r[0] = r[0][:157] + "..."
        with open(os.path.join(out_dir, f"{tag}_bench_kernel_stats.csv"), "w", newline="") as fh:
            csv.writer(fh).writerows(keep)
        print("wrote", f"profiles/{tag}_bench_kernel_stats.csv", len(keep) - 1, "kernels")
    ev = sorted(glob.glob(os.path.join(REPO, "gpurun_out", "prof_evaluate", "**", "*_kernel_stats.csv"), recursive=True),
                key=os.path.getmtime)
    if ev:                                                   # tools/profile_evaluate.py: every kernel create_speaker_models() + evaluate() ran
        rows = list(csv.reader(open(ev[-1])))
        for r in rows[1:]:
            if len(r[0]) > 160:
                r[0] = r[0][:157] + "..."
        with open(os.path.join(out_dir, f"{tag}_evaluate_kernel_stats.csv"), "w", newline="") as fh:
            csv.writer(fh).writerows(rows)
        bad = [r[0] for r in rows[1:] if any(k in r[0] for k in ("ck::", "naive_conv", "miopen", "MIOpen", "Cijk_")) or ("at::native" in r[0] and "conv" in r[0].lower())]
        print("wrote", f"profiles/{tag}_evaluate_kernel_stats.csv", len(rows) - 1, "kernels;", "framework convolution / GEMM rows:", bad or "none")
    summary = defaultdict(lambda: defaultdict(list))
    meta = {}
    # (c3d2_1 .. c3d2_3: the three stall-composition passes over `bench.py --c3d2-only`; they carry SQ_INSTS_MFMA, which
    # bench.py's roofline fractions are built on)
    for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_fetch_c3d2", "pmc_write_c3d2", "pmc_sq_c3d2", "c3d2_1", "c3d2_2", "c3d2_3"):
        paths = sorted(glob.glob(os.path.join(REPO, "gpurun_out", sub, "**", "*_counter_collection.csv"),
                                 recursive=True), key=os.path.getmtime)
        for path in paths[-1:]:                             # newest run only
            for row in csv.DictReader(open(path)):
                k = short(row["Kernel_Name"])
                if not k:
                    continue
                summary[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
                summary[k].setdefault("duration_ns_" + (sub.replace("_c3d2", "") if sub.startswith("pmc") else "pmc_sq"), []).append(
                    int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
                meta[k] = {"grid_size": int(row["Grid_Size"]), "workgroup_size": int(row["Workgroup_Size"]),
                           "vgpr": int(row["VGPR_Count"]), "accum_vgpr": int(row["Accum_VGPR_Count"]),
                           "sgpr": int(row["SGPR_Count"]), "lds_block_size": int(row["LDS_Block_Size"])}
    result = {}
    for k, counters in summary.items():
        rec = {name: sum(v) / len(v) for name, v in counters.items()}
        rec["launches_seen"] = max(len(v) for v in counters.values())
        rec.update(meta[k])
        if "FETCH_SIZE" in rec and "WRITE_SIZE" in rec:
            rec["hbm_read_bytes_per_launch"] = 2.0 * rec["FETCH_SIZE"] * 1024.0
            rec["hbm_write_bytes_per_launch"] = rec["WRITE_SIZE"] * 1024.0
            rec["hbm_traffic_bytes_per_launch"] = rec["hbm_read_bytes_per_launch"] + rec["hbm_write_bytes_per_launch"]
        if "SQ_VALU_MFMA_BUSY_CYCLES" in rec and rec.get("duration_ns_pmc_sq"):
            # the counter sums busy cycles over the chip's 1024 SIMDs; 2.4 GHz is the nominal engine clock
            rec["mfma_busy_fraction"] = rec["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * 2.4 * rec["duration_ns_pmc_sq"])
        if "SQ_LDS_BANK_CONFLICT" in rec and rec.get("SQ_LDS_IDX_ACTIVE"):
            rec["lds_conflict_fraction"] = rec["SQ_LDS_BANK_CONFLICT"] / rec["SQ_LDS_IDX_ACTIVE"]
        result[k] = rec
    if result:
        # what the counters were collected from (written on the GPU box by tools/refresh_profiles.sh) + the commit this summary
        # is made at: bench.py compares csrc_sha with the sources it runs and marks its roofline rows `stale` when they differ
        prov = {}
        ppath = os.path.join(REPO, "gpurun_out", "prof_provenance.json")
        if os.path.exists(ppath):
            prov = json.load(open(ppath))
        import subprocess as sp
        try:
            prov["git_head"] = sp.run(["git", "-C", REPO, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip()
            prov["git_dirty_csrc"] = bool(sp.run(["git", "-C", REPO, "status", "--porcelain", "--", "speaker_verification_amd/csrc", "include"],
                                                 capture_output=True, text=True).stdout.strip())
        except OSError:
            pass
        sys.path.insert(0, REPO)
        from speaker_verification_amd import _lib
        prov["csrc_sha_at_summary"] = _lib.provenance()["csrc_sha"]
        result["_provenance"] = prov
        result["_note"] = ("per-launch averages from separate rocprofv3 --pmc passes over `bench.py --frontend-only` "
                           "(1024 x 3 s clips per launch) and, for the c3d2_* kernels, `bench.py --c3d2-only` (1024 cubes "
                           "per launch); FETCH_SIZE/WRITE_SIZE in KiB, read side doubled per the gfx950 correction")
        with open(os.path.join(out_dir, f"{tag}_frontend_pmc.json"), "w") as fh:
            json.dump(result, fh, indent=1, sort_keys=True)
        print(json.dumps(result, indent=1, sort_keys=True))
    # the bench lines of the same pass and the stall-composition table (tools/refresh_profiles.sh)
    import shutil
    import subprocess
    for src, dst in (("bench_final.json", f"{tag}_bench.json"), ("bench_under_rocprof.json", f"{tag}_bench_under_rocprof.json")):
        path = os.path.join(REPO, "gpurun_out", src)
        if os.path.exists(path) and os.path.getsize(path) > 0:
            shutil.copyfile(path, os.path.join(out_dir, dst))
            print("copied", dst)
    if glob.glob(os.path.join(REPO, "gpurun_out", "final_1", "**", "*_counter_collection.csv"), recursive=True):
        table = subprocess.run([sys.executable, os.path.join(REPO, "tools", "pmc_table.py"), "final"],
                               capture_output=True, text=True).stdout
        with open(os.path.join(out_dir, f"{tag}_frontend_stalls.txt"), "w") as fh:
            fh.write("# per-launch averages, bench.py --frontend-only (1 024 x 3 s clips per launch), three separate\n"
                     "# rocprofv3 --pmc passes (tools/pmc_stalls.sh).  *_CYCLES of waves are in units of 4 clocks\n"
                     "# (SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_*); SQ_VALU_MFMA_BUSY_CYCLES is in clocks.\n" + table)
        print("wrote", f"{tag}_frontend_stalls.txt")
    if glob.glob(os.path.join(REPO, "gpurun_out", "c3d2_1", "**", "*_counter_collection.csv"), recursive=True):
        table = subprocess.run([sys.executable, os.path.join(REPO, "tools", "pmc_table.py"), "c3d2", "c3d2_"],
                               capture_output=True, text=True).stdout
        with open(os.path.join(out_dir, f"{tag}_c3d2_stalls.txt"), "w") as fh:
            fh.write("# per-launch averages, bench.py --c3d2-only (1 024 cubes per launch), three separate rocprofv3 --pmc\n"
                     "# passes (tools/pmc_stalls.sh c3d2 bench.py --c3d2-only).  SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* are\n"
                     "# in units of 4 clocks; SQ_VALU_MFMA_BUSY_CYCLES in clocks (32 per v_mfma_f32_16x16x4_f32).\n" + table)
        print("wrote", f"{tag}_c3d2_stalls.txt")
    path = os.path.join(REPO, "gpurun_out", "stages_final.json")
    if os.path.exists(path) and os.path.getsize(path) > 0:
        shutil.copyfile(path, os.path.join(out_dir, f"{tag}_stage_kernels.json"))


if __name__ == "__main__":
    main()

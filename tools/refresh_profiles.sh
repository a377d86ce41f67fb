#!/bin/bash
# Everything profiles/ is built from, in one pass on the GPU box (from the repo root):
#   bash tools/refresh_profiles.sh
# then, back in the build container:  python tools/summarize_prof.py rNN
# Separate rocprofv3 runs: kernel trace + stats on the end-to-end bench, one --pmc pass per counter
# group on `bench.py --frontend-only` (never --pmc together with other trace domains).
set -e
# every profiler pass is bounded: one that hangs (seen once: a --pmc pass stuck right after HSA initialisation, nothing of ours
# running) is cut after four minutes instead of starving the whole call of output
rp() { timeout -k 10 240 rocprofv3 "$@" || echo "profiler pass failed or timed out: rc=$?"; }
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
# what every file of this pass was measured with: sha256 over the kernel sources and of the built library (bench.py marks a
# roofline row `stale` when the committed counters come from other sources than the ones it runs)
python3 -c "import sys, json; sys.path.insert(0, '$root'); from speaker_verification_amd import _lib; print(json.dumps(_lib.provenance()))" > $out/prof_provenance.json
echo "[1/6] kernel trace + stats of the end-to-end bench"
rp --kernel-trace --stats --output-format csv -d $out/prof_bench -- python3 $root/bench.py --steps 3 --warmup 1 --no-extras > $out/bench_under_rocprof.json 2> $out/bench_under_rocprof.err
echo "[2/6] HBM read / write counters"
rp --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 $root/bench.py --frontend-only > $out/pmc_fetch.log 2>&1
rp --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 $root/bench.py --frontend-only > $out/pmc_write.log 2>&1
echo "[3/6] SQ counters"
rp --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/pmc_sq -- python3 $root/bench.py --frontend-only > $out/pmc_sq.log 2>&1
echo "[4/6] stall composition"
bash $root/tools/pmc_stalls.sh final
cd /tmp
echo "[5/6] the network kernels (all seven libsvk kernels): HBM counters + stall composition over bench.py --c3d2-only"
rp --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch_c3d2 -- python3 $root/bench.py --c3d2-only > $out/pmc_fetch_c3d2.log 2>&1
rp --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write_c3d2 -- python3 $root/bench.py --c3d2-only > $out/pmc_write_c3d2.log 2>&1
bash $root/tools/pmc_stalls.sh c3d2 bench.py --c3d2-only
cd /tmp
echo "[5b] the reference's file-driven entry points (create_speaker_models() + evaluate()) under the kernel trace"
rp --kernel-trace --stats --output-format csv -d $out/prof_evaluate -- python3 $root/tools/profile_evaluate.py > $out/prof_evaluate.log 2>&1
echo "[6/6] the bench itself + the stage-level kernels"
python3 $root/bench.py --steps 3 --warmup 1 > $out/bench_final.json 2> $out/bench_final.err
python3 $root/bench.py --stages-only > $out/stages_final.json 2> $out/stages_final.err
python3 $root/tools/prune_prof.py $out      # per-dispatch traces and other libraries' counter rows do not travel back
echo done

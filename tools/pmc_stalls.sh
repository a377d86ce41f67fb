#!/bin/bash
# Stall-composition counters for the front-end kernel (separate --pmc passes, kernel-trace only).
# Usage (on the GPU box, from the repo root): bash tools/pmc_stalls.sh <tag> [program args ...]
# (default program: bench.py --frontend-only)
set -e
rp() { timeout -k 10 240 rocprofv3 "$@" || echo "profiler pass failed or timed out: rc=$?"; }
tag=${1:-stalls}
shift || true
if [ $# -eq 0 ]; then set -- bench.py --frontend-only; fi
prog=$1; shift
root=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_IFETCH SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA" \
           "SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES"; do
  i=$((i+1))
  rp --kernel-trace --pmc $set --output-format csv -d $root/gpurun_out/${tag}_$i -- python3 $root/$prog "$@" > $root/gpurun_out/${tag}_$i.log 2>&1
done

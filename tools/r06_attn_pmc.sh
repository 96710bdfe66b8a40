#!/bin/bash
# round 6: which pipes of a CU the short-sequence attention forward keeps busy (1024 crops x 12 heads x 197 tokens), per kernel variant
# usage (GPU box): bash tools/r06_attn_pmc.sh   -> gpurun_out/r06_c/*.md
cd "$(dirname "$0")/.."
root=$(pwd); out=$root/gpurun_out/r06_c; mkdir -p "$out"
cd /tmp; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*\|TA_[A-Z_0-9]*\|TCP_[A-Z_0-9]*\|GRBM_[A-Z_0-9]*" | sort -u > "$out/counters_available.txt"
run() {  # tag, env..., counters
  tag=$1; shift; envs=$1; shift
  env $envs timeout 300 rocprofv3 --pmc "$@" -d "$out/p_$tag" -o r -- python $root/tools/attn_bench.py 1024 > "$out/p_$tag.log" 2>&1
  python $root/tools/rocprof_pmc.py "$out/p_$tag/r_results.db" | grep -v "at::native" > "$out/pmc_$tag.md"
  rm -rf "$out/p_$tag"
}
for v in "fwd4:CS_ATTN_FWD4=1" "fwd8:CS_ATTN_FWD4=0"; do
  name=${v%%:*}; e=${v#*:}
  run ${name}_busy "$e" SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_BUSY_CYCLES
  run ${name}_insts "$e" SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAVES
  run ${name}_wait "$e" SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
done
cd "$out"; for f in pmc_*.md; do echo "== $f"; cat $f; done

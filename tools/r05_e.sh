#!/bin/bash
# Round 5, fifth box: the other configurations after the attention changes (L/14-336 CLIPSelf, RegionCLIP bf16 / fp8, OpenAI ViT-B/16), kernel statistics
# of the recipe-shaped step, LDS counters of the new attention kernels.      bash tools/r05_e.sh <tag>
tag=${1:-r05e}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
timeout 600 python tools/l14_bench.py > "$out/l14_bench.txt" 2>&1
timeout 900 python tools/regionclip_bench.py 8 > "$out/regionclip_bench.txt" 2>&1
timeout 600 python tools/openai_vit_bench.py > "$out/openai_vit_bench.txt" 2>&1
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$out/prof_recipe" -o r -- python $root/tools/recipe_bench.py EVA02-CLIP-B-16 1024 2 4 > "$out/prof_recipe.log" 2>&1
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d "$out/pmc_lds" -o r -- python $root/tools/attn_long_bench.py 2 64 12 3 > "$out/pmc_lds.log" 2>&1
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE -d "$out/pmc_mfma" -o r -- python $root/tools/attn_long_bench.py 2 64 12 3 > "$out/pmc_mfma.log" 2>&1
cd "$root"
python tools/rocprof_summary.py "$out/prof_recipe/r_results.db" "$tag recipe shape after the attention changes: rocprofv3 --kernel-trace --stats -- python tools/recipe_bench.py EVA02-CLIP-B-16 1024 2 4, MI355X" > "$out/kernel_stats_recipe.md"
python tools/rocprof_pmc.py "$out/pmc_lds/r_results.db" > "$out/pmc_attention_lds.md"
python tools/rocprof_pmc.py "$out/pmc_mfma/r_results.db" > "$out/pmc_attention_mfma.md"
rm -rf "$out/prof_recipe" "$out/pmc_lds" "$out/pmc_mfma"
tail -3 "$out/l14_bench.txt" "$out/regionclip_bench.txt" "$out/openai_vit_bench.txt"; head -20 "$out/kernel_stats_recipe.md"; cat "$out/pmc_attention_lds.md" "$out/pmc_attention_mfma.md"

#!/bin/bash
# Fabric traffic (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, never with a trace domain) and bare time of the step's dominant
# launch under different tile rasters:  bash tools/raster_pmc.sh <tag>  -> gpurun_out/<tag>/raster_pmc.md
#   streaming kernel (cfg 11): M panels per raster group 2 / 4 / 8 (default) / 15 (flags bits 8-11): the 32 workgroups of an XCD then work on
#   2x16 / 4x8 / 8x4 / 15x2 (M x N) tiles at a time;  persistent kernel (cfg 9): grouped vs B-stationary raster with 1 / 2 / 4 N parts.
tag=${1:-raster}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
md="$out/raster_pmc.md"
echo "| kernel / raster | flags | us per launch (bare) | FETCH_SIZE x 2 (GB) | WRITE_SIZE (GB) | sum / 2.48 GB algorithmic |" > "$md"
echo "|---|---|---:|---:|---:|---:|" >> "$md"
specs=("stream gm=2:0x2B0" "stream gm=8 (default):0xB0" "stream gm=15:0xFB0" "persist grouped:0x90" "persist B-stationary/2:0x10290")
for spec in "${specs[@]}"; do
  name=${spec%%:*}; fl=${spec##*:}
  t=$(python $root/tools/raster_one.py $fl 2048 6 | sed -n 's/.*: *\([0-9.]*\) us per launch.*/\1/p')
  rm -rf /tmp/rp_f /tmp/rp_w
  rocprofv3 --pmc FETCH_SIZE -d /tmp/rp_f -o r -- python $root/tools/raster_one.py $fl 2048 3 > "$out/pmc_$fl.f.log" 2>&1
  rocprofv3 --pmc WRITE_SIZE -d /tmp/rp_w -o r -- python $root/tools/raster_one.py $fl 2048 3 > "$out/pmc_$fl.w.log" 2>&1
  python - "$name" "$fl" "$t" >> "$md" <<'PY'
import sqlite3, sys, glob
name, fl, t = sys.argv[1:4]
def avg(d, c):
    tot = n = 0
    for db in glob.glob(d + "/**/*.db", recursive=True):
        cur = sqlite3.connect(db).cursor()
        for k, v in cur.execute("select kernel_name, value from counters_collection where counter_name = ?", (c,)):
            if "gemm_stream_kernel" in k or "gemm_persist_kernel" in k:
                tot += v; n += 1
    return tot / max(n, 1)
f, w = avg("/tmp/rp_f", "FETCH_SIZE"), avg("/tmp/rp_w", "WRITE_SIZE")      # KB per launch
fg, wg = 2 * f * 1024 / 1e9, w * 1024 / 1e9
print(f"| {name} | {fl} | {t} | {fg:.2f} | {wg:.2f} | {(fg + wg) / 2.48:.2f} |")
PY
done
cd "$root"
cat "$md"

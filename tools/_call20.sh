out=gpurun_out/r03u
mkdir -p $out
rocm-smi --showclocks --showpower --showmaxpower 2>&1 | head -40 > $out/smi_idle.txt
(python bench.py --steps 150 --warmup 5 --no-cpu-baseline > $out/bench.json 2>/dev/null) &
BP=$!
sleep 25
for i in 1 2 3 4 5 6 7 8; do
  rocm-smi --showclocks --showpower 2>&1 | grep -i "sclk\|mclk\|fclk\|power\|socclk" >> $out/smi_load.txt
  echo "--" >> $out/smi_load.txt
  sleep 1
done
wait $BP
cut -c1-200 $out/bench.json
cat $out/smi_idle.txt | grep -i "sclk\|power\|mclk" | head
echo ==== ; cat $out/smi_load.txt | head -60

mkdir -p gpurun_out
s=$(date +%s)
(timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 --tunableop-file gpurun_out/tunableop_bench.csv > gpurun_out/r03_h_bench1.json 2> gpurun_out/r03_h_bench1.err)
e=$(date +%s); echo "bench run 1 (tuning the shapes the shipped file lacks): $((e-s)) s wall" > gpurun_out/r03_h_times.log
python tools/pretune_gemms.py --merge gpurun_out/tunableop_bench.csv >> gpurun_out/r03_h_times.log 2>&1
cp vidar_amd/tunableop_gfx950.csv gpurun_out/tunableop_gfx950_merged.csv
s=$(date +%s)
(timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --op-table > gpurun_out/r03_h_bench2.json 2> gpurun_out/r03_h_optable.txt)
e=$(date +%s); echo "bench run 2 (merged file shipped): $((e-s)) s wall" >> gpurun_out/r03_h_times.log
cat gpurun_out/r03_h_times.log; cut -c1-300 gpurun_out/r03_h_bench2.json

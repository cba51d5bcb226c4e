set -e
bash tools/fp64_counts.sh > gpurun_out/r5_fp64.log 2>&1 || { tail -20 gpurun_out/r5_fp64.log; exit 1; }
echo fp64 done
bash tools/profile_entf.sh > gpurun_out/r5_entf.log 2>&1 || true
echo entf done; head -20 gpurun_out/r5_entf.log
bash tools/profile_small_d.sh r5sd > gpurun_out/r5_small_d.log 2>&1 || true
echo small_d done; cat gpurun_out/r5_small_d.log
python bench.py > gpurun_out/r5_bench_final.json 2> gpurun_out/r5_bench_final.err || { tail -30 gpurun_out/r5_bench_final.err; exit 1; }
echo bench done

mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_trainer.py -m gpu -q -x > gpurun_out/r03_gputest8.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03_gputest8.log
tail -30 gpurun_out/r03_gputest8.log
for nt in "8192 64" "65536 32" "1048576 8"; do timeout 600 python profiles/tools/ppo_breakdown.py $nt >> gpurun_out/r03_ppo_breakdown.txt 2>&1; done
cat gpurun_out/r03_ppo_breakdown.txt

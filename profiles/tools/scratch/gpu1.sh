set -x
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r03_gputest1.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03_gputest1.log
tail -15 gpurun_out/r03_gputest1.log

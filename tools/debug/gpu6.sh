set -o pipefail
python -m pytest tests -m gpu -x -q > gpurun_out/r3_gpu_tests4.log 2>&1; echo "tests rc=$?"
tail -3 gpurun_out/r3_gpu_tests4.log
bash tools/profile_sides.sh r03 2>&1 | tail -15

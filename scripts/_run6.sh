set -x
timeout -k 10 400 python -m pytest tests/test_gpu_tiled.py -x -q -m gpu > gpurun_out/tiled4.log 2>&1; tail -4 gpurun_out/tiled4.log
bash scripts/prof_round2.sh r02 headline general_wave general_tiled configs4 batch512 > gpurun_out/prof_r02.log 2>&1; tail -5 gpurun_out/prof_r02.log

set -x
timeout -k 10 300 python -m pytest tests/test_gpu_tiled.py -x -q -m gpu > gpurun_out/tiled3_nw_default.log 2>&1; tail -5 gpurun_out/tiled3_nw_default.log
FMPC_TILED_NW=4 timeout -k 10 300 python -m pytest tests/test_gpu_tiled.py -x -q -m gpu > gpurun_out/tiled3_nw4.log 2>&1; tail -5 gpurun_out/tiled3_nw4.log
for nw in 2 4; do FMPC_TILED_NW=$nw timeout -k 10 200 python scripts/tiled_phases.py 27 30 1024 >> gpurun_out/phases3.log 2>&1; done
for nw in 4 8; do FMPC_TILED_NW=$nw timeout -k 10 200 python scripts/tiled_phases.py 65 60 256 >> gpurun_out/phases3.log 2>&1; done
cat gpurun_out/phases3.log

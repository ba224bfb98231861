#!/bin/bash
# A/B: v2 vs v3 main loop on the conv shapes + GPU parity tests of the igemm family
set -o pipefail
timeout -k 10 300 python -u -m pytest tests/test_igemm_gpu.py tests/test_i3d_gpu.py -x -q --timeout 120 --timeout-method thread > gpurun_out/ab_pytest.log 2>&1 || { tail -30 gpurun_out/ab_pytest.log; exit 1; }
tail -2 gpurun_out/ab_pytest.log
TRACE_IGEMM_V2=1 timeout -k 10 200 python -u scratch/bench_shapes.py conv > gpurun_out/ab_v2.log 2>&1 || exit 1
timeout -k 10 200 python -u scratch/bench_shapes.py conv > gpurun_out/ab_v3.log 2>&1 || exit 1
echo V2; cut -c1-200 gpurun_out/ab_v2.log | cut -d' ' -f1-12; echo V3; cut -d' ' -f1-12 gpurun_out/ab_v3.log

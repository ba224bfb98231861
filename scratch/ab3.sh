#!/bin/bash
set -o pipefail
timeout -k 10 300 python -u -m pytest tests/test_igemm_gpu.py tests/test_i3d_gpu.py -x -q --timeout 120 --timeout-method thread > gpurun_out/ab_pytest.log 2>&1 || { tail -30 gpurun_out/ab_pytest.log; exit 1; }
tail -2 gpurun_out/ab_pytest.log
TRACE_NO_TCONV=1 timeout -k 10 200 python -u scratch/bench_t.py || exit 1
timeout -k 10 200 python -u scratch/bench_t.py

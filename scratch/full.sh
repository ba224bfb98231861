#!/bin/bash
set -o pipefail
timeout -k 10 420 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/pytest_gpu.log 2>&1 || { tail -30 gpurun_out/pytest_gpu.log; exit 1; }
tail -2 gpurun_out/pytest_gpu.log
timeout -k 10 240 python -u bench.py --no-cpu-baseline > gpurun_out/bench.log 2>&1 || { tail -20 gpurun_out/bench.log; exit 1; }
tail -1 gpurun_out/bench.log | cut -c1-200
tail -1 gpurun_out/bench.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline'])"

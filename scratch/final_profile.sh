#!/bin/bash
# Round artefacts: PMC traffic passes, then the default bench command under rocprofv3 --kernel-trace --stats.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o p -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-trace > gpurun_out/pmc_fetch.log 2>&1 || exit 1
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o p -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-trace > gpurun_out/pmc_write.log 2>&1 || exit 1
python tools/pmc_traffic.py gpurun_out/pmc_fetch/p_counter_collection.csv gpurun_out/pmc_write/p_counter_collection.csv gpurun_out/traffic.json || exit 1
cp gpurun_out/traffic.json profiles/r01_igemm_traffic.json
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_final -o bench -- python3 bench.py > gpurun_out/bench_final.log 2>&1 || { tail -20 gpurun_out/bench_final.log; exit 1; }
grep '"metric"' gpurun_out/bench_final.log > gpurun_out/bench_final.json
cut -c1-300 gpurun_out/bench_final.json
timeout -k 10 240 python -u bench.py > gpurun_out/bench_plain.log 2>&1 || exit 1
grep '"metric"' gpurun_out/bench_plain.log | cut -c1-200

#!/bin/bash
# Collects the rocprofv3 evidence of one round on the GPU box.  Usage (from the repo root):
#   bash tools/profile_round.sh <tag>
# Writes gpurun_out/prof_<tag>/{stats,pmc_fetch,pmc_write}/...; tools/make_traffic.py turns the
# counter files into profiles/traffic.json.  Each profiler pass is its own short run.
set -e -o pipefail
TAG=${1:-r01}
ROOT=$PWD
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export PYTHONPATH=$ROOT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 280 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python3 $ROOT/bench.py --steps 10 --warmup 3 > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- python3 $ROOT/tools/stage_bench.py 256 2 > $OUT/pmc_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- python3 $ROOT/tools/stage_bench.py 256 2 > $OUT/pmc_write.log 2>&1
ls -R $OUT | head -40

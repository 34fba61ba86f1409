#!/bin/bash
# Collects the rocprofv3 evidence of one round on the GPU box.  Usage (from the repo root):
#   bash tools/profile_round.sh <tag>
# Writes gpurun_out/prof_<tag>/... : kernel trace + stats of the default bench command and of the headline
# workload alone, the counter passes (each its own short run) and their per-launch summaries;
# tools/make_traffic.py / tools/make_issue.py turn the counter files into profiles/traffic.json / issue.json.
set -e -o pipefail
TAG=${1:-r03}
ROOT=$PWD
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export PYTHONPATH=$ROOT
cd /tmp && export TMPDIR=/tmp
# the default command (headline workload + legs of the other configs + CPU baseline)
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python3 $ROOT/bench.py > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
# the headline workload alone: per-kernel averages that are not mixed with the other configs' launches
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_headline -o bench -- python3 $ROOT/bench.py --no-extras --no-cpu-baseline > $OUT/bench_headline_under_rocprof.json 2> $OUT/bench_headline_under_rocprof.err
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- python3 $ROOT/tools/stage_bench.py 256 2 > $OUT/pmc_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- python3 $ROOT/tools/stage_bench.py 256 2 > $OUT/pmc_write.log 2>&1
# issue / wait counters: at most 8 SQ and 4 TCP counters per pass
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_sq1 -o pmc -- python3 $ROOT/tools/stage_bench.py 256 2 > $OUT/pmc_sq1.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_sq2 -o pmc -- python3 $ROOT/tools/stage_bench.py 256 2 > $OUT/pmc_sq2.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum --output-format csv -d $OUT/pmc_tcp -o pmc -- python3 $ROOT/tools/stage_bench.py 256 2 > $OUT/pmc_tcp.log 2>&1
# the per-frame steps after extraction (undistortion, grid search, stereo searches) and the 4K configuration
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_frame -o fb -- python3 $ROOT/tools/frame_bench.py 128 30 > $OUT/frame_bench_under_rocprof.txt 2> $OUT/frame_bench_under_rocprof.err
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_4k -o bench -- python3 $ROOT/bench.py --config 4k-map --no-cpu-baseline --steps 10 > $OUT/bench_4k_under_rocprof.json 2> $OUT/bench_4k_under_rocprof.err
# the brute-force matcher alone: matrix pipe and vector issue
timeout -s KILL 90 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/pmc_match1 -o pmc -- python3 $ROOT/tools/match_bench.py 3 > $OUT/pmc_match1.log 2>&1
timeout -s KILL 90 rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_match2 -o pmc -- python3 $ROOT/tools/match_bench.py 3 > $OUT/pmc_match2.log 2>&1
# the FP4 sweep on one shape (8 x 8000 queries against one 100000-point map): counters of one kernel on one grid
timeout -s KILL 90 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/pmc_fp4a -o pmc -- python3 $ROOT/tools/match_one_shape.py > $OUT/pmc_fp4a.log 2>&1
timeout -s KILL 90 rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_fp4b -o pmc -- python3 $ROOT/tools/match_one_shape.py > $OUT/pmc_fp4b.log 2>&1
timeout -s KILL 90 rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_fp4c -o pmc -- python3 $ROOT/tools/match_one_shape.py > $OUT/pmc_fp4c.log 2>&1
cd $ROOT
python3 tools/pmc_summary.py $(find $OUT/pmc_match1 $OUT/pmc_match2 -name '*counter_collection.csv') > $OUT/pmc_match_per_launch.txt
python3 tools/pmc_summary.py $(find $OUT/pmc_fp4a $OUT/pmc_fp4b $OUT/pmc_fp4c -name '*counter_collection.csv') > $OUT/pmc_match_fp4_8x8000x100000.txt
python3 tools/pmc_summary.py $(find $OUT/pmc_sq1 $OUT/pmc_sq2 $OUT/pmc_tcp -name '*counter_collection.csv') > $OUT/pmc_sq_tcp_per_launch.txt
python3 tools/pmc_summary.py $(find $OUT/pmc_fetch $OUT/pmc_write -name '*counter_collection.csv') > $OUT/pmc_fetch_write_per_launch.txt
python3 tools/make_traffic.py $(find $OUT/pmc_fetch -name '*counter_collection.csv') $(find $OUT/pmc_write -name '*counter_collection.csv') $OUT/traffic.json > /dev/null
python3 tools/make_issue.py $(find $OUT/pmc_sq1 -name '*counter_collection.csv') $OUT/issue.json > /dev/null
python3 tools/make_match_issue.py $(find $OUT/pmc_fp4a $OUT/pmc_fp4b -name '*counter_collection.csv') $OUT/match_issue.json > /dev/null
python3 tools/trace_filter.py $(find $OUT/stats -name '*kernel_trace.csv') k_fast_blur > $OUT/fast_blur_headline_shape_in_default_run.txt
ls -R $OUT | head -80

#!/bin/bash
# Copies the summaries of gpurun_out/prof_<tag> (tools/profile_round.sh) into profiles/ under the round's name.
# Usage: bash tools/collect_profiles.sh <tag>
set -e
TAG=${1:-r03}
P=gpurun_out/prof_$TAG
cp $P/stats/bench_kernel_stats.csv profiles/${TAG}_bench_kernel_stats.csv
cp $P/stats_headline/bench_kernel_stats.csv profiles/${TAG}_bench_headline_only_kernel_stats.csv
cp $P/bench_under_rocprof.json profiles/${TAG}_bench_under_rocprof.json
cp $P/bench_headline_under_rocprof.json profiles/${TAG}_bench_headline_only_under_rocprof.json
cp $P/pmc_sq_tcp_per_launch.txt profiles/${TAG}_pmc_sq_tcp_per_launch.txt
cp $P/pmc_fetch_write_per_launch.txt profiles/${TAG}_pmc_fetch_write_per_launch.txt
python3 tools/trace_filter.py $P/stats/bench_kernel_trace.csv k_fast_blur > profiles/${TAG}_fast_blur_by_grid_default_run.txt
python3 tools/trace_filter.py $P/stats_headline/bench_kernel_trace.csv k_fast_blur > profiles/${TAG}_fast_blur_by_grid_headline_only_run.txt
cp $P/stats_frame/fb_kernel_stats.csv profiles/${TAG}_frame_steps_kernel_stats.csv
cp $P/frame_bench_under_rocprof.txt profiles/${TAG}_frame_steps_under_rocprof.txt
cp $P/stats_4k/bench_kernel_stats.csv profiles/${TAG}_bench_4k_map_kernel_stats.csv
cp $P/bench_4k_under_rocprof.json profiles/${TAG}_bench_4k_map_under_rocprof.json
cp $P/pmc_match_per_launch.txt profiles/${TAG}_pmc_match_per_launch.txt
cp $P/pmc_match_fp4_8x8000x100000.txt profiles/${TAG}_pmc_match_fp4_8x8000x100000.txt
cp $P/traffic.json profiles/traffic.json
cp $P/issue.json profiles/issue.json
cp $P/match_issue.json profiles/match_issue.json
ls -la profiles

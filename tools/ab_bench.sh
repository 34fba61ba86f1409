#!/bin/bash
# A/B of two builds of liborbx.so in one GPU session (run-to-run noise is below 1 %, box-to-box
# differences are not).  Usage: bash tools/ab_bench.sh <other.so> [rounds]
# Alternates the in-tree library ("new") with <other.so> ("other") and prints the stage times.
set -e
export PYTHONPATH=$PWD
OTHER=$1
N=${2:-3}
cp orb_slam_amd/lib/liborbx.so /tmp/ab_new.so
trap 'cp /tmp/ab_new.so orb_slam_amd/lib/liborbx.so' EXIT
for i in $(seq $N); do
  cp /tmp/ab_new.so orb_slam_amd/lib/liborbx.so; echo new; timeout -k 10 100 python3 tools/stage_bench.py 256 30 | sed -n 1,2p
  cp $OTHER orb_slam_amd/lib/liborbx.so; echo other; timeout -k 10 100 python3 tools/stage_bench.py 256 30 | sed -n 1,2p
done

#!/bin/bash
set -e
export PYTHONPATH=$PWD
echo new; timeout -k 10 100 python3 tools/match_bench.py 30
for e in "$@"; do
  echo $e; ORBX_LIBRARY=$PWD/build/liborbx_$e.so timeout -k 10 100 python3 tools/match_bench.py 30
done
echo new; timeout -k 10 100 python3 tools/match_bench.py 30

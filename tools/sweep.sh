#!/bin/bash
# scratch: sweep of fp32 tree build parameters
for ci in 1.2 2 3 5; do for leaf in 1 2; do
  echo "== ci $ci leaf $leaf"
  PT_TUNE_CI=$ci PT_TUNE_LEAF=$leaf timeout -k 10 200 python3 tools/quick_bench.py car atrium || exit 1
done; done

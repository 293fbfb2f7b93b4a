#!/bin/bash
# usage: diff2.sh lib...  -- runs diff_pipe.py with RO_TEST_LIB pointing at each lib
for l in "$@"; do echo "== $l"; RO_TEST_LIB=$l timeout -k 10 200 python tools/r2/diff_pipe.py 2048 2>/dev/null; done

#!/bin/bash
set -e
O=gpurun_out/r3_12; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
timeout -k 10 120 python __graft_entry__.py smoke 2>&1 | tail -1

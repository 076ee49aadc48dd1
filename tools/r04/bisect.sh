#!/bin/bash
cd $GRAFT_REPO_ROOT
for f in test_gpu_edge_cases test_gpu_configs; do
  timeout -k 10 600 python -m pytest tests/$f.py tests/test_gpu_efa.py -m gpu -q -k "asynchronous or not efa" 2>&1 | tail -4 | tr '\n' ' ' | cut -c1-900; echo " <- $f"
done

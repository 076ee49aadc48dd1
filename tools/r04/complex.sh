#!/bin/bash
# round 4: where complex hoppings (T = ComplexF64) stand against real ones; smoke() at the split library
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python tools/complex_scan.py 16 2>&1 | tee gpurun_out/r04_complex_scan.txt
python tools/complex_scan.py 1 2>&1 | tee -a gpurun_out/r04_complex_scan.txt

#!/bin/bash
# A/B of prebuilt library variants (variants/lib_<tag>.so, built here with a modified kernel): each is copied over the in-tree library
# and the same commands run against it.  usage: tools/ab_libs.sh <tag> [<tag> ...]
L=smoqyelphqmc.jl_amd/csrc/libsmoqy_hip.so
cp $L /tmp/lib_keep.so
for t in "$@"; do
  cp variants/lib_$t.so $L && touch $L
  a=$(python tools/matvec_only.py 16 200 | awk '{print $3}')
  b=$(python tools/matvec_only.py 128 50 | awk '{print $3}')
  v1=$(python bench.py --timed-only --steps 6 --no-mtm-sampling | python -c "import json,sys; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'],1))")
  v2=$(python bench.py --timed-only --steps 6 --no-mtm-sampling | python -c "import json,sys; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'],1))")
  echo "variant $t: MtM b16 $a us, b128 $b us, bench $v1 $v2 sweeps/s" | tee -a gpurun_out/ab_libs.txt
done
cp /tmp/lib_keep.so $L

#!/bin/bash
# one stream, 8 / 16 walkers, bench.py's sweep: the round-3 library (variants/lib_r03.so) against the current one, alternating
cd $GRAFT_REPO_ROOT
L=smoqyelphqmc.jl_amd/csrc/libsmoqy_hip.so
cp $L /tmp/lib_keep.so
for rep in 1 2; do
  for t in r03 cur; do
    if [ $t = r03 ]; then cp variants/lib_r03.so $L; else cp /tmp/lib_keep.so $L; fi
    touch $L smoqyelphqmc.jl_amd/csrc/libsmoqy_member.so
    for nw in 16 8; do
      r=$(SMOQY_AB_OLD_LIBRARY=1 SMOQY_EFA=1 SMOQY_PREFETCH=1 SMOQY_SPLIT=0 python tools/one_stream.py $nw 2>&1 | tail -1)
      echo "$t nw=$nw: $r"
    done
  done
done
cp /tmp/lib_keep.so $L

#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/profile_bench.sh r02 && echo prof ok
bash tools/profile_roofline.sh r02 && echo roof ok

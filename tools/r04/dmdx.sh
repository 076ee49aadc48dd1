#!/bin/bash
# round 4: dmdx_kernel with |u'> and |v'> sharing their first colour passes: force / trajectory tests, then its solo duration
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_force.py tests/test_gpu_complex_T.py tests/test_gpu_efa.py tests/test_gpu_sweep_parity.py tests/test_gpu_team.py tests/test_gpu_phonon_fields.py tests/test_gpu_irregular.py tests/test_gpu_large_lattice.py -m gpu -x -q 2>&1 | tail -4
bash tools/solo_profile.sh r04_hc16_dmdx 16 && grep "dmdx\|sweep ms" gpurun_out/solo_r04_hc16_dmdx.txt | cut -c1-170
SMOQY_EFA=1 SMOQY_PREFETCH=1 python tools/one_stream.py 1 | tail -1; SMOQY_DMDX_FAST=0 SMOQY_EFA=1 SMOQY_PREFETCH=1 python tools/one_stream.py 1 | tail -1

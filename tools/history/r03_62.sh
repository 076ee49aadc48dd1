#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for nw in 32 64; do echo "auto nw=$nw: $(timeout -k 10 120 python tools/history/one_stream_iters.py $nw 2>&1 | tail -1)"; echo "off  nw=$nw: $(SMOQY_FDM_OWNSTREAM=0 timeout -k 10 120 python tools/history/one_stream_iters.py $nw 2>&1 | tail -1)"; done
done
timeout -k 10 200 python bench.py --no-cpu-baseline --no-proc-scan --timed-only --steps 0 --warmup 0 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print([(b['batch'],round(b['us'],1)) for b in r.get('batch_scan',[])]); print(r.get('hbm_resident_point'))"

#!/bin/bash
# Cache-hierarchy counters of one kernel family, several rocprofv3 --pmc passes (counters only, --kernel-trace for the names).
# Each pass holds at most four counters of one block (more is refused by the hardware and the aborting profiler then hangs: hence the timeout).
# PMC_SET=sq selects the wave-scheduler counters instead of the cache ones.
# usage: tools/pmc_explore.sh <tag> <kernel-substring> -- <python script and args>     -> gpurun_out/pmc_explore_<tag>.txt
tag=$1; pat=$2; shift 3
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_explore_$tag.txt
: > $out
groups=(
 "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum"
 "TCC_READ_sum TCC_WRITE_sum TCC_TAG_STALL_sum"
 "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_ACCESSES_sum"
 "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum"
 "TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum TCC_BUSY_avr"
 "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum"
 "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_UTCL1_TRANSLATION_MISS_sum"
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS"
 "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS GRBM_GUI_ACTIVE"
 "TA_BUSY_avr TCP_TCP_TA_DATA_STALL_CYCLES_sum"
)
if [ "$PMC_SET" = "sq" ]; then groups=(
 "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
 "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC"
 "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"
 "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_DATA_FIFO_FULL"
 "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_CMD_FIFO_FULL"
 "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_FLAT SQ_BUSY_CU_CYCLES SQ_INSTS_VALU"
); fi
for g in "${groups[@]}"; do
  d=$GRAFT_REPO_ROOT/gpurun_out/pmcx
  rm -rf $d
  timeout -k 10 100 rocprofv3 --pmc $g --kernel-trace --output-format csv -d $d -o run -- python3 "$@" > gpurun_out/pmcx.log 2>&1 || { echo "pass failed: $g" >> $out; tail -3 gpurun_out/pmcx.log >> $out; continue; }
  python3 - "$pat" >> $out <<'PY'
import csv, glob, sys, collections
pat = sys.argv[1]
f = glob.glob("gpurun_out/pmcx/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if pat in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(f"{k:40s} launches {len(v):4d}  mean {sum(v)/len(v):16.1f}  min {min(v):16.1f}  max {max(v):16.1f}")
PY
done
rm -rf gpurun_out/pmcx
cat $out

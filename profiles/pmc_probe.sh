#!/bin/bash
# Diagnostic (GPU box): SQ / TCP / TCC counters per conv kernel variant for one bench step.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_probe
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/s$i -o p -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/s$i.log 2>&1 || { echo "set $i failed"; tail -3 $OUT/s$i.log; }
done
python3 - <<PY
import csv, collections, glob
out=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set)
for f in glob.glob("$OUT/s*/**/p_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "conv_igemm" not in k: continue
        k=k[k.find("<"):k.find(">")+1]
        out[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[(k,f)].add(r["Dispatch_Id"])
for k,v in sorted(out.items()):
    gui=v.get("GRBM_GUI_ACTIVE",0); wc=v.get("SQ_WAVE_CYCLES",1)
    print(k)
    print("   mfma_busy=%.3f wait_inst=%.3f wait_lds=%.3f valu_active=%.3f | valu=%.2e lds=%.2e vmem_rd=%.2e ldsconf/idx=%.3f | tcp_acc=%.2e tcp->tcc=%.2e tcc_hit=%.2e tcc_miss=%.2e ea_rd=%.2e"%(
      v.get("SQ_VALU_MFMA_BUSY_CYCLES",0)/(gui/8*1024) if gui else 0, v.get("SQ_WAIT_INST_ANY",0)/wc, v.get("SQ_WAIT_INST_LDS",0)/wc, v.get("SQ_ACTIVE_INST_VALU",0)/wc,
      v.get("SQ_INSTS_VALU",0), v.get("SQ_INSTS_LDS",0), v.get("SQ_INSTS_VMEM_RD",0), v.get("SQ_LDS_BANK_CONFLICT",0)/max(v.get("SQ_LDS_IDX_ACTIVE",1),1),
      v.get("TCP_TOTAL_CACHE_ACCESSES_sum",0), v.get("TCP_TCC_READ_REQ_sum",0), v.get("TCC_HIT_sum",0), v.get("TCC_MISS_sum",0), v.get("TCC_EA0_RDREQ_sum",0)))
PY

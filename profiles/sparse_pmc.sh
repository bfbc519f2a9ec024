#!/bin/bash
# Diagnostic (GPU box): SQ counters of the sparse consumer vs the dense kernel on the expanded tensor (profiles/sparse_ab.py).
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/sparse_pmc
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/s$i -o p -- python3 $ROOT/profiles/sparse_ab.py ${1:-block4_conv3} > $OUT/s$i.log 2>&1 || { echo "set $i failed"; tail -3 $OUT/s$i.log; }
done
python3 - <<PY
import csv, collections, glob
out=collections.defaultdict(lambda: collections.defaultdict(float)); nd=collections.defaultdict(set)
for f in glob.glob("$OUT/s*/**/p_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "conv_igemm" not in k and "conv_sparse_kernel" not in k: continue
        k="sparse" if "sparse" in k else "dense"+k[k.find("<"):k.find(">")+1]
        out[k][r["Counter_Name"]]+=float(r["Counter_Value"]); nd[(k,r["Counter_Name"])].add(r["Dispatch_Id"])
for k,v in sorted(out.items()):
    gui=v.get("GRBM_GUI_ACTIVE",0); wc=v.get("SQ_WAVE_CYCLES",1)
    n=len(nd[(k,"SQ_INSTS_VALU")]) or 1
    print(k, "dispatches", n)
    print("   mfma_busy=%.3f wait_inst=%.3f wait_lds=%.3f valu_active=%.3f | per launch: valu=%.3e lds=%.3e vmem_rd=%.3e ldsconf/idx=%.3f | tcp_acc=%.2e tcp->tcc=%.2e tcc_hit=%.2e tcc_miss=%.2e ea_rd=%.2e"%(
      v.get("SQ_VALU_MFMA_BUSY_CYCLES",0)/(gui/8*1024) if gui else 0, v.get("SQ_WAIT_INST_ANY",0)/wc, v.get("SQ_WAIT_INST_LDS",0)/wc, v.get("SQ_ACTIVE_INST_VALU",0)/wc,
      v.get("SQ_INSTS_VALU",0)/n, v.get("SQ_INSTS_LDS",0)/n, v.get("SQ_INSTS_VMEM_RD",0)/n, v.get("SQ_LDS_BANK_CONFLICT",0)/max(v.get("SQ_LDS_IDX_ACTIVE",1),1),
      v.get("TCP_TOTAL_CACHE_ACCESSES_sum",0)/n, v.get("TCP_TCC_READ_REQ_sum",0)/n, v.get("TCC_HIT_sum",0)/n, v.get("TCC_MISS_sum",0)/n, v.get("TCC_EA0_RDREQ_sum",0)/n))
PY

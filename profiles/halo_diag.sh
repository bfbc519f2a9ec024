#!/bin/bash
# Diagnostic (GPU box): SQ counters of the conv kernels with and without the halo-resident variant.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/halo_diag
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for H in 0 1; do
  export LRP_CONV_HALO=$H
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/h$H -o p -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/h$H.log 2>&1 || { tail -5 $OUT/h$H.log; exit 1; }
done
python3 - <<PY
import csv, collections
for H in (0,1):
    out=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set)
    import glob
    f=glob.glob("$OUT/h%d/**/p_counter_collection.csv"%H, recursive=True)[0]
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "conv_igemm" not in k: continue
        k=k[k.find("<"):k.find(">")+1]
        out[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    print("HALO env", H)
    for k,v in sorted(out.items()):
        gui=v["GRBM_GUI_ACTIVE"]
        print("  %-40s n=%3d gui=%.3e mfma_busy=%.3f ldsconf/idx=%.3f valu=%.3e lds=%.3e wavecyc=%.3e"%(k,len(n[k]),gui,v["SQ_VALU_MFMA_BUSY_CYCLES"]/(gui/8*1024) if gui else 0, v["SQ_LDS_BANK_CONFLICT"]/max(v["SQ_LDS_IDX_ACTIVE"],1), v["SQ_INSTS_VALU"], v["SQ_INSTS_LDS"], v["SQ_WAVE_CYCLES"]))
PY

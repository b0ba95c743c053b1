#!/bin/bash
# PMC passes over the fm-lf/no_ftab series for the k_fm_sweep kernels (what are the waves doing?)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
ARGS="--lean --steps 3 --warmup 1 --mode locus-table --tune 3 --ftab -1"
i=0
for PMC in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/swp$i
  rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $R/gpurun_out/swp$i -- python3 $R/bench.py $ARGS > $R/gpurun_out/swp$i.log 2>&1
done
python3 - > $R/gpurun_out/sweep_pmc.txt <<PY
import csv, glob, collections
rows=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(set)
for f in glob.glob('$R/gpurun_out/swp*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].replace("(anonymous namespace)::","").split("(")[0]
        if 'k_fm_sweep' not in k: continue
        rows[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(k,r["Counter_Name"])].add(r["Dispatch_Id"])
for k in sorted(rows):
    print(k)
    for c,v in sorted(rows[k].items()):
        print('   %-24s %16.0f per dispatch'%(c, v/len(cnt[(k,c)])))
PY
cat $R/gpurun_out/sweep_pmc.txt
find $R/gpurun_out/swp* -name "*counter_collection.csv" -delete; find $R/gpurun_out/swp* -name "*kernel_trace.csv" -delete; find $R/gpurun_out/swp* -name "*.db" -delete

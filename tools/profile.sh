#!/bin/bash
# Collects rocprofv3 kernel stats and PMC passes for bench.py on the GPU box.
# usage (on the box, from the repo root): bash tools/profile.sh <tag> [bench args...]
set -u
TAG=${1:-x}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--lean --steps 3 --warmup 1 $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py $ARGS > $OUT/stats.log 2>&1
i=0
for PMC in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_ATOMIC_sum" "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $OUT/pmc$i -- python3 $R/bench.py $ARGS > $OUT/pmc$i.log 2>&1
done
cd $R
python3 - <<PY
import csv, glob, collections, os
out="$OUT"
def short(n):
    n=n.replace("(anonymous namespace)::","")
    return n.split("(")[0]
rows=[]
for d in sorted(glob.glob(out+"/pmc*/")):
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            k=short(r["Kernel_Name"]); agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
        for k in agg:
            for c,v in agg[k].items():
                rows.append((k,c,v/len(cnt[k]),len(cnt[k])))
with open(out+"/pmc_summary.csv","w") as fh:
    fh.write("kernel,counter,avg_per_dispatch,dispatches\n")
    for r in sorted(rows): fh.write("%s,%s,%.1f,%d\n"%r)
print(open(out+"/pmc_summary.csv").read())
for f in glob.glob(out+"/stats/**/*kernel_stats.csv", recursive=True):
    print(open(f).read())
# traffic per launch (bytes): FETCH_SIZE / WRITE_SIZE are KiB; random 64-B sector reads are counted
# exactly, wide coalesced streams at half (profiles/r01_fetch_size_calibration.txt)
import json
# per STEP (one call of the pipeline = one k_seed_pack dispatch), not per dispatch: the traverser is launched
# twice in a step -- the loci, then the few items of its spill queue -- and an average over the dispatches halves it
calls={}
# (a step = one dispatch of the seeding kernel: k_kmer_step since round 5 -- also in traverse mode -- or k_seed_pack; whichever
# ran more often is the one of the timed steps, the other one belongs to a single set-up call)
for k,c,v,n in rows:
    if k.startswith("void k_seed_pack") or k.startswith("k_seed_pack") or k.startswith("void k_kmer_step") or k.startswith("k_kmer_step"):
        calls[c]=max(calls.get(c,0),n)
tr={}
for k,c,v,n in rows:
    if c in ("FETCH_SIZE","WRITE_SIZE"):
        per=max(1.0, round(n/calls[c])) if calls.get(c) else 1.0
        tr.setdefault(k,{})[c.lower()+"_bytes"]=v*1024*per
        if per!=1.0: tr[k]["dispatches_per_step"]=per
mode="kmer-table"
a="$ARGS".split()
if "--mode" in a: mode=a[a.index("--mode")+1]
series=""
if "--series" in a: series=a[a.index("--series")+1]
json.dump({"args":"$ARGS","mode":mode,"series":series,"unit":"bytes per step (FETCH_SIZE / WRITE_SIZE x 1024, summed over the dispatches of a step)","per_launch":tr}, open(out+"/traffic.json","w"), indent=1)
PY
# keep the summaries, drop the raw traces (gpurun merges at most 64 MiB back)
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*.db" -delete
du -sh $OUT | cut -f1

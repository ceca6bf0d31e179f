#!/bin/bash
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp
export TMPDIR=/tmp
rocm-smi --showclocks --showpower --showperflevel 2>&1 | head -40 > $O/rocm_smi.txt
for abl in 0 1 2 3 4; do
  timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/pmc_abl/abl$abl -o pmc -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --option ablate=$abl > $O/pmc_abl$abl.log 2>&1
  python - <<PY
import csv, collections
acc=collections.defaultdict(list); dur=[]
for r in csv.DictReader(open("$O/pmc_abl/abl$abl/pmc_counter_collection.csv")):
    if "syrk" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Counter_Name"]=="GRBM_GUI_ACTIVE": dur.append((float(r["End_Timestamp"])-float(r["Start_Timestamp"]))/1e3)
m={k:sum(v)/len(v) for k,v in acc.items()}
d=sum(dur)/len(dur)
cyc=m["GRBM_GUI_ACTIVE"]/8
print("ablate=$abl  dur %.1f us  cycles/XCD %.0f  clock %.3f GHz  mfma_util %.3f  wave_alive %.3f  wait_any %.3f  wait_inst %.3f"%(d,cyc,cyc/d/1e3,m["SQ_VALU_MFMA_BUSY_CYCLES"]/1024/cyc, m["SQ_WAVE_CYCLES"]*4/4040/cyc, m["SQ_WAIT_ANY"]/m["SQ_WAVE_CYCLES"], m["SQ_WAIT_INST_ANY"]/m["SQ_WAVE_CYCLES"]))
PY
done
cat $O/rocm_smi.txt | head -30
find $O -name "*.csv" -size +8M -delete

cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out
export PHMRF_TRACE_PERT=0.05
timeout -k 10 240 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAVES --output-format csv -d $O/ws_pm -- python3 tools/trace.py 20 4980 1000 > /dev/null 2> $O/r3_clk.err
python3 - <<'PY'
import csv,glob,collections
rows=[]
for f in glob.glob("gpurun_out/ws_pm/**/*counter_collection.csv", recursive=True):
    rows+=list(csv.DictReader(open(f)))
kt=[]
for f in glob.glob("gpurun_out/ws_pm/**/*kernel_trace.csv", recursive=True):
    kt+=list(csv.DictReader(open(f)))
dur={int(r["Dispatch_Id"]):(int(r["End_Timestamp"])-int(r["Start_Timestamp"])) for r in kt}
per=collections.defaultdict(dict)
for r in rows:
    if "strip_cols" in r["Kernel_Name"] or "emission" in r["Kernel_Name"] or "energy_grid" in r["Kernel_Name"]:
        per[(int(r["Dispatch_Id"]), r["Kernel_Name"][:60])][r["Counter_Name"]]=float(r["Counter_Value"])
for (d,k),v in sorted(per.items())[-14:]:
    ns=dur.get(d,0)
    print(d,k[-40:],ns/1e3,"us", {a:int(b) for a,b in v.items()}, "clk_GHz=%.2f"%(v.get("GRBM_GUI_ACTIVE",0)/8/max(ns,1)))
PY
rm -rf $O/ws_pm

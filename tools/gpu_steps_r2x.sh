out=$1
R=$GRAFT_REPO_ROOT
cd /tmp
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $R/$out/pmc1 -o p -- python3 $R/tools/panel_pmc_probe.py; echo "pmc1 rc=$?"
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $R/$out/pmc2 -o p -- python3 $R/tools/panel_pmc_probe.py; echo "pmc2 rc=$?"
cd $R
python - <<PY
import csv,glob,collections
for d in ("$out/pmc1","$out/pmc2"):
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        acc=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            if "panel_max" not in r["Kernel_Name"]: continue
            acc[r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for gs,c in acc.items():
            print(d, "grid", gs, {k: round(sum(v)/len(v)) for k,v in c.items()})
PY
rm -f $out/pmc1/*kernel_trace* $out/pmc2/*kernel_trace*

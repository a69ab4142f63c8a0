R=$PWD
mkdir -p $R/gpurun_out/r2e
cd /tmp && export TMPDIR=/tmp
for cfg in "fluid_m1:--scene fluid --spp 2" "fluid_m0:--scene fluid --spp 2 --opt merged=0" "show_m1:--scene showcase --opt steal=0" "show_m0:--scene showcase --opt merged=0"; do
  tag=${cfg%%:*}; args=${cfg#*:}; mkdir -p $R/gpurun_out/r2e/$tag
  for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_BRANCH"; do
    N=$(echo $C | tr ' ' '_' | cut -c1-24)
    rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/r2e/$tag/$N -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline $args > $R/gpurun_out/r2e/$tag/$N.log 2>&1 || echo "pass $tag $N failed"
  done
done
cd $R && python3 - <<'PY'
import csv, glob, collections
for tag in ("fluid_m1","fluid_m0","show_m1","show_m0"):
    agg = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/r2e/{tag}/*/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "path_trace" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(tag, {k: round(sum(v)/len(v)/1e6,1) for k,v in sorted(agg.items())})
PY

#!/bin/bash
# usage: profiles/mix_pass.sh <tag> <round> <bench args...>   (GPU box, repo root)
# Instruction mix of the path-trace kernel a bench run settles on: three rocprofv3 --pmc passes of 8 counters (never combined
# with a trace domain), every frame ONE launch ordered behind the stream (--no-pipeline) so that a dispatch is a frame.
# Writes profiles/<round>_<tag>_instruction_mix.json (mean per launch over the launches after the first five).
TAG=$1; RND=$2; shift; shift
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/mix_$TAG
i=0
for C in "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_THREAD_CYCLES_VALU" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_WAVE_CYCLES SQ_IFETCH" \
         "SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/mix_$TAG/p$i -- python3 $R/bench.py --steps 8 --warmup 8 --no-pipeline --no-ramp --no-cpu-baseline --no-configs3 "$@" > $R/gpurun_out/mix_$TAG/p$i.log 2>&1 || echo "mix pass $i failed"
done
cd $R
python3 - "$TAG" "$RND" "$*" <<'PY'
import collections, csv, glob, json, sys
tag, rnd, args = sys.argv[1], sys.argv[2], sys.argv[3]
agg, kernel = collections.defaultdict(list), None
for f in glob.glob(f"gpurun_out/mix_{tag}/p*/*/*_counter_collection.csv"):
    rows = [r for r in csv.DictReader(open(f)) if "path_trace" in r["Kernel_Name"]]
    if not rows:
        continue
    kernel = rows[-1]["Kernel_Name"]
    ids = sorted({int(r["Dispatch_Id"]) for r in rows if r["Kernel_Name"] == kernel})
    skip = set(ids[:5]) if len(ids) > 8 else set()
    for r in rows:
        if r["Kernel_Name"] == kernel and int(r["Dispatch_Id"]) not in skip:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
c = {k: sum(v) / len(v) for k, v in sorted(agg.items())}
d = {}
if c.get("SQ_INSTS_VALU"):
    valu = c["SQ_INSTS_VALU"]
    d["active_lanes_per_valu_instruction"] = c.get("SQ_THREAD_CYCLES_VALU", 0) / max(1.0, c.get("SQ_ACTIVE_INST_VALU", valu)) / 1.0
    arith = sum(c.get(k, 0) for k in ("SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_TRANS_F32",
                                      "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_CVT"))
    d["valu_arithmetic_share (fma+mul+add+int32+trans+cvt)"] = arith / valu
    d["branches_per_valu"] = c.get("SQ_INSTS_BRANCH", 0) / valu
    d["salu_cycles_per_valu"] = c.get("SQ_INST_CYCLES_SALU", 0) / valu
    d["useful_fp32_flop_per_launch (fma x2 + mul + add, x active lanes)"] = (
        (2 * c.get("SQ_INSTS_VALU_FMA_F32", 0) + c.get("SQ_INSTS_VALU_MUL_F32", 0) + c.get("SQ_INSTS_VALU_ADD_F32", 0)) * d["active_lanes_per_valu_instruction"])
json.dump({"what": f"rocprofv3 --pmc, bench.py {args} --no-pipeline; kernel {kernel}; mean per launch, first five launches dropped; three passes of 8 counters "
                   "(profiles/mix_pass.sh)", "kernel": kernel, "counters": c, "derived": d},
          open(f"profiles/{rnd}_{tag}_instruction_mix.json", "w"), indent=1)
print("instruction mix:", kernel, {k: round(v, 4) for k, v in d.items()})
PY

#!/bin/bash
# PMC passes of every bench configuration in one call (separate rocprofv3 runs per counter group, kernel trace only):
#   c2 bf16: all six groups (SQ busy / MFMA, LDS, FETCH_SIZE, WRITE_SIZE, L2 hit/miss, GRBM)   c2 bf16x3: MFMA + HBM
#   c3 bf16, c4 bf16: HBM (FETCH_SIZE, WRITE_SIZE); c3 on one stream so that the launches of a step keep their order
# bash profiles/scripts/gpu_pmc_all.sh <tag>     -> gpurun_out/pmc_<tag>_{c2_bf16,c2_bf16x3,c3_bf16,c4_bf16}/
cd /root/repo; OUT=/root/repo/gpurun_out; TAG=${1:-pmc}
eval "$(timeout 600 python -m reagent_amd.device_preflight | tee /dev/stderr | grep "^export ")"; timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); assert float((x * 2).sum()) == 2 << 20" || { echo "preflight failed: faulty GPU node, aborting"; exit 97; }
export TMPDIR=/tmp
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA"
G2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM"
G5="TCC_HIT_sum TCC_MISS_sum"; G6="GRBM_GUI_ACTIVE GRBM_COUNT"
run_cfg() {  # config precision groups...
  local cfg=$1 prec=$2; shift 2
  local D=$OUT/pmc_${TAG}_${cfg}_${prec}; mkdir -p $D
  local CMD="python /root/repo/bench.py --config $cfg --precision $prec --steps 3 --warmup 2 --repeats 1 --no-cpu-baseline --no-kernel-profile --no-parity --no-accurate --no-also --sustained-steps 0 --launch eager --no-graph"
  local i=0
  for grp in "$@"; do
    i=$((i+1))
    (cd /tmp && RG_QR_STREAMS=0 timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $D/g$i -o p -- $CMD > $D/g$i.log 2>&1; echo "$cfg $prec group $i ($grp) rc=$?")
  done
  python - <<PY
import csv, glob, collections, re
out = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$D/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("rg::", "")[:44]
        out[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$D/summary.txt", "w") as fh:
    for k, cs in sorted(out.items()):
        if k.startswith("at::") or k.startswith("__amd"): continue
        line = k + " | " + " ".join(f"{c}={sum(v)/len(v):.4g}(n={len(v)})" for c, v in sorted(cs.items()))
        fh.write(line + "\n")
print(open("$D/summary.txt").read()[:3000])
PY
  find $D -name "*kernel_trace.csv" -delete; find $D -name "*agent_info.csv" -delete
}
run_cfg c2 bf16 "$G1" "$G2" "FETCH_SIZE" "WRITE_SIZE" "$G5" "$G6"
run_cfg c2 bf16x3 "$G1" "FETCH_SIZE" "WRITE_SIZE"
run_cfg c3 bf16 "FETCH_SIZE" "WRITE_SIZE"
run_cfg c4 bf16 "FETCH_SIZE" "WRITE_SIZE"

#!/bin/bash
# SQ counters of the C3 step's kernels (what bounds the quantile head): two PMC passes, kernel trace only
cd /root/repo; OUT=/root/repo/gpurun_out; TAG=${1:-c3sq}
eval "$(timeout 600 python -m reagent_amd.device_preflight | tee /dev/stderr | grep "^export ")"; timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); assert float((x * 2).sum()) == 2 << 20" || { echo "preflight failed: faulty GPU node, aborting"; exit 97; }
export TMPDIR=/tmp
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA"
G2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM"
G3="SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAVES SQ_ACTIVE_INST_MISC"
D=$OUT/pmc_${TAG}; mkdir -p $D
CMD="python /root/repo/bench.py --config c3 --precision bf16 --steps 3 --warmup 2 --repeats 1 --no-cpu-baseline --no-kernel-profile --no-parity --no-accurate --no-also --sustained-steps 0 --launch eager --no-graph"
i=0
for grp in "$G1" "$G2" "$G3"; do
  i=$((i+1))
  (cd /tmp && RG_QR_STREAMS=0 RG_QR_WGRAD_STREAMS=0 timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $D/g$i -o p -- $CMD > $D/g$i.log 2>&1; echo "group $i rc=$?")
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
        fh.write(k + " | " + " ".join(f"{c}={sum(v)/len(v):.4g}(n={len(v)})" for c, v in sorted(cs.items())) + "\n")
print(open("$D/summary.txt").read()[:6000])
PY
find $D -name "*kernel_trace.csv" -delete; find $D -name "*agent_info.csv" -delete

#!/bin/bash
# PMC passes (separate runs per counter group, kernel-trace only — never combined with sys/hip traces)
cd /root/repo; OUT=/root/repo/gpurun_out; TAG=${1:-pmc}; mkdir -p $OUT/pmc_$TAG
# preflight: a node whose first device touch faults (seen once: "Memory access fault by GPU" on tensor.to)
# would otherwise burn the whole GPU budget in core dumps and timeouts
eval "$(timeout 600 python -m reagent_amd.device_preflight | tee /dev/stderr | grep "^export ")"; timeout 120 python -c "import torch; x = torch.ones(1 << 20, device='cuda'); assert float((x * 2).sum()) == 2 << 20" || { echo "preflight failed: faulty GPU node, aborting"; exit 97; }
export TMPDIR=/tmp
CMD="python /root/repo/bench.py --config ${PMC_CONFIG:-c2} --precision ${PMC_PREC:-bf16} --steps 3 --warmup 2 --repeats 1 --no-cpu-baseline --no-kernel-profile --no-parity --no-accurate --no-also --sustained-steps 0 --launch eager"
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  (cd /tmp && timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pmc_$TAG/g$i -o p -- $CMD > $OUT/pmc_$TAG/g$i.log 2>&1; echo "group $i rc=$?")
done
python - <<PY
import csv, glob, collections, re
out = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pmc_$TAG/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("rg::", "")[:40]
        out[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$OUT/pmc_$TAG/summary.txt", "w") as fh:
    for k, cs in sorted(out.items()):
        if not any(x in k for x in ("fused", "x3", "wgrad", "gather", "replay", "gemm", "reduce_group", "dqn_head", "update", "qr_", "group_")): continue
        line = k + " | " + " ".join(f"{c}={sum(v)/len(v):.4g}(n={len(v)})" for c, v in sorted(cs.items()))
        print(line); fh.write(line + "\n")
PY
find $OUT/pmc_$TAG -name "*kernel_trace.csv" -delete; find $OUT/pmc_$TAG -name "*agent_info.csv" -delete

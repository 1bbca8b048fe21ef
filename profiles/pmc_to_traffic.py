#!/usr/bin/env python3
"""PMC passes (profiles/scripts/gpu_pmc.sh: separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs of bench.py, kernel trace
only) -> profiles/traffic.json, the per-launch HBM bytes bench.py reports as `roofline.traffic`.

Corrections as the guide prescribes (/opt/skills/guides/MI355X_MICROARCH.md, HBM section): both counters are in KB;
on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced reads, so it is doubled; WRITE_SIZE is taken as is.
The file is stamped with the hash of the kernel sources it was measured on (bench.source_stamp()): bench.py reports
the figure only while that hash matches what is running.

usage: python profiles/pmc_to_traffic.py gpurun_out/pmc_<tag>_<config>_<precision> <config> <precision> ["<where it came from>"]
(profiles/scripts/gpu_pmc_all.sh <tag> runs the passes of c2 bf16 / c2 bf16x3 / c3 bf16 / c4 bf16 in one call)
"""
import collections
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def per_dispatch(pmc_dir, counter):
    out = collections.defaultdict(list)
    for f in sorted(glob.glob(os.path.join(pmc_dir, "g*", "**", "*counter_collection.csv"), recursive=True)):
        rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
        rows.sort(key=lambda r: int(r.get("Dispatch_Id", 0)))
        for r in rows:
            name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("rg::", "").strip()
            out[name].append(float(r["Counter_Value"]))
    return out


def main():
    pmc_dir, config, prec = sys.argv[1], sys.argv[2], sys.argv[3]
    where = sys.argv[4] if len(sys.argv) > 4 else pmc_dir
    import bench

    fetch, write = per_dispatch(pmc_dir, "FETCH_SIZE"), per_dispatch(pmc_dir, "WRITE_SIZE")
    kernels, raw = {}, {}

    def put(key, names, select=lambda i, n: True):
        total_f = total_w = 0.0
        if not names:  # the kernel is not in this run (e.g. the three-launch sampler once the one-launch form serves the store)
            return
        for nm in names:
            fs = [v for i, v in enumerate(fetch.get(nm, [])) if select(i, len(fetch[nm]))]
            wsz = [v for i, v in enumerate(write.get(nm, [])) if select(i, len(write[nm]))]
            if not fs or not wsz:
                return
            total_f += sum(fs) / len(fs)
            total_w += sum(wsz) / len(wsz)
        kernels[f"{config}:{prec}:{key}"] = int(round((2.0 * total_f + total_w) * 1024))
        raw[key] = {"fetch_size_kb_raw": round(total_f, 1), "write_size_kb_raw": round(total_w, 1)}

    fwd = [n for n in fetch if n.startswith("mlp_fwd_fused_kernel") or n.startswith("mlp_fwd_x3_kernel")]
    # (round 4: the backward of a stack with a grouped output layer is its own instantiation, mlp_bwd_grouped_kernel)
    bwd = [n for n in fetch if n.startswith(("mlp_bwd_grouped_kernel", "mlp_bwd_x3_grouped_kernel"))] or \
          [n for n in fetch if n.startswith("mlp_bwd_fused_kernel") or n.startswith("mlp_bwd_x3_kernel")]
    # the weight gradient's second launch: reduce_tail_kernel (split reduce + bias column reduce + loss mean, round 3) or
    # reduce_group_kernel (split reduce alone)
    reduce_k = "reduce_tail_kernel" if "reduce_tail_kernel" in fetch else "reduce_group_kernel"
    first = lambda prefix: [n for n in fetch if n.startswith(prefix)][:1]  # noqa: E731
    if config in ("c2", "c5"):
        # a DQN step launches the forward three times: next state online, next state target (save = 0), state (save = 1)
        put("rg_mlp_forward_fused:save=0", fwd[:1], lambda i, n: i % 3 != 2)
        put("rg_mlp_forward_fused:save=1", fwd[:1], lambda i, n: i % 3 == 2)
        put("rg_mlp_backward_fused", bwd[:1])
        put("rg_mlp_wgrad_fused", ["wgrad_group_kernel", reduce_k])
        put("rg_replay_dqn_batch", first("replay_dqn_batch_kernel"))
        put("rg_mlp_update_fused", first("mlp_update_tiles_kernel"))
        put("rg_dqn_head", first("dqn_head"))
    elif config == "c3":
        # QR-DQN, grouped wide layer, ONE stream (RG_QR_STREAMS=0 in gpu_pmc_all.sh, so a step's launches keep their order):
        # forward of the per-action mean layer (a*), grouped forward of the target network, grouped SAVING forward of the
        # online network
        grouped = first("mlp_fwd_grouped_kernel")  # round 3: the grouped forwards are their own kernel instantiation
        if grouped:
            put("rg_mlp_forward_fused:save=0", grouped, lambda i, n: i % 2 == 0)   # target network, scattered output
            put("rg_mlp_forward_fused:save=1", grouped, lambda i, n: i % 2 == 1)   # online network, saving
            put("rg_mlp_forward_fused:save=0:mean_layer", fwd[:1])
        else:
            put("rg_mlp_forward_fused:save=0", fwd[:1], lambda i, n: i % 3 != 2)
            put("rg_mlp_forward_fused:save=1", fwd[:1], lambda i, n: i % 3 == 2)
        put("rg_mlp_backward_fused", bwd[:1])
        put("rg_mlp_wgrad_fused", ["wgrad_group_kernel", reduce_k])
        put("rg_group_head_wgrad", ["wgrad_grouped_kernel", "reduce_grouped_kernel"])
        put("rg_qr_compact_head", first("qr_compact_head_kernel"))
        put("rg_replay_dqn_batch", first("replay_dqn_batch_kernel"))
        put("rg_mlp_update_fused", first("mlp_update_tiles_kernel"))
    elif config == "c4":
        # SAC: mlp_bwd_fused_kernel runs three times per step — critic q1, critic q2 (full backward, the entry point
        # bench.py names as dominant), then the actor's
        put("rg_mlp_backward_fused", bwd[:1], lambda i, n: i % 3 != 2)
        put("rg_mlp_backward_fused:actor", bwd[:1], lambda i, n: i % 3 == 2)
        put("rg_mlp_backward_fused:dx_only", first("mlp_bwd_dx_kernel"))
        put("rg_mlp_forward_fused:all", fwd[:1])
        put("rg_mlp_wgrad_fused:all", ["wgrad_group_kernel", reduce_k])
        put("rg_replay_gather", first("replay_gather_kernel"))
        put("rg_replay_policy_batch", first("replay_policy_batch_kernel"))  # round 6: the one-launch sampler
        put("rg_mlp_update_fused:all", first("mlp_update_tiles_kernel"))
    # every rg:: kernel of the run, averaged over its launches (evidence; the keys above are what bench.py looks up)
    per_kernel = {}
    for nm in sorted(set(fetch) & set(write)):
        if nm.startswith("at::") or nm.startswith("__amd") or not fetch[nm] or not write[nm]:
            continue
        f, w = sum(fetch[nm]) / len(fetch[nm]), sum(write[nm]) / len(write[nm])
        per_kernel[nm.split("<")[0]] = {"hbm_bytes_per_launch": int(round((2.0 * f + w) * 1024)), "launches": len(fetch[nm])}
    path = os.path.join(ROOT, "profiles", "traffic.json")
    old = {}
    if os.path.exists(path):
        try:
            old = json.load(open(path))
        except Exception:
            old = {}
    stamp = bench.source_stamp()
    merged = dict(old.get("kernels", {})) if old.get("source_stamp") == stamp else {}
    merged.update(kernels)
    doc = {"source_stamp": stamp, "from": where,
           "correction": "KB units; FETCH_SIZE x2 on gfx950 for wide coalesced reads (MI355X_MICROARCH.md, HBM section); "
                         "hbm bytes per launch = 2 * FETCH_SIZE + WRITE_SIZE",
           "kernels": merged, "raw": {**(old.get("raw", {}) if old.get("source_stamp") == stamp else {}),
                                      **{f"{config}:{prec}:{k}": v for k, v in raw.items()}},
           "per_kernel": {**(old.get("per_kernel", {}) if old.get("source_stamp") == stamp else {}),
                          f"{config}:{prec}": per_kernel}}
    json.dump(doc, open(path, "w"), indent=1)
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Per-kernel-class summary of one rocprofv3 --pmc pass of SQ counters over bench.py.

    python tools/pmc_sq_summary.py <counter_collection.csv> <out.json>

Counters (one pass, <= 8 SQ): SQ_WAVES, SQ_BUSY_CYCLES, SQ_WAVE_CYCLES, SQ_INSTS_VALU, SQ_VALU_MFMA_BUSY_CYCLES,
SQ_LDS_BANK_CONFLICT, SQ_ACTIVE_INST_LDS, SQ_WAIT_INST_ANY.  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count
quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles (MI355X_MICROARCH.md, 's_memtime tick vs SQ PMC units')."""
import csv
import json
import sys
from collections import defaultdict

CLASSES = (("igemm3", "igemm3_bf16_kernel"), ("igemm2", "igemm2_bf16_kernel"), ("tconv3", "tconv3_bf16_kernel"),
           ("sconv9", "sconv9_bf16_kernel"), ("stem16", "stem16_bf16_kernel"), ("splitk_reduce", "igemm_splitk_reduce"),
           ("roi_align_mfma", "roi_align_fwd_mfma"), ("roi_align_avg", "roi_align_avg"), ("pair_pool", "pair_pool_kernel"),
           ("pair_combine", "pair_combine_kernel"), ("pair_infer", "pair_infer_kernel"), ("pair_mfma", "pair_mfma_kernel"), ("rank_clip", "rank_clip_topk"), ("tree_gru", "tree_gru"),
           ("nms_reduce", "nms_reduce"), ("nms_mask", "nms_mask"), ("nms_sort", "nms_sort"))

tot = defaultdict(lambda: defaultdict(float))
disp = defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"]
    cls = next((c for c, key in CLASSES if key in name), None)
    if cls is None:
        continue
    tot[cls][r["Counter_Name"]] += float(r["Counter_Value"])
    disp[cls].add(r["Dispatch_Id"])
out = {}
for cls, c in tot.items():
    n = len(disp[cls])
    d = {"launches": n}
    d.update({k: v / n for k, v in sorted(c.items())})
    if c.get("SQ_BUSY_CYCLES") and c.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        # MFMA-pipe busy share of the time a SIMD-group was busy; BUSY_CYCLES is summed over the shader engines' SQs
        d["mfma_busy_per_busy_cycle"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / c["SQ_BUSY_CYCLES"]
    if c.get("SQ_WAVE_CYCLES") and c.get("SQ_WAIT_INST_ANY"):
        d["wait_inst_any_share_of_wave_cycles"] = c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"]
    if c.get("SQ_ACTIVE_INST_LDS") and c.get("SQ_LDS_BANK_CONFLICT"):
        d["lds_bank_conflict_per_active_lds"] = c["SQ_LDS_BANK_CONFLICT"] / c["SQ_ACTIVE_INST_LDS"]
    out[cls] = d
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps({k: {kk: round(vv, 3) for kk, vv in v.items() if not kk.startswith("SQ_")} for k, v in out.items()}))

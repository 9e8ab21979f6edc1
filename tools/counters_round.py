"""Occupancy and issue shares of the three kernels the bench line quotes -- k_decode, k_decode2, k_encode_wsd -- from
separate rocprofv3 --pmc passes (--kernel-trace only, as MI355X_MICROARCH.md prescribes), tied to the kernels' sources:
    gpurun -- 'cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && python3 tools/counters_round.py r06'
writes gpurun_out/<tag>_counters.json (+ the raw means as gpurun_out/<tag>_counters.txt); bench.py quotes a record only when
its kernel_source_sha256 equals the hash of the sources it is run from.
SQ counters of this profiler count in units of four cycles (SQ_WAVE_CYCLES of a 12-wave workgroup = 3 x its cycles)."""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (the hashes)

tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
out = os.path.join(ROOT, "gpurun_out", f"{tag}_counters")
os.makedirs(out, exist_ok=True)
PASSES = ["SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS",
          "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES"]
# (kernel-name filter, minimum grid size in threads, child command, utterances, samples per utterance, streams per workgroup)
JOBS = {
    "k_decode": ("k_decode<", 256 * 768, [sys.executable, "tools/stamp_probe.py", "256"], 256, 100 * 160 - 17, 1),
    "k_decode2": ("k_decode2<", 256 * 768, [sys.executable, "tools/stamp_probe.py", "512"], 512, 100 * 160 - 17, 2),
    "k_encode_wsd": ("k_encode_wsd", 0, [sys.executable, "tools/ws_time.py", "--child"], 128, 300, 0),
}
txt = []
rec = {}
for kern, (pat, mingrid, cmd, B, per, spw) in JOBS.items():
    acc, dur = collections.defaultdict(list), []
    for i, counters in enumerate(PASSES):
        d = os.path.join(out, f"{kern}_{i}")
        subprocess.run(["rm", "-rf", d])
        r = subprocess.run(["rocprofv3", "--kernel-trace", "--pmc"] + counters.split() + ["--output-format", "csv", "-d", d, "-o", "r", "--"] + cmd,
                           cwd=ROOT, capture_output=True, text=True, timeout=400)
        if r.returncode != 0:
            txt.append(f"{kern} pass {i} FAILED: {r.stderr[-300:]}")
            continue
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if pat in row["Kernel_Name"] and int(row.get("Grid_Size", 0) or 0) >= mingrid:
                    acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
        for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if pat in row["Kernel_Name"] and int(row.get("Grid_Size", row.get("Grid_Size_X", 0)) or 0) >= mingrid:
                    dur.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-9)
    if not acc:
        continue
    m = {k: sum(v) / len(v) for k, v in acc.items()}
    txt.append(f"{kern}: means over the launches, raw counter values")
    for k in sorted(m):
        txt.append(f"  {k:24s} {m[k]:16.1f}   ({len(acc[k])} launches)")
    if kern.startswith("k_decode"):
        wgs = B // spw
        unit = wgs * per  # one sample of one workgroup (k_decode2: a PAIR of samples, one of each utterance)
        cyc = 4.0 * m["SQ_WAVE_CYCLES"] / 12.0 / unit  # 12 waves per workgroup
        e = {"kernel_source_sha256": bench.decode_kernel_hash() if spw == 1 else bench.decode2_kernel_hash(),
             "workload": f"{B} utterances x 100 frames (tools/stamp_probe.py), {wgs} workgroups",
             "waves_per_workgroup": 12, "waves_per_simd": 3, "max_waves_per_simd": 8, "occupancy_waves": 3 / 8,
             "occupancy_limited_by": "168 VGPRs per wave (512 / 168 = 3 waves per SIMD) and one workgroup's LDS per CU",
             "cus_busy": min(wgs, 256), "cus": 256,
             "cycles_per_workgroup_sample": cyc, "utterances_per_workgroup": spw,
             "valu_instructions": m["SQ_INSTS_VALU"] / unit, "lds_instructions": m["SQ_INSTS_LDS"] / unit,
             "salu_instructions": m["SQ_INSTS_SALU"] / unit,
             # every VALU instruction holds its SIMD's vector issue for four cycles; four SIMDs
             "valu_issue_floor_cycles": 4.0 * m["SQ_INSTS_VALU"] / unit / 4.0,
             "valu_busy_share": (4.0 * m["SQ_ACTIVE_INST_VALU"] / unit / 4.0) / cyc,
             "lds_busy_share": (m["SQ_LDS_IDX_ACTIVE"] / unit) / cyc,
             "lds_bank_conflict_share_of_lds": m["SQ_LDS_BANK_CONFLICT"] / max(m["SQ_LDS_IDX_ACTIVE"], 1.0),
             "wave_wait_share": m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"],
             "note": "per sample of one workgroup (k_decode2: per PAIR of samples, one of each of its two utterances); "
                     "valu_issue_floor_cycles = VALU instructions x 4 cycles / 4 SIMDs: the time the vector issue alone needs; "
                     "the north_star's 200x real time for one utterance is 750 cycles per sample"}
        if dur:
            e["launch_ms_under_the_profiler"] = 1e3 * sum(dur) / len(dur)
        rec[kern] = e
    else:
        e = {"kernel_source_sha256": bench.predictor_kernel_hash(), "workload": "128 utterances x 300 frames (tools/ws_time.py)",
             "workgroups": 256, "waves_per_workgroup": 8, "waves_per_simd": 2, "max_waves_per_simd": 8, "occupancy_waves": 2 / 8,
             "occupancy_limited_by": "255 VGPRs per wave (2 waves per SIMD) and 153 kB of LDS per workgroup (one per CU)",
             "cus_busy": 256, "cus": 256,
             "wave_wait_share": m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"],
             "valu_instructions_per_frame_and_workgroup": m["SQ_INSTS_VALU"] / (256 * 300.0),
             "lds_bank_conflict_share_of_lds": m["SQ_LDS_BANK_CONFLICT"] / max(m["SQ_LDS_IDX_ACTIVE"], 1.0),
             "note": "SQ_WAIT_ANY / SQ_WAVE_CYCLES: the share of wave cycles spent waiting (the frame's four L2 round trips)"}
        rec[kern] = e
json.dump(rec, open(os.path.join(ROOT, "gpurun_out", f"{tag}_counters.json"), "w"), indent=1)
open(os.path.join(ROOT, "gpurun_out", f"{tag}_counters.txt"), "w").write("\n".join(txt) + "\n")
print("\n".join(txt))
print(json.dumps(rec, indent=1)[:3000])

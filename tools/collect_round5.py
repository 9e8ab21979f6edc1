"""copies the summaries of `tools/measure_round5.sh` (gpurun_out/r05m, gpurun_out/r05*) into profiles/r05_* (tracked)"""
import csv, glob, json, os, shutil
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
o = os.path.join(R, "gpurun_out", "r05m")
P = os.path.join(R, "profiles")
head = open(os.path.join(o, "head.txt")).read().strip() if os.path.exists(os.path.join(o, "head.txt")) else "?"
shutil.copy(os.path.join(o, "bench.json"), os.path.join(P, "r05_bench.json"))
for f in glob.glob(os.path.join(o, "prof", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(P, "r05_kernel_stats.csv"))
rows = []
for f in glob.glob(os.path.join(o, "prof", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_decode<" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r.get("Grid_Size_X", ""),
                         r.get("VGPR_Count", ""), r.get("LDS_Block_Size", "")))
rows.sort()
with open(os.path.join(P, "r05_kernel_trace_k_decode.txt"), "w") as f:
    f.write(f"rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline   (tree {head})\n")
    f.write("k_decode launches in time order: duration ms, grid size, VGPRs, LDS bytes\n")
    for _, ms, g, v, l in rows:
        f.write(f"  {ms:9.3f}  grid {g}  vgpr {v}  lds {l}\n")
if os.path.exists(os.path.join(R, "gpurun_out", "r05_traffic.json")):
    shutil.copy(os.path.join(R, "gpurun_out", "r05_traffic.json"), os.path.join(P, "r05_traffic.json"))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(os.path.join(R, "gpurun_out", "r05_traffic", c, "*counter_collection.csv")):
        keep = [l for i, l in enumerate(open(f)) if i == 0 or "k_decode" in l]
        open(os.path.join(P, f"r05_pmc_{c.lower()}.csv"), "w").writelines(keep)
shutil.copy(os.path.join(o, "pmc_decode.txt"), os.path.join(P, "r05_pmc_decode.txt"))
e = os.path.join(R, "gpurun_out", "r05", "ws_pmc")
if os.path.exists(os.path.join(e, "summary.txt")):
    shutil.copy(os.path.join(e, "summary.txt"), os.path.join(P, "r05_encode_pmc_summary.txt"))
    for f in glob.glob(os.path.join(e, "trace", "**", "*kernel_stats.csv"), recursive=True):
        shutil.copy(f, os.path.join(P, "r05_encode_kernel_stats.csv"))
txt = open(os.path.join(o, "ws_check.txt")).read() + \
    "\n---- stage profile (-DFPC_WS_PROF -DFPC_WS_PROF_TAIL build, workgroup 5 of group 0; cycles per frame incl. ~150 per stamp) ----\n" + \
    "\n".join(l for l in open(os.path.join(o, "ws_prof.txt")).read().splitlines() if "cycles/frame" in l or "frames" in l or "frame above" in l)
open(os.path.join(P, "r05_predictor_ws.txt"), "w").write(f"tree {head}; library: " + open(os.path.join(o, "build_info.txt")).read() + txt + "\n")
t = os.path.join(R, "gpurun_out", "r05", "train_prof", "kernel_stats.csv")
if os.path.exists(t):
    shutil.copy(t, os.path.join(P, "r05_train_kernel_stats.csv"))
tp = os.path.join(R, "gpurun_out", "r05", "train_pmc", "summary.txt")
if os.path.exists(tp):
    shutil.copy(tp, os.path.join(P, "r05_train_pmc_summary.txt"))
open(os.path.join(P, "r05_train_timing.txt"), "w").write(
    f"tree {head}; tools/time_train.py at the reference's batch (100 x 150, train_frame.py:198-204), twice; then the backward pass on the "
    "row-split kernel; then the stage profiles of k_train_bwd_ws (one thread of each track); the rocprofv3 kernel table of the step: r05_train_kernel_stats.csv\n" +
    "".join(l for l in open(os.path.join(o, "train.txt")) if "amdgpu.ids" not in l))
km = os.path.join(o, "kmeans.txt")
if os.path.exists(km):
    old = open(os.path.join(P, "r05_kmeans.txt")).read().split("\n\n")
    hdr = "\n".join(l for l in old[0].splitlines() if not l.startswith("(tree "))
    open(os.path.join(P, "r05_kmeans.txt"), "w").write(hdr + f"\n(tree {head})\n\n" + open(km).read() + "\n" + "\n\n".join(old[2:]))
print("collected into profiles/r05_*")

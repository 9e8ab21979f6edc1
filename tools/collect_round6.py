"""copies the summaries of `tools/measure_round6.sh` (gpurun_out/r06m, gpurun_out/r06_*) into profiles/r06_* (tracked)"""
import csv, glob, os, shutil
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
o = os.path.join(R, "gpurun_out", "r06m")
P = os.path.join(R, "profiles")
head = open(os.path.join(o, "head.txt")).read().strip() if os.path.exists(os.path.join(o, "head.txt")) else "?"
shutil.copy(os.path.join(o, "bench.json"), os.path.join(P, "r06_bench.json"))
for f in glob.glob(os.path.join(o, "prof", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(P, "r06_kernel_stats.csv"))
rows = []
for f in glob.glob(os.path.join(o, "prof", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_decode" in r["Kernel_Name"]:
            name = "k_decode2" if "k_decode2" in r["Kernel_Name"] else "k_decode "
            rows.append((int(r["Start_Timestamp"]), name, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r.get("Grid_Size_X", r.get("Grid_Size", "")),
                         r.get("VGPR_Count", ""), r.get("LDS_Block_Size", "")))
rows.sort()
with open(os.path.join(P, "r06_kernel_trace_k_decode.txt"), "w") as f:
    f.write(f"rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline   (tree {head})\n")
    f.write("k_decode / k_decode2 launches in time order: duration ms, grid size (threads), VGPRs, LDS bytes\n")
    for _, n, ms, g, v, l in rows:
        f.write(f"  {n} {ms:9.3f}  grid {g}  vgpr {v}  lds {l}\n")
for n in ("r06_traffic.json", "r06_counters.json", "r06_counters.txt"):
    if os.path.exists(os.path.join(R, "gpurun_out", n)):
        shutil.copy(os.path.join(R, "gpurun_out", n), os.path.join(P, n))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(os.path.join(R, "gpurun_out", "r06_traffic", c, "*counter_collection.csv")):
        keep = [l for i, l in enumerate(open(f)) if i == 0 or "k_decode" in l]
        open(os.path.join(P, f"r06_pmc_{c.lower()}.csv"), "w").writelines(keep)
hdr = f"tree {head}; library: " + open(os.path.join(o, "build_info.txt")).read()
open(os.path.join(P, "r06_many_stream.txt"), "w").write(
    hdr + "tools/pair_probe.py: parity of k_decode2 (small cases, both instances, chunked) and the decode launch (HIP events) of B utterances x 300 "
    "frames as rounds of k_decode (one/wg) and on k_decode2 (two/wg); last line: half of the frames voiced\n" + open(os.path.join(o, "pair_probe.txt")).read())
open(os.path.join(P, "r06_phase_stamps.txt"), "w").write(
    hdr + "FPC_DECODE_STAMPS=1 (diagnostic instances <true, ..>: raw arrival stamps at the barriers, block 0, 63 samples; the stamps cost\n"
    "cycles and change the register allocation: the phase SHARES are what to read, the totals are those of r06_many_stream.txt), B = 512 x 100 frames\n" +
    open(os.path.join(o, "stamps.txt")).read())
print("collected into profiles/r06_*")

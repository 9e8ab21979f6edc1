#!/bin/bash
# per-kernel resources of one HIP source of csrc/ (VGPRs, spills, LDS) from the code object's metadata: no GPU needed
# usage: tools/kernel_resources.sh predictor.hip [extra -D flags]
set -e
cd "$(dirname "$0")/../feature-predictor-for-speech-codec_amd/csrc"
src=$1; shift
out=$(mktemp /tmp/kres.XXXXXX.co)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I../../include --cuda-device-only -c "$src" -o "$out" "$@"
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input="$out" --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output="$out.elf"
/opt/rocm/lib/llvm/bin/llvm-readelf --notes "$out.elf" | python3 -c '
import sys, re
txt = sys.stdin.read()
ks = re.findall(r"\.group_segment_fixed_size:\s*(\d+).*?\.name:\s*(\S+).*?\.sgpr_count:\s*(\d+).*?\.vgpr_count:\s*(\d+).*?\.vgpr_spill_count:\s*(\d+)", txt, re.S)
print(f"{len(ks)} kernels")
for lds, name, sg, vg, sp in ks:
    print(f"  {name[:70]:70s} vgpr {vg:>3s} spill {sp:>3s} sgpr {sg:>3s} lds {lds:>6s}")
'
rm -f "$out" "$out.elf"

#!/bin/bash
# The PMC passes behind backbone_bf16_b64.roofline.{traffic, peak_sustained} (BASELINE configs[2]):
#   two --pmc passes (FETCH_SIZE, WRITE_SIZE: separate runs) of `bench.py --workload backbone_bf16` -> pmc_traffic_bf16.json
#   one --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES pass of the same command                       -> pmc_busy.json (+ table)
#   gpurun -- 'bash scripts/lp_pmc.sh <tag> <commit>'     results in gpurun_out/<tag>/
tag=${1:-lp_pmc}; commit=${2:-}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"; cd "$root"
export TMPDIR=/tmp
db() { find "$1" -name '*.db' | head -1; }
lp="--workload backbone_bf16 --steps 20 --min-seconds 0"
for c in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c -d "$out/pmc_lp_$c" -- python "$root/bench.py" $lp > /dev/null 2>> "$out/prof.err")
    python scripts/rocpd_pmc.py "$(db "$out/pmc_lp_$c")" > "$out/config3_pmc_${c,,}_kb.txt"
done
python scripts/pmc_lp_traffic.py "$(db "$out/pmc_lp_FETCH_SIZE")" "$(db "$out/pmc_lp_WRITE_SIZE")" "$out/pmc_traffic_bf16.json" "$commit" > /dev/null
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES -d "$out/pmc_lp_busy" -- python "$root/bench.py" $lp > /dev/null 2>> "$out/prof.err")
cp "$root/profiles/pmc_busy.json" "$out/pmc_busy.json" 2>/dev/null
python scripts/pmc_busy.py "$(db "$out/pmc_lp_busy")" "$out/pmc_busy.json" "$commit" 20 > "$out/config3_mfma_busy.txt"
rm -rf "$out/pmc_lp_FETCH_SIZE" "$out/pmc_lp_WRITE_SIZE" "$out/pmc_lp_busy"
python -c "import json; j = json.load(open('$out/pmc_traffic_bf16.json')); print(j['hbm_bytes_per_step'], j['launches_per_step'], j['_meta'])"
head -12 "$out/config3_mfma_busy.txt"

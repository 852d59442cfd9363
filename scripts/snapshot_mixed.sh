#!/bin/bash
# The configs[4] part of scripts/round_snapshot.sh alone (kernel table, FETCH / WRITE passes -> pmc_traffic_mixed.json, bench line):
#   gpurun -- 'bash scripts/snapshot_mixed.sh round5_b <commit>'
tag=${1:-snapshot}; commit=${2:-}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"; cd "$root"
export TMPDIR=/tmp
db() { find "$1" -name '*.db' | head -1; }
mx="--workload track_mixed --batch 32 --steps 20 --min-seconds 0"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$out/prof_mx" -- python "$root/bench.py" $mx > "$out/config5_profiled_bench.json" 2>> "$out/prof.err")
python scripts/rocpd_stats.py "$(db "$out/prof_mx")" > "$out/config5_kernel_stats.txt"
for c in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c -d "$out/pmc_mx_$c" -- python "$root/bench.py" $mx > /dev/null 2>> "$out/prof.err")
    python scripts/rocpd_pmc.py "$(db "$out/pmc_mx_$c")" > "$out/config5_pmc_${c,,}_kb.txt"
done
python scripts/pmc_lp_traffic.py "$(db "$out/pmc_mx_FETCH_SIZE")" "$(db "$out/pmc_mx_WRITE_SIZE")" "$out/pmc_traffic_mixed.json" "$commit" stem_pool_lp "bench.py $mx" > /dev/null
rm -rf "$out/prof_mx" "$out/pmc_mx_FETCH_SIZE" "$out/pmc_mx_WRITE_SIZE"
timeout 300 python bench.py --workload track_mixed --batch 32 > "$out/config5_fp16_mixed_b32_bench.json" 2>> "$out/bench.err"
head -c 600 "$out/config5_fp16_mixed_b32_bench.json"; echo; head -8 "$out/config5_kernel_stats.txt"

#!/bin/bash
# One GPU-box pass that produces everything profiles/<tag>_* holds: smoke, the gpu test suite, the default bench
# line, the kernel table of the headline workload under rocprofv3, and the two PMC passes behind roofline.traffic.
#   gpurun --timeout 1500 -- 'bash scripts/round_snapshot.sh round2_c <commit>'
# Results land in gpurun_out/<tag>/ (scratch); copy what is to be judged into profiles/.
tag=${1:-snapshot}; commit=${2:-}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"; cd "$root"
export TMPDIR=/tmp
hl="--min-seconds 0 --no-extras --no-xcorr --no-cpu-baseline"
db() { find "$1" -name '*.db' | head -1; }

timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$out/smoke.log" 2>&1
timeout 1500 python -m pytest tests -m gpu -q > "$out/pytest.log" 2>&1; tail -3 "$out/pytest.log"
timeout 600 python bench.py --no-cpu-baseline > "$out/bench_before_counters.json" 2> "$out/bench.err"
timeout 300 python bench.py --streams-per-gpu 4 --no-extras --no-xcorr --no-cpu-baseline > "$out/streams4_bench.json" 2>> "$out/bench.err"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$out/prof" -- python "$root/bench.py" --steps 100 --warmup 10 $hl > "$out/profiled_bench.json" 2> "$out/prof.err")
python scripts/rocpd_stats.py "$(db "$out/prof")" > "$out/kernel_stats.txt"
for c in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c -d "$out/pmc_$c" -- python "$root/bench.py" --steps 30 $hl > /dev/null 2>> "$out/prof.err")
    python scripts/rocpd_pmc.py "$(db "$out/pmc_$c")" > "$out/pmc_${c,,}_kb.txt"
done
python scripts/pmc_to_traffic.py "$(db "$out/pmc_FETCH_SIZE")" "$(db "$out/pmc_WRITE_SIZE")" "$out/pmc_traffic.json" "$commit" > /dev/null
# configs[2] (batch-64 bf16 backbone): kernel table + the two PMC passes behind backbone_bf16_b64.roofline.traffic
lp="--workload backbone_bf16 --steps 20 --min-seconds 0"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$out/prof_lp" -- python "$root/bench.py" $lp > "$out/config3_bf16_profiled_bench.json" 2>> "$out/prof.err")
python scripts/rocpd_stats.py "$(db "$out/prof_lp")" > "$out/config3_kernel_stats.txt"
for c in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c -d "$out/pmc_lp_$c" -- python "$root/bench.py" $lp > /dev/null 2>> "$out/prof.err")
    python scripts/rocpd_pmc.py "$(db "$out/pmc_lp_$c")" > "$out/config3_pmc_${c,,}_kb.txt"
done
python scripts/pmc_lp_traffic.py "$(db "$out/pmc_lp_FETCH_SIZE")" "$(db "$out/pmc_lp_WRITE_SIZE")" "$out/pmc_traffic_bf16.json" "$commit" > /dev/null
# configs[4] per-GPU share (32 streams, fp16 backbone + head convs, fp32 xcorr): kernel table + the two PMC passes behind
# track_mixed_b32.roofline.traffic (profiles/pmc_traffic_mixed.json) + the busy pass of its kernels (below)
mx="--workload track_mixed --batch 32 --steps 20 --min-seconds 0"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$out/prof_mx" -- python "$root/bench.py" $mx > "$out/config5_profiled_bench.json" 2>> "$out/prof.err")
python scripts/rocpd_stats.py "$(db "$out/prof_mx")" > "$out/config5_kernel_stats.txt"
for c in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c -d "$out/pmc_mx_$c" -- python "$root/bench.py" $mx > /dev/null 2>> "$out/prof.err")
    python scripts/rocpd_pmc.py "$(db "$out/pmc_mx_$c")" > "$out/config5_pmc_${c,,}_kb.txt"
done
python scripts/pmc_lp_traffic.py "$(db "$out/pmc_mx_FETCH_SIZE")" "$(db "$out/pmc_mx_WRITE_SIZE")" "$out/pmc_traffic_mixed.json" "$commit" stem_pool_lp "bench.py $mx" > /dev/null
# clock + matrix-pipe utilisation (SQ_BUSY_CYCLES, SQ_VALU_MFMA_BUSY_CYCLES) of the bf16 step's kernels and of the fp32 frame's
rm -f "$out/pmc_busy.json"      # start empty: an entry merged from an older pass would carry this pass's csrc_tree
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES -d "$out/pmc_lp_busy" -- python "$root/bench.py" $lp > /dev/null 2>> "$out/prof.err")
python scripts/pmc_busy.py "$(db "$out/pmc_lp_busy")" "$out/pmc_busy.json" "$commit" 20 > "$out/config3_mfma_busy.txt"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES -d "$out/pmc_mx_busy" -- python "$root/bench.py" $mx > /dev/null 2>> "$out/prof.err")
python scripts/pmc_busy.py "$(db "$out/pmc_mx_busy")" "$out/pmc_busy.json" "$commit" 20 > "$out/config5_mfma_busy.txt"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES -d "$out/pmc_busy_f32" -- python "$root/bench.py" --steps 30 $hl > /dev/null 2>> "$out/prof.err")
python scripts/pmc_busy.py "$(db "$out/pmc_busy_f32")" "$out/pmc_busy.json" "$commit" 8 > "$out/mfma_busy.txt"
# GroupDW at 2048 samples: kernel table + the two PMC passes behind xcorr_hbm.traffic
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$out/prof_xc" -- python "$root/scripts/xcorr_pmc_run.py" > /dev/null 2>> "$out/prof.err")
python scripts/rocpd_stats.py "$(db "$out/prof_xc")" > "$out/xcorr_kernel_stats.txt"
for c in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c -d "$out/pmc_xc_$c" -- python "$root/scripts/xcorr_pmc_run.py" > /dev/null 2>> "$out/prof.err")
    python scripts/rocpd_pmc.py "$(db "$out/pmc_xc_$c")" > "$out/xcorr_pmc_${c,,}_kb.txt"
done
python scripts/pmc_xcorr_to_json.py "$(db "$out/pmc_xc_FETCH_SIZE")" "$(db "$out/pmc_xc_WRITE_SIZE")" 2048 "$out/pmc_xcorr.json" "$commit" > /dev/null
rm -rf "$out/pmc_lp_busy" "$out/pmc_mx_busy" "$out/prof_mx" "$out/pmc_mx_FETCH_SIZE" "$out/pmc_mx_WRITE_SIZE" "$out/pmc_busy_f32" "$out/prof_xc" "$out/pmc_xc_FETCH_SIZE" "$out/pmc_xc_WRITE_SIZE"
# the three bench lines LAST, with this pass's counters in place (profiles/pmc_*.json are keyed to the source tree: bench.py prints null
# for traffic / mfma_busy / clock when they were measured on another one) - on the box only; copy them into profiles/ at home as well
cp "$out"/pmc_traffic.json "$out"/pmc_busy.json "$out"/pmc_traffic_bf16.json "$out"/pmc_traffic_mixed.json "$out"/pmc_xcorr.json profiles/
timeout 600 python bench.py > "$out/bench.json" 2>> "$out/bench.err"
timeout 300 python bench.py --workload backbone_bf16 > "$out/config3_bf16_bench.json" 2>> "$out/bench.err"
timeout 300 python bench.py --workload track_mixed --batch 32 > "$out/config5_fp16_mixed_b32_bench.json" 2>> "$out/bench.err"
rm -rf "$out/prof" "$out/pmc_FETCH_SIZE" "$out/pmc_WRITE_SIZE" "$out/prof_lp" "$out/pmc_lp_FETCH_SIZE" "$out/pmc_lp_WRITE_SIZE"
cat "$out/smoke.log" | tail -2; head -c 600 "$out/bench.json"; echo; head -12 "$out/kernel_stats.txt"

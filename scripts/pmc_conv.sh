#!/bin/bash
# Hardware-counter picture of ONE fp32 conv tile on one shape: which unit a k-step waits for.
#   gpurun -- 'bash scripts/pmc_conv.sh conf 55 [lib.so]'   -> gpurun_out/pmc_conv_<shape>_<tile>.txt
# Separate --pmc passes (kernel-trace only), per-dispatch means from the rocpd database.
shape=${1:-conf}; tile=${2:-55}; lib=${3:-usot_amd/csrc/libusot_hip.so}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/pmc_conv_${shape}_${tile}.txt
mkdir -p "$root/gpurun_out"; : > "$out"
export TMPDIR=/tmp
i=0
for set in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_DATA_FIFO_FULL" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_INSTS_VALU" \
           "TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_LATENCY_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum"; do
    i=$((i+1))
    d=/tmp/pmc_conv_$i
    rm -rf $d
    (cd /tmp && SHAPE=$shape timeout 300 rocprofv3 --kernel-trace --pmc $set -d $d -- python "$root/scripts/ablate_kstep.py" "$root/$lib" $tile > /dev/null 2>> /tmp/pmc_conv.err)
    db=$(find $d -name '*.db' | head -1)
    [ -n "$db" ] && python "$root/scripts/rocpd_pmc.py" "$db" | grep -E "conv_igemm|conv_wstat|counter" >> "$out"
done
cat "$out"

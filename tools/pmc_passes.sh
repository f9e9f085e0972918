#!/bin/bash
# Collects PMC counters for the step kernels, one rocprofv3 run per counter group (a run can hold
# at most 8 SQ, 4 TCC, 4 TCP, 2 TA, 2 TD and 2 GRBM counters).  Usage, on the GPU box:
#   tools/pmc_passes.sh <variant f64|fr> <out dir under gpurun_out/> [extra bench args]
# PMC_ONLY="0 9 11": run only these counter groups (indices into the list below).
# PMC_SCRIPT=tools/bench_share.py: profile that script (with the extra args) instead of bench.py; PMC_KERNEL=<substring>
# picks the kernels the summary lists (default k_step).
set -u
VARIANT=$1; OUT=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$ROOT/gpurun_out/$OUT"
cd /tmp && export TMPDIR=/tmp
CGROUPS=(
 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD"
 "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR"
 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
 "TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_DRAM_sum"
 "TCC_READ_sum TCC_READ_SECTORS_sum TCC_STREAMING_REQ_sum TCC_EA0_WRREQ_64B_sum"
 "FETCH_SIZE"
 "WRITE_SIZE"
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
 "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum"
 "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE GRBM_COUNT"
 "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TD_TD_BUSY_sum TD_TC_STALL_sum"
 "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_LDS"
)
i=0
for g in "${CGROUPS[@]}"; do
  if [ -n "${PMC_ONLY:-}" ] && ! echo " $PMC_ONLY " | grep -q " $i "; then i=$((i+1)); continue; fi
  d="$ROOT/gpurun_out/$OUT/pass$i"
  if [ -n "${PMC_SCRIPT:-}" ]; then
    timeout -s KILL ${PMC_TIMEOUT:-240} rocprofv3 --pmc $g --output-format csv -d "$d" -- python3 "$ROOT/$PMC_SCRIPT" "$@" > "$ROOT/gpurun_out/$OUT/pass$i.log" 2>&1
  else
    timeout -s KILL ${PMC_TIMEOUT:-120} rocprofv3 --pmc $g --output-format csv -d "$d" -- python3 "$ROOT/bench.py" --variant "$VARIANT" --steps 2 --warmup 1 --no-cpu-baseline --no-extras "$@" > "$ROOT/gpurun_out/$OUT/pass$i.log" 2>&1
  fi
  rc=$?
  echo "pass $i rc=$rc : $g"
  if [ $rc -ne 0 ]; then tail -5 "$ROOT/gpurun_out/$OUT/pass$i.log"; exit $rc; fi
  i=$((i+1))
done
python3 "$ROOT/tools/pmc_summary.py" "$ROOT/gpurun_out/$OUT" ${PMC_KERNEL:-k_step} > "$ROOT/gpurun_out/$OUT/summary.txt"
cat "$ROOT/gpurun_out/$OUT/summary.txt"

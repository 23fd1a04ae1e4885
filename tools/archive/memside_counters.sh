#!/bin/bash
# tools/archive/memside_counters.sh [bytes] -- run on the GPU box from the repo root.  Memory-side counters of the
# L2 -> fabric interface (TCC_EA*), the L2's own stall counters and the vector L1's pending stalls, for four kernels with
# the SAME schedule at the same size: read-only, write-only, copy, and the product's cycle kernel (tools/ubench_queue_rw).
# One rocprofv3 --pmc pass per counter group (TCC has 4 slots per pass on gfx950); never together with a trace option.
# A pass rocprofv3 refuses is recorded as refused and the script goes on.  tools/archive/summarize_memside.py distils the CSVs.
BYTES=${1:-4294967296}
OUT=gpurun_out/memside
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
PASSES=(
 "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_CYCLE_sum"
 "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum"
 "TCC_EA0_RDREQ_GMI_CREDIT_STALL_sum TCC_EA0_RDREQ_IO_CREDIT_STALL_sum TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum TCC_EA0_WRREQ_IO_CREDIT_STALL_sum"
 "TCC_TAG_STALL_sum TCC_BUSY_sum TCC_REQ_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum"
 "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_32B_sum"
 "TCC_HIT_sum TCC_MISS_sum TCC_READ_sum TCC_WRITE_sum"
 "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"
 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"
)
i=0
for P in "${PASSES[@]}"; do
  i=$((i+1))
  echo "== pass $i: $P"
  if timeout -k 10 240 rocprofv3 --pmc $P --output-format csv -d $OUT/pass$i -- tools/ubench_queue_rw $BYTES 200 2 > $OUT/pass$i.log 2>&1; then
    echo "pass $i ok" >> $OUT/status.txt
  else
    echo "pass $i REFUSED/FAILED as a group: $P" >> $OUT/status.txt
    tail -3 $OUT/pass$i.log
    j=0
    for C in $P; do   # one by one: which of them does gfx950 refuse?
      j=$((j+1))
      if timeout -k 10 240 rocprofv3 --pmc $C --output-format csv -d $OUT/pass${i}_$j -- tools/ubench_queue_rw $BYTES 200 2 > $OUT/pass${i}_$j.log 2>&1; then
        echo "  $C ok alone" >> $OUT/status.txt
      else
        echo "  $C REFUSED: $(grep -i -m1 "error\|not\|invalid\|unsupported" $OUT/pass${i}_$j.log | cut -c1-160)" >> $OUT/status.txt
      fi
    done
  fi
done
# the same program without the profiler, for the rates the counters belong to
timeout -k 10 120 tools/ubench_queue_rw $BYTES 200 8 > $OUT/rates.txt 2>&1
cat $OUT/status.txt $OUT/rates.txt
python3 tools/archive/summarize_memside.py $OUT > $OUT/summary.log 2>&1 || true
tail -60 $OUT/summary.log

#!/bin/bash
# GPU box: three rocprofv3 --pmc passes (counters alone: no trace domains) over tools/trace_c5shard.py -- the sharded SVGD step of
# one rank of eight at C5 (ring forward, head, four-wave weight gradients, Gram pass, kernel matrix, combine).
#   usage (through gpurun): bash tools/run_pmc_shard.sh <tag>  ->  gpurun_out/pmc_<tag>/p{1,2,3}; tools/pmc_agg.py reads them
set -e
R=$PWD
OUT=$R/gpurun_out/pmc_$1
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout -k 10 200 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD --output-format csv -d $OUT/p1 -- python3 $R/tools/trace_c5shard.py local > $OUT/p1.log 2>&1
timeout -k 10 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum --output-format csv -d $OUT/p2 -- python3 $R/tools/trace_c5shard.py local > $OUT/p2.log 2>&1
timeout -k 10 200 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TCP_TCP_TA_DATA_STALL_CYCLES_sum --output-format csv -d $OUT/p3 -- python3 $R/tools/trace_c5shard.py local > $OUT/p3.log 2>&1
cd $R
python3 tools/pmc_agg.py $OUT/p1 $OUT/p2 $OUT/p3 > $OUT/counters.jsonl
cat $OUT/counters.jsonl | cut -c1-400

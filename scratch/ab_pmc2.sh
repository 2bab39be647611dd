#!/bin/bash
# usage: scratch/ab_pmc2.sh <tag> <rows> <variant filter> : memory-side PMC passes (L2 <-> fabric, L1 <-> L2 latencies, TA stalls) of scratch/bin/headline_ab
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=$1; ROWS=$2; FILT=$3
OUT=gpurun_out/abpmc_$TAG
rm -rf $OUT && mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- scratch/bin/headline_ab $ROWS 5 $FILT > $OUT/stats.log 2>&1
i=0
for C in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" \
         "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
         "TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_NORMAL_WRITEBACK_sum TCC_NORMAL_EVICT_sum" \
         "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum TCC_BUSY_sum" \
         "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" \
         "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_BUSY_avr" \
         "SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY" \
         "TCC_WRITE_sum TCC_READ_sum TCC_STREAMING_REQ_sum TCC_WRITEBACK_sum" \
         "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" \
         "TCC_EA0_WRREQ_DRAM_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_WRITE_DRAM_32B_sum TCC_EA0_RDREQ_DRAM_32B_sum" ; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/p$i -- scratch/bin/headline_ab $ROWS 2 $FILT > $OUT/p$i.log 2>&1 || tail -3 $OUT/p$i.log
done
python3 - <<PY | tee $OUT/summary.txt
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "tile_kernel" in k or "rw_ceiling" in k or "stream_kernel" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for f in glob.glob("$OUT/stats/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
names = sorted(acc)
short = [re.sub(r"void |\(.*", "", n).replace("tile_kernel", "T").replace("rw_ceiling_kernel", "RW").replace(" ", "") for n in names]
counters = sorted({c for n in names for c in acc[n]})
print("%-38s" % "counter" + "".join("%22s" % s for s in short))
print("%-38s" % "median ms (kernel trace)" + "".join("%22.4f" % sorted(dur.get(n, [0]))[len(dur.get(n, [0])) // 2] for n in names))
for c in counters:
    print("%-38s" % c + "".join("%22.5g" % (sum(acc[n][c]) / len(acc[n][c]) if acc[n].get(c) else float("nan")) for n in names))
PY

# What bounds K1 at 512^3: one launch each of the one-column kernel (8 B per lane), the two-column kernel (16 B per lane) and a flat copy of the same
# arrays, under five rocprofv3 --pmc passes (vector-memory instruction counts, L1->L2 requests, L2 hits, L2->fabric requests by size, stalls / queue levels)
set -x
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03g
mkdir -p $O
V="one:INS_DISABLE_FLUX128=1 two:"
i=0
for C in "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM" \
         "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" \
         "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum" \
         "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum TCP_TCC_READ_REQ_LATENCY_sum GRBM_GUI_ACTIVE TCC_BUSY_avr" \
         "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C -d $O/p$i -o c --output-format csv -- python3 tools/k1_lab.py 512 --once $V > $O/p$i.log 2>&1 || { tail -5 $O/p$i.log; }
done
python3 tools/pmc_table.py "k_flux64,k_flux128,elementwise_kernel" $O/p1/c_counter_collection.csv $O/p2/c_counter_collection.csv $O/p3/c_counter_collection.csv $O/p4/c_counter_collection.csv $O/p5/c_counter_collection.csv > $O/k1_pmc_512.txt 2>&1
cat $O/k1_pmc_512.txt
python3 tools/k1_lab.py 512 $V > $O/k1_lab_512.txt 2>&1; grep -v amdgpu $O/k1_lab_512.txt

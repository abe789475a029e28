set -u
# The four GRBM_* counters were ONE pass in round 4 and rocprofv3 aborted on it before any kernel ran ("error code 38: Request
# exceeds the capabilities of the hardware to collect", gpurun_out/r4c/pmc4.log): the GRBM block has two counter slots.  They
# are two passes of two now.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${PMC_OUT:-r4c}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS" \
           "TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum" \
           "TCC_EA0_WRREQ_LEVEL TCC_EA0_RDREQ_LEVEL TCC_EA0_RDREQ_DRAM_CREDIT_STALL TCC_EA0_WRREQ_64B" \
           "GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE" \
           "GRBM_EA_BUSY GRBM_TC_BUSY" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS TCP_UTCL1_STALL_INFLIGHT_MAX TCP_UTCL1_SERIALIZATION_STALL"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $O/pmc$i -- python3 $R/tools/probe_twophase_counters.py > $O/pmc$i.log 2>&1 || { echo "pass $i failed"; tail -5 $O/pmc$i.log; }
  echo "== pass $i: $set" >> $O/summary.txt
  python3 $R/tools/pmc_summary.py $O/pmc$i tp_expand >> $O/summary.txt 2>&1
done
tail -50 $O/summary.txt

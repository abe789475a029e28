# tools/pmc_probe_regacc.sh - L2 / HBM counters of tools/probe_regacc.hip on C2, one rocprofv3 --pmc pass per counter set (run on the GPU box;
# build/tools/probe_regacc must have been built here first, see the probe's header).  Summary: gpurun_out/s2d/summary.txt
set -e
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/s2d; mkdir -p $O
i=0
for set in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $O/pmc$i -- build/tools/probe_regacc > $O/pmc$i.log 2>&1 || { tail -5 $O/pmc$i.log; exit 1; }
  echo "== $set" >> $O/summary.txt
  python3 tools/pmc_summary.py $O/pmc$i regacc_kernel >> $O/summary.txt 2>&1 || true
done
cat $O/summary.txt

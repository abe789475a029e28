#!/usr/bin/env bash
# tools/env_sweeps.sh — the whole GPU suite WITH its wall-clock expectations asserted (-m "gpu or gpu_perf": tests/conftest.py perf_expect) under the alternative paths the environment selects, and the fuzz tests on other seed bases
set -u
O=gpurun_out/${1:-r6s}; mkdir -p $O
: > $O/env_sweeps.txt
for e in "SPMV_PANEL_TRIAL=0" "SPMV_TP_PLACEMENT_BUDGET_MB=0" "SPMV_TP_PLACEMENT_BUDGET_MB=3072" "SPMV_TP_PAD=2" "SPMV_TP_PAD=16" "SPMV_CG_THREE_LAUNCHES=1" "SPMV_COMM=peer" "SPMV_HOST_STORES=0" "SPMV_COMPAT_PARTITION=nnz"; do
  env $e timeout -k 10 600 python -m pytest tests -x -q -m "gpu or gpu_perf" > $O/env_last.txt 2>&1  # (one pytest process per environment, one after the other)
  r=$(tail -1 $O/env_last.txt)
  echo "$e: $r" | tee -a $O/env_sweeps.txt
  case "$r" in *failed*) grep -n "^E \|^>\|^FAILED" $O/env_last.txt | head -20 | tee -a $O/env_sweeps.txt;; esac
done
: > $O/fuzz_bases.txt
for b in 1000 2000 3000 4000 5000 6000; do
  r=$(SPMV_FUZZ_BASE=$b timeout -k 10 600 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -1)
  echo "SPMV_FUZZ_BASE=$b: $r" | tee -a $O/fuzz_bases.txt
done

#!/usr/bin/env bash
# tools/pmc_vector_kernel.sh [TAG] - HBM-side traffic of the kernel BASELINE configs[1] names (csr_vector_kernel, forced with
# spmv_mat_set_kernel(VECTOR)) on the C2 matrix: separate rocprofv3 --pmc passes over `bench.py --pmc-child --kernel 1` (one
# warm-up + three products), as MI355X_MICROARCH.md prescribes (FETCH_SIZE and WRITE_SIZE do not fit one pass; FETCH_SIZE is
# doubled on gfx950).  The summary goes to gpurun_out/TAG/; the constants are stamped into profiles/pmc_traffic.json by hand.
set -u
TAG=${1:-r5p}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d "$O/pmc$i" -- python3 "$R/bench.py" --pmc-child --kernel 1 > "$O/pmc$i.log" 2>&1 || { echo "pass $i ($set) failed"; tail -5 "$O/pmc$i.log"; exit 1; }
  echo "== pass $i: $set" >> "$O/pmc_vector_summary.txt"
  python3 "$R/tools/pmc_summary.py" "$O/pmc$i" csr_vector_kernel >> "$O/pmc_vector_summary.txt" 2>&1
done
cat "$O/pmc_vector_summary.txt"

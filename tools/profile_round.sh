#!/usr/bin/env bash
# tools/profile_round.sh TAG — collect the evidence of one round on the GPU box into gpurun_out/TAG/
# (bench line, rocprofv3 kernel stats, PMC passes in separate runs, per-config sweeps).  Copy what is to be judged
# into profiles/ afterwards.  Run from anywhere; every GPU step has its own timeout.
set -u
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd "$R" || exit 1
timeout -k 10 300 python3 bench.py > "$O/bench_c2_uniform.json" 2> "$O/bench_c2_uniform.err" || exit 1
timeout -k 10 300 python3 bench.py --band 65536 --no-cpu-baseline > "$O/bench_c2_band65536.json" 2>/dev/null || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- python3 "$R/bench.py" --steps 50 --warmup 5 --no-cpu-baseline > "$O/rocprof_stats.log" 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/pmc_fetch" -- python3 "$R/bench.py" --steps 5 --warmup 1 --no-cpu-baseline > "$O/pmc_fetch.log" 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/pmc_write" -- python3 "$R/bench.py" --steps 5 --warmup 1 --no-cpu-baseline > "$O/pmc_write.log" 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --output-format csv -d "$O/pmc_tcc" -- python3 "$R/bench.py" --steps 5 --warmup 1 --no-cpu-baseline > "$O/pmc_tcc.log" 2>&1 || exit 1
cd "$R" || exit 1
python3 - "$O" <<'PY'
import csv, glob, statistics, sys
O = sys.argv[1]
f = glob.glob(O + "/stats/*/*_kernel_trace.csv")[0]
# the product launches; the build-time trials run the same code under the name csr_panel_kernel<..., true>
rows = [r for r in csv.DictReader(open(f)) if "csr_panel_kernel" in r["Kernel_Name"] and ", true>" not in r["Kernel_Name"]]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
name = rows[-1]["Kernel_Name"].split("(anonymous namespace)::")[-1].split("(")[0]
st = [r for r in csv.DictReader(open(glob.glob(O + "/stats/*/*_kernel_stats.csv")[0])) if name in r["Name"]]
with open(O + "/timed_region.txt", "w") as out:
    out.write(f"{name}: {len(d)} dispatches in the kernel trace (warm-up + timed steps); mean {statistics.mean(d):.4f} ms, "
              f"median {statistics.median(d):.4f}, min {min(d):.4f}, max {max(d):.4f}; the last 50 (the timed region): mean "
              f"{statistics.mean(d[-50:]):.4f} ms.  kernel_stats.csv row of the same kernel: Calls {st[0]['Calls']}, "
              f"AverageNs {float(st[0]['AverageNs']):.0f}\n")
print(open(O + "/timed_region.txt").read())
PY
for what in ell coo dia blas1; do timeout -k 10 200 python3 tools/tune.py $what --rounds 3 > "$O/tune_$what.log" 2>&1 || exit 1; done
timeout -k 10 300 python3 tools/tune.py csr --rounds 3 --reps 10 --panel "0,0,-1,-1,0,3;8,0,-1,1,0,3;8,0,0,2,0,3;8,0,-1,1,0,0" > "$O/tune_csr_uniform.log" 2>&1 || exit 1
timeout -k 10 300 python3 tools/tune.py csr --band 4096 --rounds 3 --reps 10 --panel "0,0,-1,-1,0,3;8,0,-1,1,0,0" > "$O/tune_csr_band4096.log" 2>&1 || exit 1
for d in pmc_fetch pmc_write pmc_tcc; do python3 tools/pmc_summary.py "$O/$d" csr_panel | tail -3; done
tail -n 3 "$O"/tune_*.log

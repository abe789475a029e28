#!/usr/bin/env bash
# tools/profile_round.sh TAG — collect the evidence of one round on the GPU box into gpurun_out/TAG/
# (bench line with extras and CPU modes, rocprofv3 kernel stats of the same command, PMC passes in separate runs).
# Copy what is to be judged into profiles/ afterwards.  Run from anywhere; every GPU step has its own timeout.
set -u
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd "$R" || exit 1
timeout -k 10 600 python3 bench.py > "$O/bench_c2_uniform.json" 2> "$O/bench_c2_uniform.err" || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- python3 "$R/bench.py" --steps 50 --warmup 5 --no-cpu-baseline --no-live-counters > "$O/rocprof_stats.log" 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/pmc_fetch" -- python3 "$R/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --no-live-counters > "$O/pmc_fetch.log" 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/pmc_write" -- python3 "$R/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --no-live-counters > "$O/pmc_write.log" 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --output-format csv -d "$O/pmc_tcc" -- python3 "$R/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --no-live-counters > "$O/pmc_tcc.log" 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_BUSY_sum --output-format csv -d "$O/pmc_req" -- python3 "$R/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --no-live-counters > "$O/pmc_req.log" 2>&1 || exit 1
cd "$R" || exit 1
python3 - "$O" <<'PY'
import csv, glob, re, statistics, sys
O = sys.argv[1]
trace = list(csv.DictReader(open(glob.glob(O + "/stats/*/*_kernel_trace.csv")[0])))
stats = list(csv.DictReader(open(glob.glob(O + "/stats/*/*_kernel_stats.csv")[0])))
def short(n):
    return n.split("(anonymous namespace)::", 1)[-1].split("(")[0]
def is_trial(n):  # csr_panel_kernel<U, LAYOUT, PIPE, TRIAL, TRACE, SYNCT>
    m = re.search(r"csr_panel(?:_pp)?_kernel<([^>]*)>", n)
    return bool(m) and m.group(1).split(",")[3].strip() == "true"
rows = sorted(trace, key=lambda r: int(r["Start_Timestamp"]))
# (an ELL handle running from its DIA-order copy launches two kernels per product: dia_kernel and, for the few rows that are not
# pure diagonals, ell_rows_list_kernel; the side kernel is reported by itself and left out of the runs)
side = [r for r in rows if "ell_rows_list_kernel" in r["Kernel_Name"]]
rows_all, rows = rows, [r for r in rows if "ell_rows_list_kernel" not in r["Kernel_Name"]]
# runs of consecutive dispatches of one kernel: bench.py launches every workload's product 5 + 50 times back to back
runs, cur = [], []
for r in rows:
    if cur and r["Kernel_Name"] != cur[0]["Kernel_Name"]:
        runs.append(cur)
        cur = []
    cur.append(r)
if cur:
    runs.append(cur)
with open(O + "/timed_region.txt", "w") as out:
    for run in runs:
        name = run[0]["Kernel_Name"]
        if len(run) < 50 or is_trial(name) or not any(k in name for k in ("csr_panel_kernel", "csr_panel_pp_kernel", "ell_kernel", "ell_diag_kernel", "coo_segscan_kernel", "coo_segscan_bins_kernel", "dia_kernel")):
            continue
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in run]
        tail = d[-50:]
        st = [r for r in stats if r["Name"] == name]
        out.write(f"{short(name)}: run of {len(d)} consecutive dispatches; mean {statistics.mean(d):.4f} ms, median {statistics.median(d):.4f}, "
                  f"min {min(d):.4f}, max {max(d):.4f}; the last 50: mean {statistics.mean(tail):.4f} ms.  (kernel_stats.csv row of this name, all its "
                  f"dispatches in the process: Calls {st[0]['Calls'] if st else '?'}, AverageNs {float(st[0]['AverageNs']) if st else 0:.0f})\n")
if side:
    with open(O + "/timed_region.txt", "a") as out:
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in side]
        out.write(f"ell_rows_list_kernel (the side kernel of the DIA-order product, one launch behind every dia_kernel above): {len(d)} dispatches, mean {statistics.mean(d):.4f} ms\n")
with open(O + "/timed_region.txt", "a") as out:
    # two-phase: the phases alternate (no runs); piece searches launch 3 products per configuration, each C5-shard extra 5 + 50.
    # Since round 5 small matrices run two-phase products too (the skewed extra's shards, trial candidates): only dispatches of
    # the C5 shard's size count here (expand > 0.9 ms, reduce > 0.4 ms).
    for key, floor in (("tp_expand_kernel", 0.9), ("tp_reduce_kernel", 0.4)):
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows_all if key in r["Kernel_Name"]]
        d = [t for t in d if t > floor]
        if d:
            out.write(f"{key}: {len(d)} dispatches of the C5 shard's size in the process (piece searches: 3 products per configuration; then 5 + 50 products per "
                      f"C5-shard extra); the last 55 (the last extra): mean {statistics.mean(d[-55:]):.4f} ms, median {statistics.median(d[-55:]):.4f}; all: median {statistics.median(d):.4f}\n")
print(open(O + "/timed_region.txt").read())
PY
# product launches only (the trial launches carry `true` as their fourth template argument); per workload: 1 warm-up + 5
for d in pmc_fetch pmc_write pmc_tcc pmc_req; do echo "== $d"; python3 tools/pmc_summary.py "$O/$d" csr_panel --runs | grep -v ", true," ; python3 tools/pmc_summary.py "$O/$d" ell_ --runs; python3 tools/pmc_summary.py "$O/$d" dia_kernel --runs; python3 tools/pmc_summary.py "$O/$d" coo_segscan --runs; python3 tools/pmc_summary.py "$O/$d" tp_ | tail -4; done > "$O/pmc_summary.txt" 2>&1
cat "$O/pmc_summary.txt"

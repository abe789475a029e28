set -u
O=gpurun_out/r4i; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_solver.py tests/test_gpu_fuzz.py -x -q -m gpu -k "cg or solver or spd" > $O/pytest_solver.txt 2>&1; tail -3 $O/pytest_solver.txt
for n in 100 512 2048; do
  for mode in 0 1; do
    echo "## n=$n SPMV_CG_THREE_LAUNCHES=$mode"
    SPMV_CG_THREE_LAUNCHES=$mode timeout -k 10 200 python tools/tune.py cg --n $n 2>&1 | grep -E "every= 50|Laplacian"
  done
done > $O/tune_cg_ab.txt 2>&1
cat $O/tune_cg_ab.txt

set -u
O=gpurun_out/r4x; mkdir -p $O; : > $O/fuzz_bases.txt
for b in 7000 8000 9000 10000 11000 12000 13000 14000 15000 16000 17000 18000 19000 20000 21000 22000 23000 24000 25000 26000; do
  r=$(SPMV_FUZZ_BASE=$b timeout -k 10 600 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -1)
  echo "SPMV_FUZZ_BASE=$b: $r" | tee -a $O/fuzz_bases.txt
done

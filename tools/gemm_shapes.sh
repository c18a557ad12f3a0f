# A/B of the GEMM tile kernels on the production shapes: bash tools/gemm_shapes.sh  (env passes through)
for s in "8192 8192 8192" "4096 4096 4096" "2249 37888 3584" "2249 4608 3584" "2249 3584 3584" "2249 3584 18944" "4900 3840 1280" "4900 1280 1280" "4900 5120 1280" "4900 1280 5120" "19600 3840 1280" "19600 5120 1280" "19600 1280 5120"; do
  timeout -k 10 120 python tools/gemm_bench.py $s 10 2>&1 | grep "^gemm\|Error\|mismatch" || echo "FAILED $s"
done

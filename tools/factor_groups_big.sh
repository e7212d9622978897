mkdir -p gpurun_out; rm -f gpurun_out/fb_big.txt
for w in 4096 8192 16384 32768; do
  echo "want $w" >> gpurun_out/fb_big.txt
  GPRC_FACTOR=$w timeout -k 10 300 python tools/factor_bench.py 49152 65536 >> gpurun_out/fb_big.txt 2>&1 || exit 1
done
grep -v amdgpu gpurun_out/fb_big.txt

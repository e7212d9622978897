mkdir -p gpurun_out; rm -f gpurun_out/fb_groups2.txt
for w in 8192 12000 16000 24000; do
  echo "want $w" >> gpurun_out/fb_groups2.txt
  GPRC_FACTOR=$w timeout -k 10 300 python tools/factor_bench.py 20480 24576 32768 40960 >> gpurun_out/fb_groups2.txt 2>&1 || exit 1
done
grep -v amdgpu gpurun_out/fb_groups2.txt

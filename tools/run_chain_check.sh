mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_device_level.py tests/test_gpu_parity.py -x -q > gpurun_out/dl.log 2>&1; tail -3 gpurun_out/dl.log
grep -q " passed" gpurun_out/dl.log && ! grep -q failed gpurun_out/dl.log || exit 1
for w in default 1000 2000 3000 4000 6000; do
  echo "want $w" >> gpurun_out/fb_groups.txt
  if [ $w = default ]; then timeout -k 10 200 python tools/factor_bench.py 12288 16384 24576 32768 >> gpurun_out/fb_groups.txt 2>&1 || exit 1
  else GPRC_FACTOR=$w timeout -k 10 200 python tools/factor_bench.py 12288 16384 24576 32768 >> gpurun_out/fb_groups.txt 2>&1 || exit 1; fi
done
grep -v amdgpu gpurun_out/fb_groups.txt

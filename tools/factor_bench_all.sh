mkdir -p gpurun_out
echo "service:" > gpurun_out/fb.txt
timeout -k 10 120 python tools/factor_bench.py 8192 12288 16384 20480 24576 >> gpurun_out/fb.txt 2>&1 &&
echo "left-looking grouped, fused launch per panel:" >> gpurun_out/fb.txt &&
GPRC_SERVICE=0 timeout -k 10 120 python tools/factor_bench.py 8192 12288 16384 20480 24576 >> gpurun_out/fb.txt 2>&1 &&
echo "right-looking, fused launch per panel:" >> gpurun_out/fb.txt &&
GPRC_SERVICE=0 GPRC_FACTOR=right timeout -k 10 120 python tools/factor_bench.py 8192 12288 16384 20480 24576 >> gpurun_out/fb.txt 2>&1
cat gpurun_out/fb.txt

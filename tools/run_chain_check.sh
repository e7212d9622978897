mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_device_level.py tests/test_gpu_parity.py -x -q > gpurun_out/dl.log 2>&1; tail -2 gpurun_out/dl.log
GPRC_PANEL_TRACE=0 GPRC_SERVICE=0 timeout -k 10 120 python tools/panel_trace.py 8192 2>&1 | tail -1
timeout -k 10 120 python tools/factor_bench.py 8192 16384 > gpurun_out/fb_now.txt 2>&1; grep factor_all gpurun_out/fb_now.txt
timeout -k 10 120 python bench.py --workload c2 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/c2.json 2>gpurun_out/c2.err; python -c "
import json
d=json.load(open('gpurun_out/c2.json')); print(d['ms_per_step'], d['phases_ms']['fit_F1_F3'], d.get('parity_timed_config_normwise_err'))"
timeout -k 10 200 python tools/service_trace.py 8192 2>&1 | tail -1

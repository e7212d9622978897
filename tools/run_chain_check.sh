mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_device_level.py -x -q > gpurun_out/dl.log 2>&1; tail -2 gpurun_out/dl.log
timeout -k 10 300 python tools/trsv_bench.py 8192 16384 65536 > gpurun_out/trsv.txt 2>&1 ; grep forward gpurun_out/trsv.txt
timeout -k 10 120 python bench.py --workload c2 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/c2.json 2>gpurun_out/c2.err; python -c "
import json
d=json.load(open('gpurun_out/c2.json')); print(d['ms_per_step'], d['phases_ms']['fit_F1_F3'], d.get('parity_timed_config_normwise_err'), d['kernels']['trsv'])"
timeout -k 10 200 python tools/bench_gpc.py > gpurun_out/c5.json 2>gpurun_out/c5.err; python -c "
import json
d=json.load(open('gpurun_out/c5.json')); print(d['ms_per_irls_iteration'], d['kernels']['trsv'])"

# On the GPU box: what happens where kernel dispatches are serialised (rocprofv3 --pmc): the fit entry points and bench.py fall back to one
# launch per panel by themselves (gprc_factor_service); logs under gpurun_out/pmc_test/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/pmc_test
timeout -k 10 300 python -m pytest tests/test_gpu_device_level.py tests/test_gpu_parity.py -x -q > gpurun_out/dl.log 2>&1; tail -2 gpurun_out/dl.log
(time timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_test/a -- python3 bench.py --workload c2 --steps 1 --warmup 0 --no-cpu-baseline) > gpurun_out/pmc_test/bench_c2_pmc.log 2>&1
grep -E "^\{|Error|real" gpurun_out/pmc_test/bench_c2_pmc.log | cut -c1-260
cat > /tmp/gpr_small.py <<'PY'
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np, time
import gprc_amd
from gprc_amd import GPR, cov_func, sqrexp
rng = np.random.default_rng(0); X = rng.uniform(-1, 1, (3, 3000)); y = rng.normal(size=3000)
t0 = time.time(); g = GPR(X, y, 0.1, cov_func(sqrexp, l=0.7)); print("fit 1", round(time.time() - t0, 2), "s logp", g.logp)
t0 = time.time(); g2 = GPR(X, y, 0.1, cov_func(sqrexp, l=0.7)); print("fit 2", round(time.time() - t0, 2), "s logp", g2.logp)
PY
(time timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_test/b -- python3 /tmp/gpr_small.py) > gpurun_out/pmc_test/gpr_pmc.log 2>&1
grep -E "^fit|Error|real" gpurun_out/pmc_test/gpr_pmc.log
python3 /tmp/gpr_small.py 2>&1 | grep fit
rm -rf gpurun_out/pmc_test/a gpurun_out/pmc_test/b

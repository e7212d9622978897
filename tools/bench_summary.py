"""Print a one-line digest of bench.py JSON logs: python tools/bench_summary.py LOG..."""
import json, sys
for path in sys.argv[1:]:
    l = [x for x in open(path) if x.startswith("{")]
    if not l:
        print(path, "NO JSON:", open(path).read()[-800:]); continue
    j = json.loads(l[-1]); k = j["kernels"]
    names = ("trailing_update", "solve_update_k512", "gemm_inner_k128", "trsm_panel", "potf2_inv", "trsv", "fill", "row_reduce")
    print(path.split("/")[-1], f'{j["ms_per_step"]:.1f} ms/step', f'{j["value"]:.2f} TF |',
          " ".join(f'{n.split("_")[0]}={k[n]["tflops"]}TF/{k[n]["ms_total"]:.0f}ms' for n in names if n in k))

timeout 1200 python -m pytest tests/test_pvgo_gpu.py tests/test_solver_stress_gpu.py tests/test_dist_c_gpu.py tests/test_dist_gpu.py tests/test_reproj_gpu.py tests/test_configs_gpu.py tests/test_surface_gpu.py -x -q -m gpu 2>&1 | tail -5
python bench.py --no-frontend --no-cpu-baseline 2>/dev/null > gpurun_out/bench_dpp.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_dpp.json').read().strip().splitlines()[-1])
print(d["value"], d["us_per_lm_iter"], d["roofline"]["frac"], d["roofline"]["solve_launch_us"], d["reject_heavy"]["value"], d["large_graph"]["value"])
PY

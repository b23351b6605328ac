HAMSPINE_TOWER_EXEC=0 python -m pytest tests/test_product_gpu.py tests/test_edge_cases_gpu.py -q -x > gpurun_out/a_tests.log 2>&1; echo tests_notower=$?; tail -2 gpurun_out/a_tests.log
HAMSPINE_TOWER_EXEC=0 python bench.py --no-f32 --no-cpu-baseline > gpurun_out/a_bench.json 2> gpurun_out/a_bench.err; echo "c2 per-block path: $(grep 'steps in' gpurun_out/a_bench.err | cut -c1-150)"
HAMSPINE_TOWER_EXEC=0 python bench.py --workload c3 --no-f32 --no-cpu-baseline > gpurun_out/a_bench.json 2> gpurun_out/a_bench.err; echo "c3 per-block path: $(grep 'steps in' gpurun_out/a_bench.err | cut -c1-150)"

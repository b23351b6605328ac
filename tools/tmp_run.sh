python -m pytest tests/test_convnext_gpu.py tests/test_product_gpu.py tests/test_tower_gpu.py -q -x > gpurun_out/a_tests.log 2>&1; echo tests=$?; tail -3 gpurun_out/a_tests.log
python bench.py --workload c4 --no-f32 --no-cpu-baseline --gemm-log gpurun_out/launches_c4.csv > gpurun_out/a_bench.json 2> gpurun_out/a_bench.err; echo "c4 nt: $(grep 'steps in' gpurun_out/a_bench.err | cut -c1-80)"
python tools/launch_table.py gpurun_out/launches_c4.csv > gpurun_out/launch_table_c4.txt
HAMSPINE_WGRAD_NT=0 python bench.py --workload c4 --no-f32 --no-cpu-baseline > gpurun_out/a2_bench.json 2> gpurun_out/a2_bench.err; echo "c4 tn: $(grep 'steps in' gpurun_out/a2_bench.err | cut -c1-80)"

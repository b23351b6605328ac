python -m pytest tests/test_gemm_gpu.py tests/test_tower_gpu.py tests/test_product_gpu.py -q -x > gpurun_out/a_tests.log 2>&1; echo tests=$?; tail -3 gpurun_out/a_tests.log
python bench.py --no-f32 --no-cpu-baseline --gemm-log gpurun_out/launches_a.csv > gpurun_out/a_bench.json 2> gpurun_out/a_bench.err; grep "steps in" gpurun_out/a_bench.err | cut -c1-300
python tools/launch_table.py gpurun_out/launches_a.csv > gpurun_out/launch_table_a.txt
HAMSPINE_PERSISTENT=0 python bench.py --no-f32 --no-cpu-baseline --gemm-log gpurun_out/launches_a2.csv > gpurun_out/a2_bench.json 2> gpurun_out/a2_bench.err; grep "steps in" gpurun_out/a2_bench.err | cut -c1-300
python tools/launch_table.py gpurun_out/launches_a2.csv > gpurun_out/launch_table_a2.txt

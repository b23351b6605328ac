python -m pytest tests/test_tower_gpu.py tests/test_product_gpu.py -q -x > gpurun_out/a_tests.log 2>&1; echo tests=$?; tail -3 gpurun_out/a_tests.log
for i in 1 2; do
python bench.py --no-f32 --no-cpu-baseline > gpurun_out/a_bench.json 2> gpurun_out/a_bench.err; echo "dgrad nt: $(grep 'steps in' gpurun_out/a_bench.err | cut -c1-80)"
HAMSPINE_DGRAD_NT=0 python bench.py --no-f32 --no-cpu-baseline > gpurun_out/a2_bench.json 2> gpurun_out/a2_bench.err; echo "dgrad nn: $(grep 'steps in' gpurun_out/a2_bench.err | cut -c1-80)"
done

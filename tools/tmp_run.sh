python -m pytest tests/test_convnext_gpu.py tests/test_product_gpu.py tests/test_tower_gpu.py -q -x > gpurun_out/a_tests.log 2>&1; echo tests=$?; tail -3 gpurun_out/a_tests.log
python bench.py --workload c4 --no-f32 --no-cpu-baseline > gpurun_out/a_bench.json 2> gpurun_out/a_bench.err; echo "c4: $(grep 'steps in' gpurun_out/a_bench.err | cut -c1-80)"
python bench.py --no-f32 --no-cpu-baseline > gpurun_out/a_bench2.json 2> gpurun_out/a_bench2.err; echo "c2: $(grep 'steps in' gpurun_out/a_bench2.err | cut -c1-80)"
bash tools/iso_profile.sh c4 > gpurun_out/isoq_c4.txt 2>&1

python -m pytest tests/test_convnext_gpu.py -q -x > gpurun_out/a_tests.log 2>&1; echo tests=$?; tail -3 gpurun_out/a_tests.log
python bench.py --workload c4 --no-cpu-baseline > gpurun_out/a_bench.json 2> gpurun_out/a_bench.err; echo "c4: $(grep 'steps in' gpurun_out/a_bench.err | cut -c1-80)"; grep -o '"f32_mode": {[^}]*}' gpurun_out/a_bench.json | cut -c1-120

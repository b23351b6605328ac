python -m pytest tests -q -m gpu > gpurun_out/r2_gpu_suite.log 2>&1; echo suite=$?; tail -4 gpurun_out/r2_gpu_suite.log
python bench.py > gpurun_out/r2_bench_c2.json 2> gpurun_out/r2_bench_c2.err; grep "steps in" gpurun_out/r2_bench_c2.err | cut -c1-200
HAMSPINE_TOWER_OVERLAP=0 python bench.py --no-f32 --no-cpu-baseline > gpurun_out/r2_bench_c2_serial.json 2> gpurun_out/r2_bench_c2_serial.err; grep "steps in" gpurun_out/r2_bench_c2_serial.err | cut -c1-200

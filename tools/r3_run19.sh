#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/t19.txt 2>&1; tail -6 gpurun_out/t19.txt

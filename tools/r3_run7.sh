#!/bin/bash
set -o pipefail
timeout -k 10 900 python -m pytest tests/test_train_loop_gpu.py -x -q 2>&1 | tail -12

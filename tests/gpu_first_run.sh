#!/bin/bash
# first GPU contact: parity tests, then a rough timing of NTT and MSM at BASELINE sizes
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tee gpurun_out/pytest_gpu.log

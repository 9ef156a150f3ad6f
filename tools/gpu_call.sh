#!/bin/bash
# one gpurun call: $1 = tag; runs the steps listed in tools/gpu_steps_$1.sh with output under gpurun_out/$1/
set -o pipefail
tag=$1
mkdir -p gpurun_out/$tag
export TMPDIR=/tmp
bash tools/gpu_steps_$tag.sh gpurun_out/$tag

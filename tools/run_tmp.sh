#!/bin/bash
set -e -o pipefail
export DGNN_PROFILE_TCC=1
bash tools/profile.sh r2h edgeconv tools/profile_ops.py edgeconv

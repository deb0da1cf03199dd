#!/bin/bash
# final kernel-stats + HBM-byte profiles of the two GAT bench workloads and the GMM / DGCNN lines
set -u
root=${GRAFT_REPO_ROOT:-$PWD}; cd $root
DGNN_PROFILE_SQ=1 bash tools/profile.sh r3n products || exit 1
bash tools/profile.sh r3n reddit || exit 1
DGNN_PROFILE_TCC=0 bash tools/profile.sh r3n gmm-reddit || exit 1
DGNN_PROFILE_TCC=0 bash tools/profile.sh r3n dgcnn || exit 1
echo done

#!/bin/bash
# The example models with the reference's run matrix (test/gat_result.sh, gmm_result.sh,
# edgeconv_result.sh) on synthetic data of the same shapes; one CSV line each.
#   gpurun -- bash tools/run_examples.sh <tag>      -> gpurun_out/examples_<tag>.csv
set -u
tag=${1:-r1}; root=${GRAFT_REPO_ROOT:-$PWD}; out=$root/gpurun_out/examples_$tag.csv; log=$root/gpurun_out/examples_$tag.log
rm -f $out; : > $log
run() { timeout -k 10 300 python3 "$@" --output $out >> $log 2>&1 || { echo "FAILED: $*"; exit 1; }; }
cd $root/examples
for d in cora pubmed; do run train_gatconv.py --dataset $d --n-heads 4 --n-hidden 64 --n-epochs 200; done
run train_gatconv.py --dataset reddit --n-heads 4 --n-hidden 64 --n-epochs 20
run train_gatconv.py --dataset reddit --n-heads 1 --n-hidden 64 --n-epochs 20
for d in cora pubmed; do run train_gmmconv.py --dataset $d --n-hidden 64 --n-epochs 200; done
run train_gmmconv.py --dataset reddit --n-hidden 64 --n-kernels 3 --n-epochs 20
run train_gmmconv.py --dataset reddit --n-hidden 64 --n-kernels 2 --n-epochs 20
run train_edgeconv.py --batch-size 32 --num-points 2048 --k 20 --n-epochs 20
run train_edgeconv.py --batch-size 64 --num-points 1024 --k 40 --n-epochs 20
run train_gatconv.py --dataset products --n-heads 4 --n-hidden 64 --n-epochs 20
cat $out

#!/bin/bash
# On the GPU box (scratch snapshot): A/B of one source-level constant.  usage: ab_build.sh <file> <sed-expr-A> <sed-expr-B> -- cmd...
# Builds variant A, runs cmd, builds variant B, runs cmd, then A and B again (interleaved rounds).
set -e
f=$1; ea=$2; eb=$3; shift 4
cp $f /tmp/ab_orig
for round in 1 2; do
  for v in A B; do
    cp /tmp/ab_orig $f
    if [ $v = A ]; then sed -i "$ea" $f; else sed -i "$eb" $f; fi
    make -C lp_amd/csrc -j16 > /dev/null 2>&1
    echo "== variant $v (round $round)"
    "$@"
  done
done

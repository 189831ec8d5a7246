#!/bin/bash
# k_short<32> at dim 1, 2, 3 (and Helfand, dim 3): 12 GB of input each (n_atoms = 1.5e9 / (32 dim))
cd "$(dirname "$0")"
for bp in 0 1; do
  timeout -k 5 60 ./short_test_d1 32 46874880 $bp 5 || exit 1
  timeout -k 5 60 ./short_test_d2 32 23437440 $bp 5 || exit 1
  timeout -k 5 60 ./short_test_d3 32 15624960 $bp 5 || exit 1
  timeout -k 5 60 ./short_test_h3 32 15624960 $bp 5 || exit 1
done

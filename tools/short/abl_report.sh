#!/bin/bash
# what bounds k_short<64> (VACF, D = 3): the kernel with parts compiled out (tools/short/ablations.patch), 12 GB of input
cd "$(dirname "$0")"
for bp in 0 1; do
  for T in 40 64; do
    A=$((500000000 / T / 64 * 64))
    for v in a0 a1 a2 a4 a5 a7; do
      [ -x ./short_test_$v ] && { timeout -k 5 60 ./short_test_$v $T $A $bp 5 || exit 1; }
    done
  done
done
# the same kernel with the rows of a pair further apart (pitch 72 instead of 64 rows)
timeout -k 5 60 ./short_test_a0 64 7812480 0 5 72
timeout -k 5 60 ./short_test_t32 32 15624960 0 5
timeout -k 5 60 ./short_test_t32 32 15624960 0 5 40

#!/bin/bash
cd "$(dirname "$0")"
for bp in 0 1; do for v in a0 a8; do timeout -k 5 60 ./short_test_$v 64 7812480 $bp 5 || exit 1; done; done

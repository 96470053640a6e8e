#!/bin/bash
# A/B of library builds on one box: profiles/tools/ab.sh "<bench args>" ...; libs = default + profiles/ab/*.so
for args in "$@"; do
  for lib in default profiles/ab/*.so; do
    if [ "$lib" = default ]; then unset PDS_LIB; else export PDS_LIB=$PWD/$lib; fi
    for rep in 1 2; do
      python bench.py $args --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('$args', '$lib', round(d['ms_per_step']*1000, 2), 'us', round(d['roofline']['frac']*100, 1), '%')"
    done
  done
done

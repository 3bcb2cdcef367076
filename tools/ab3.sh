#!/bin/bash
# usage: tools/ab3.sh <out.txt> <repeats> "<lib or ->:<options...>" ...  -- like ab.sh, every variant `repeats` times, interleaved; prints the median
out=$1; rep=$2; shift; shift
: > $out.raw
for r in $(seq 1 $rep); do
  for v in "$@"; do
    lib=${v%%:*}; opts=${v#*:}
    if [ "$lib" = "-" ]; then unset CRT_LIB; else export CRT_LIB=$PWD/$lib; fi
    t=$(timeout -k 10 120 python tools/util_dbg.py 64 5 $opts 2>&1 | grep "ms per call" | awk '{print $4}')
    echo "$lib [$opts] $t" >> $out.raw
  done
done
python3 - $out.raw > $out <<'PY'
import sys, collections, statistics
d = collections.OrderedDict()
for line in open(sys.argv[1]):
    k, _, v = line.rstrip().rpartition(' ')
    try: d.setdefault(k, []).append(float(v))
    except ValueError: d.setdefault(k, [])
for k, v in d.items():
    print('%-44s median %.2f  (%s)' % (k, statistics.median(v) if v else float('nan'), ' '.join('%.2f' % x for x in v)))
PY
cat $out

#!/bin/bash
# usage: tools/ab.sh <out.txt> "<lib or ->:<options...>" ...   -- ms per 64-spp call (S2, 1080p, pipelined) per variant
out=$1; shift
: > $out
for v in "$@"; do
  lib=${v%%:*}; opts=${v#*:}
  if [ "$lib" = "-" ]; then unset CRT_LIB; else export CRT_LIB=$PWD/$lib; fi
  r=$(timeout -k 10 120 python tools/util_dbg.py 64 5 $opts 2>&1 | grep "ms per call")
  echo "$lib [$opts] $r" >> $out
done
cat $out

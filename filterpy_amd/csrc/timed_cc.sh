#!/bin/bash
# timed_cc.sh <log> <compiler> <args...>: runs the compile and appends "<seconds> <object>" to <log> (csrc/Makefile: CC1).
# __graft_entry__.build() prints the slowest objects, so that a build that ran out of time names its culprit.
log=$1; shift
t0=$(date +%s%N)
"$@"
rc=$?
t1=$(date +%s%N)
out=""
while [ $# -gt 0 ]; do if [ "$1" = "-o" ]; then out=$2; fi; shift; done
printf "%d.%03d %s\n" $(( (t1 - t0) / 1000000000 )) $(( ((t1 - t0) / 1000000) % 1000 )) "$out" >> "$log"
exit $rc

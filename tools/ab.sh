#!/bin/bash
# run on the GPU box: A/B of two builds of the library over several PROCESSES each (the time of the same binary moves by +-8 % from one
# process to the next with the physical placement of its buffers: single runs do not compare builds).  usage: ab.sh <other.so> <reps> <probe args...>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B=$1; N=$2; shift 2
cp aquery2_amd/libaqg.so /tmp/a.so; cp $B /tmp/b.so
for i in $(seq $N); do
  cp /tmp/a.so aquery2_amd/libaqg.so; echo "A $(timeout -k 10 100 python3 tools/q5_probe.py "$@" | grep rep2 | tr '\n' ' ')"
  cp /tmp/b.so aquery2_amd/libaqg.so; echo "B $(timeout -k 10 100 python3 tools/q5_probe.py "$@" | grep rep2 | tr '\n' ' ')"
done
cp /tmp/a.so aquery2_amd/libaqg.so

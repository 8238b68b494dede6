#!/bin/bash
# one line: average duration of the kernels matching $2 in a short rocprofv3 run of bench.py (tag $1, further args passed on)
tag=$1; pat=$2; shift 2
bash tools/kstats.sh $tag "$@" | grep "$pat" | sed "s/^/$tag: /"

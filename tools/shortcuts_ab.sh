#!/bin/bash
# Runs on the GPU box: the two sweep shortcuts (no own box test for untransformed swept mesh instances; the bounding light's distance
# taken from the light pre-pass) on and off, same box, C2.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
line() { tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']
print('$1: %.1f Ms/s' % d['value'], {n: round(v['avg_us']) for n,v in k.items()})"; }
A="--steps 3 --warmup 1 --cpu-seconds 0 --spp-per-step 480"
python bench.py $A 2>/dev/null | line "both shortcuts"
PT_AMD_NO_KNOWN_LIGHT=1 python bench.py $A 2>/dev/null | line "light tested again"
PT_AMD_OWN_TESTS=1 python bench.py $A 2>/dev/null | line "own box tests"
PT_AMD_NO_KNOWN_LIGHT=1 PT_AMD_OWN_TESTS=1 python bench.py $A 2>/dev/null | line "neither (the kernels before)"
python bench.py $A 2>/dev/null | line "both shortcuts (again)"

#!/bin/bash
# Runs on the GPU box: A/B of prebuilt engine libraries (variants/*.so, built here with tools/build_variant.sh so that no GPU time goes
# into compiling) on chosen bench workloads, alternating the libraries twice.
# usage: tools/ab_libs.sh "<lib A> <lib B> ..." -- <bench args...>   (several workloads: separate them with ';;'; "ENV=val" words before a lib's path set its environment: "PT_AMD_X=1:variants/a.so")
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
LIBS="$1"; shift 2
one() { timeout 300 python bench.py --steps 3 --warmup 1 --cpu-seconds 0 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']
print('   %.1f Ms/s' % d['value'], {n: round(v['avg_us']) for n,v in k.items()})"; }
for round in 1 2; do for spec in $LIBS; do
  lib=${spec##*:}; envs=""; [ "$spec" != "$lib" ] && envs=${spec%:*}
  echo "== $spec"
  args=(); for a in "$@"; do if [ "$a" == ";;" ]; then env ${envs//,/ } PT_AMD_LIBRARY=$ROOT/$lib bash -c "$(declare -f one); one ${args[*]}"; args=(); else args+=("$a"); fi; done
  env ${envs//,/ } PT_AMD_LIBRARY=$ROOT/$lib bash -c "$(declare -f one); one ${args[*]}"
done; done

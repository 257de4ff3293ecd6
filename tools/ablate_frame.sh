#!/bin/bash
# builds variants of the library with extra -D flags and times the FRAME path with each (GPU box), checking the
# results against the first variant's.  usage: tools/ablate_frame.sh "<args of variant_check.py after ref>" "<flags1>" "<flags2>" ...
cd ${GRAFT_REPO_ROOT:-/root/repo}
ARGS="$1"; shift
REF=/tmp/variant_ref_$$.pt
for f in "$@"; do
  echo "== variant: [$f]"
  TS_EXTRA_HIPCC_FLAGS="$f" python -m tinysplat_amd._build > /tmp/build.log 2>&1 || { tail -5 /tmp/build.log; continue; }
  TS_ALLOW_VARIANT_LIB=1 python tools/variant_check.py $REF $ARGS 2>&1 | tail -1
done
echo "== default rebuild"; python -m tinysplat_amd._build > /tmp/build.log 2>&1 || tail -5 /tmp/build.log

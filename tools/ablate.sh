#!/bin/bash
# builds variants of the library with extra -D flags and times each (GPU box). usage: tools/ablate.sh "<flags1>" "<flags2>" ...
cd ${GRAFT_REPO_ROOT:-/root/repo}
for f in "$@"; do
  echo "== variant: [$f]"
  TS_EXTRA_HIPCC_FLAGS="$f" python -m tinysplat_amd._build > /tmp/build.log 2>&1 || { tail -5 /tmp/build.log; continue; }
  TS_ALLOW_VARIANT_LIB=1 python tools/time_raster.py 2>&1 | tail -1
done
# leave a default library behind (variants carry a .flags stamp and _lib.load() refuses them)
echo "== default rebuild"; python -m tinysplat_amd._build > /tmp/build.log 2>&1 || tail -5 /tmp/build.log

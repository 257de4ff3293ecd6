#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for f in "$@"; do
  echo "== variant: [$f]"
  TS_EXTRA_HIPCC_FLAGS="$f" python -m tinysplat_amd._build > /tmp/build.log 2>&1 || { tail -5 /tmp/build.log; continue; }
  python tools/time_binning.py 5000000 3840 2160 2>&1 | tail -1 | cut -c1-100
done
# leave a default library behind (variants carry a .flags stamp and _lib.load() refuses them)
echo "== default rebuild"; python -m tinysplat_amd._build > /tmp/build.log 2>&1 || tail -5 /tmp/build.log

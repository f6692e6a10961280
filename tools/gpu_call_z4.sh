#!/bin/bash
export PYTHONPATH=$GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
for bf in auto 6 12; do
echo "== fill $bf"
if [ $bf = auto ]; then PGH_DEBUG=1 timeout 600 python tools/probe_partition.py --seeds --worlds 4 2>&1 | grep -E "pb:|step=" | cut -c1-260;
else PGH_PB_BINFILL=$bf PGH_DEBUG=1 timeout 600 python tools/probe_partition.py --seeds --worlds 4 2>&1 | grep -E "pb:|step=" | cut -c1-260; fi
done

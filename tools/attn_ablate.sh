#!/bin/bash
# timing ablations of the pipelined attention forward (VT_ATTN_DBG bit mask, WRONG results) at 4 and 8 waves per workgroup
R=${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
for w in 4 8; do
for d in 0 1 4 16 30; do
  echo -n "waves $w dbg $d: "; VT_ATTN_PIPE=1 VT_ATTN_WAVES=$w VT_ATTN_DBG=$d python $R/tools/attn_bench.py 20 2>/dev/null | grep fwd
done
done
echo -n "plain: "; python $R/tools/attn_bench.py 20 2>/dev/null | grep fwd
done

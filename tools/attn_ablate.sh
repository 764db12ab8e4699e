#!/bin/bash
# timing ablations of the pipelined attention forward (VT_ATTN_DBG bit mask, WRONG results): what does each ingredient cost?
R=${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
for d in 0 1 2 4 6 8 16 22 30; do
  echo -n "dbg $d: "; VT_ATTN_DBG=$d python $R/tools/attn_bench.py 20 2>/dev/null | grep fwd
done
echo -n "plain: "; VT_ATTN_PLAIN=1 python $R/tools/attn_bench.py 20 2>/dev/null | grep fwd
done

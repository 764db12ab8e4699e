#!/bin/bash
# The reproducing case of the stale scalar read inside graph replays (a loss whose scalar factors change from step to step) with the
# runtime's graph packet capture ON (explicitly; the ROCm default) and OFF (what `import video_tokenizer_amd` sets unless the
# environment already carries a value).  With it ON the GraphedStep constructor's self-check must refuse the graph.
# usage: bash tools/graph_stale_scalar_check.sh   (GPU box, repo root)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
T="tests/test_model_gpu.py::test_graphed_step_self_check_with_a_product_of_scalars_in_the_loss"
for rep in 1 2; do
echo -n "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 (runtime default): "; DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 python -m pytest $T -x -q -m gpu 2>&1 | grep -E "RuntimeError: GraphedStep|passed|failed" | head -2 | tr '\n' ' '; echo
echo -n "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 (package default): "; python -m pytest $T -x -q -m gpu 2>&1 | tail -1
done

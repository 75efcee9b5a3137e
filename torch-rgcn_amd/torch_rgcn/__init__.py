"""torch_rgcn on MI355X: the reference's layer API over hand-written gfx950 kernels."""
import os as _os

import torch as _torch

# hipGraph replays and the HIP runtime's "graph packet capture".  The runtime this image ships (ROCm 7.0.x inside PyTorch 2.10) replays
# some graph nodes -- memset nodes, which ATen's own ops emit (nll_loss backward, index_put ...) -- with STALE arguments once eager
# kernels have run between two replays: a captured training step then silently trains on garbage (measured, round 4: the MUTAG
# step replayed 0.788 -> 0.114 -> 0.041 where the eager loop and the same replay with the feature off go 0.788 -> 0.489 -> 0.424;
# tools/hipgraph_repro/ has the stand-alone reproducer from round 2).  The feature is read once, when the runtime initialises, so it is
# switched off here -- before this process makes its first HIP call -- unless the user has set the variable himself.  Cost: replays of
# the small configs get ~10 % slower (AIFB 0.40 -> 0.45 ms), eager execution is not affected.
_VAR = "DEBUG_CLR_GRAPH_PACKET_CAPTURE"
if not _torch.cuda.is_initialized():
    _os.environ.setdefault(_VAR, "0")
REPLAY_SAFE = _os.environ.get(_VAR) == "0"
"""True when captured steps may be replayed in this process (experiments capture by default only then)"""

"""MI355X-native LARP tokenizer hot path (encode -> quantize -> decode training step).

Host-side mirror of the reference's model-registry interface over the C ABI of libvt_hip.so
(include/vt_hip.h).  There is no CPU fallback: every compute entry point raises if the HIP
library is missing or the tensors are not on a GPU.
"""
import os as _os

# Graph replays (engine.GraphedStep, the AR prior's decode loop) are only correct on this ROCm build with the runtime's graph PACKET
# CAPTURE off: with it, a kernel node of a replay can read a small tensor an earlier node of the same replay wrote as the PREVIOUS
# replay left it (round 4: tools/graph_stale_scalar_check.sh, profiles/r04_graph_stale_scalar.log).  The runtime reads the
# flag once, when the process first touches HIP; an explicit setting of the user is respected.
# If the process touched HIP BEFORE this import and the flag was not set, the setting below comes too late (advisor finding, round 4):
# GRAPH_FLAG_LATE is then True, GraphedStep still verifies itself against eager steps, and larp_ar.generate keeps its eager loop.
import sys as _sys
_torch = _sys.modules.get("torch")
GRAPH_FLAG_LATE = bool("DEBUG_CLR_GRAPH_PACKET_CAPTURE" not in _os.environ and _torch is not None and _torch.cuda.is_initialized())
_os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
if GRAPH_FLAG_LATE:
    import warnings as _warnings
    _warnings.warn("video_tokenizer_amd: the HIP runtime was initialised before this package could set DEBUG_CLR_GRAPH_PACKET_CAPTURE=0; hipGraph replays "
                   "of its steps may read stale scalars on this ROCm build (DESIGN 6b).  Import video_tokenizer_amd before the first CUDA call or set "
                   "the variable in the environment; until then generation runs its eager loop and GraphedStep relies on its self-check.")


def graph_replay_safe():
    """False when the runtime's graph packet capture may be on (see above): callers that replay hipGraphs without a self-check fall back to eager."""
    return not GRAPH_FLAG_LATE and _os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE") == "0"


from . import hip  # noqa: F401  (ctypes binding; loading is lazy)
from . import config  # noqa: F401
from .registry import make, models, register  # noqa: F401
from . import transformer, bottleneck, larp_tokenizer, loss, titok, sq, larp_ar  # noqa: F401  (registers the classes)
from .larp_tokenizer import LARPTokenizer  # noqa: F401
from .larp_ar import LARP_AR  # noqa: F401
from .loss import TransformerDiscriminator, VQLPIPSWithDiscriminator  # noqa: F401
from .fsq import FSQ  # noqa: F401


def invalidate_weight_packs(module):
    """Drop every cached bf16 operand copy below `module`.  The copies are keyed by (owner identity, parameter address,
    `_version`); optimizer steps, load_state_dict and .to() change one of these, a write through `.data` (`p.data.copy_(...)`,
    e.g. a hand-rolled EMA swap) does not -- call this after such a write."""
    for m in module.modules():
        if hasattr(m, "invalidate_packs"):
            m.invalidate_packs()
        m.__dict__.pop("_vt_pack", None)                      # larp_ar._pack caches
        eng = m.__dict__.get("_engine")
        if eng is not None:
            eng.param_epoch = getattr(eng, "param_epoch", 0) + 1   # the tokenizer engine re-packs on the next forward

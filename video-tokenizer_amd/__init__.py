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
_os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")

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

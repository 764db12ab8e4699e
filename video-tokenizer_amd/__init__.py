"""MI355X-native LARP tokenizer hot path (encode -> quantize -> decode training step).

Host-side mirror of the reference's model-registry interface over the C ABI of libvt_hip.so
(include/vt_hip.h).  There is no CPU fallback: every compute entry point raises if the HIP
library is missing or the tensors are not on a GPU.
"""
from . import hip  # noqa: F401  (ctypes binding; loading is lazy)
from . import config  # noqa: F401
from .registry import make, models, register  # noqa: F401
from . import transformer, bottleneck, larp_tokenizer, loss, titok, sq, larp_ar  # noqa: F401  (registers the classes)
from .larp_tokenizer import LARPTokenizer  # noqa: F401
from .larp_ar import LARP_AR  # noqa: F401
from .loss import TransformerDiscriminator, VQLPIPSWithDiscriminator  # noqa: F401
from .fsq import FSQ  # noqa: F401

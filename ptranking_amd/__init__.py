"""ptranking_amd — MI355X-native (gfx950 / CDNA4) implementation of wildltr/ptranking's ltr_adhoc loss + metric hot path.

    functional  fused HIP losses (autograd Functions) and device metrics on [B, L] tensors
    rankers     RankNet / LambdaRank / LambdaLoss / ApproxNDCG / ListNet / ListMLE with the reference's plugin surface
    host        LABEL_TYPE, DeviceEvaluator, DeviceTrainLoop, the stand-alone pointsf base ranker
    scorer      the pointsf MLP scorer on fused fp32-MFMA kernels (FusedPointScorer) + FlatAdam
    batching    PaddedQueryBatches: device-resident padded query batches (+ lens) replacing the reference's loader stack
    dp          data-parallel gradient exchange (one RCCL all-reduce per step)
    install()   rebinds the six ranker names inside an installed ptranking so LTREvaluator uses them unchanged

The only compute implementation is the HIP library ptranking_amd/libptranking_amd.so (C ABI: include/ptranking_amd.h);
there is no CPU fallback.  Build it with `python -m ptranking_amd.build`.
"""
from . import _lib, batching, dp, functional, host, rankers, scorer   # noqa: F401
from .batching import PaddedQueryBatches            # noqa: F401
from . import letor                                   # noqa: F401
from .host import LABEL_TYPE, DeviceEvaluator       # noqa: F401
from .install import install, uninstall             # noqa: F401
from .rankers import (ApproxNDCG, LambdaLoss, LambdaRank, ListMLE, ListNet, RankNet, STListNet, RankCosine, RankMSE, SoftRank, DASALC, MDPRank,  # noqa: F401
                      DEFAULT_PARAS, EXTRA_RANKER_NAMES, RANKER_NAMES, make_ranker_classes)

__version__ = "0.1.0"

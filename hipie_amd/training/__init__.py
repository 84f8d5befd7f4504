"""Training-side host logic of the reference, device-agnostic PyTorch: matching costs + assignment, the
set-prediction losses of the two heads, contrastive de-noising query preparation.  The operator backward it goes with is
``hipie_msda_backward`` (csrc/msda_bwd.hip); functions.py holds the autograd Functions of the kernels that have a backward (mask
contraction, CondInst dynamic mask head), ddp.py the bucketed gradient all-reduce.  NOT a training step: the ViT / encoder / decoder
kernels of the inference path have no backward and there is no coco_forward, so nothing here is on a timed path.  Every function that draws random numbers in the reference takes them as an
argument (or from a ``draw`` callable) so that results can be compared with the reference bit for bit."""
from .boxes import box_cxcywh_to_xyxy, generalized_box_iou, paired_giou_loss, paired_iou      # noqa: F401
from .matcher import HungarianMatcher, MatchWeights                                            # noqa: F401
from .dn import cdn_queries, dn_split_outputs, dn_match_indices, maskdino_dn_queries           # noqa: F401
from .criterion import DetCriterion, MaskCriterion                                             # noqa: F401
from .weights import maskdino_loss_plan, weighted_merge                                        # noqa: F401
from .targets import prepare_targets, split_things_stuff                                       # noqa: F401
from .ddp import GradientBuckets                                                               # noqa: F401

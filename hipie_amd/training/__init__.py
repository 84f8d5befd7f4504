"""The training side of the path.

  step.py      ONE TRAINING STEP: the training branch of HIPIE_IMG.forward around DDETRSegmUniDN.coco_forward (de-noising queries, the three
               query groups with their matchings and DINO criterion calls, the MaskDINO branch and its criterion) as a weighted loss
               dictionary over the product model's parameters -- loss entries, total and every parameter gradient pinned against the
               reference's own step (tests/golden/train_step_tiny.npz, tests/test_training.py::test_train_step_*);
  net.py       the differentiable network under it: hand-written HIP forward AND backward for multi-scale deformable attention, the mask
               contraction, the CondInst dynamic mask head, the large linears (split-fp16 GEMM: y, dx, dW) and the attention of the
               global ViT blocks (csrc/attn_train.hip: no (heads, N, N) tensor in HBM) -- functions.py, ../msda_shim.py; everything
               else on the library kernels with torch.autograd;
  matcher.py, criterion.py, dn.py, targets.py, weights.py, boxes.py
               the host logic: matching costs + assignment (Hungarian, SimOTA), the set-prediction losses of both heads, contrastive
               de-noising queries, target preparation, loss weighting -- each pinned against the reference's classes;
  ddp.py       bucketed gradient all-reduce over RCCL (what create_ddp_model does for the reference).
Every function that draws random numbers in the reference takes them as an argument (or from a ``draw`` callable), so results compare
with the reference entry for entry."""
from .boxes import box_cxcywh_to_xyxy, generalized_box_iou, paired_giou_loss, paired_iou      # noqa: F401
from .matcher import HungarianMatcher, MatchWeights                                            # noqa: F401
from .dn import cdn_queries, dn_split_outputs, dn_match_indices, maskdino_dn_queries           # noqa: F401
from .criterion import DetCriterion, MaskCriterion                                             # noqa: F401
from .weights import maskdino_loss_plan, weighted_merge                                        # noqa: F401
from .targets import prepare_targets, split_things_stuff                                       # noqa: F401
from .ddp import GradientBuckets                                                               # noqa: F401
from .step import TrainStep, build_optimizer, train_iteration                                  # noqa: F401

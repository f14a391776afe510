"""dlsg_amd -- MI355X-native implementation of the D-LSG (baiyang4/D-LSG-Video-Caption) per-clip
encoder/decoder hot path: CapGnnModel forward / train step / inference on hand-written gfx950 kernels."""
from .config import make_args, make_vocab, msvd_shaped, msrvtt_shaped, apply_dataset_overrides, Vocabulary  # noqa: F401
from .model import CapGnnModel, CapBaseline1, CapBaselineModel, Trainer, GreedyGraph, BeamGraph, ss_epsilon, multistep_lr  # noqa: F401
from .gan import DiscV2, GanTrainer, GANLambdaHandler, save_checkpoint, load_checkpoint  # noqa: F401
from .data import H5File, CaptionSet, ResidentFeatures, StreamedFeatures, TrainLoader, EvalLoader, distributed_indices  # noqa: F401
from .scoring import CaptionScorer, convert_data_to_coco_scorer_format, convert_prediction, evaluate  # noqa: F401

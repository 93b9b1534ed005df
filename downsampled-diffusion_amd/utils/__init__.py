"""Drop-in ``utils`` package (names of reference utils/__init__.py:1-13 that the hot path and CLI use).
The TF-Inception FID evaluator (utils/evaluator.py) is out of scope."""
from .data import DATASETS, get_color_channels, get_dataloader
from .cli_args import get_args
from .utils import (flat_bits, get_model_state_dict, load_checkpoint_file, min_max_norm_batch, min_max_norm_image, modify_config,
                    reduce_mean, reduce_sum)
from .rnd_seed import seed_everything
from .eval_helpers import OutputStage, fix_samples, merge_rank_shards
from .paths import (CHECKPOINT_DIR, DATA_DIR, LOGGING_DIR, REFERENCE_DIR, SAMPLE_DIR, SAMPLE_LATENT_DIR, WORK_DIR)

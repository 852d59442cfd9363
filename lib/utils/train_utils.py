"""reference lib/utils/train_utils.py — only the checkpoint-ingest part is on the
tracking path (`load_pretrain`, `remove_prefix`, `check_keys`); schedulers, loggers and
checkpoint writers belong to training and are out of scope."""
from usot_amd.io_utils import check_keys, load_pretrain, remove_prefix  # noqa: F401

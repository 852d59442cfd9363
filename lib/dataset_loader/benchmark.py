"""reference lib/dataset_loader/benchmark.py: `load_dataset`."""
from usot_amd.benchmarks import load_dataset  # noqa: F401

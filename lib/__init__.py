"""Drop-in module paths of VISION-SJTU/USOT (`lib.models.models`, `lib.tracker.usot_tracker`,
`lib.utils.*`, `lib.dataset_loader.benchmark`) re-exported from the MI355X-native
implementation in `usot_amd`, so that the reference's scripts/test_usot.py runs on this
repo unchanged."""

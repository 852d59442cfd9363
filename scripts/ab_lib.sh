#!/bin/bash
# Same-box A/B of two builds of the library on the tracked frame: alternates processes (USOT_HIP_LIB), three rounds.
#   scripts/ab_lib.sh usot_amd/csrc/alt/lib_A.so [usot_amd/csrc/libusot_hip.so]
A=$1; B=${2:-usot_amd/csrc/libusot_hip.so}
for r in 1 2 3; do
  for L in $A $B; do
    USOT_HIP_LIB=$PWD/$L python - <<PY
import sys, time, torch
sys.path.insert(0, '.')
import bench
dev = torch.device('cuda:0')
model, _ = bench.build_model(0, 1, dev)
sess, crops, p = bench.open_stream(model, dev, seed=0)
conf = bench.Confidences()
bench.run_frames(sess, crops, p, conf, 200)
torch.cuda.synchronize(); t0 = time.perf_counter()
bench.run_frames(sess, crops, p, conf, 2000)
torch.cuda.synchronize()
print('%-44s loop %.1f us/frame' % ('$L', (time.perf_counter() - t0) / 2000 * 1e6), flush=True)
PY
  done
done

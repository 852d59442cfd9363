mkdir -p gpurun_out/s16
T="55 91 92 93 99 100 97 98 101 94 53"
for shape in conv2 tower conf; do
  echo "== $shape"
  for lib in usot_amd/csrc/libusot_hip.so usot_amd/csrc/alt/lib_NOMMA.so usot_amd/csrc/alt/lib_NOREAD.so usot_amd/csrc/alt/lib_NOLOAD.so usot_amd/csrc/alt/lib_NOSTORE.so usot_amd/csrc/alt/lib_NOBARRIER.so; do
    SHAPE=$shape timeout 120 python scripts/ablate_kstep.py $lib $T 2>&1 | tail -1
  done
  SHAPE=$shape timeout 120 python scripts/ablate_kstep.py usot_amd/csrc/libusot_hip.so 55:2 91:2 92:2 93:2 99:2 97:2 97:4 94:2 91:3 91:4 2>&1 | tail -1
done

# same-box A/B of two builds of libfgvc_hip.so (build/ab/libfgvc_hip_old.so against the in-tree one), interleaved
for r in 1 2; do
  for lib in build/ab/libfgvc_hip_old.so fgvc_amd/lib/libfgvc_hip.so; do
    echo "== $lib"
    FGVC_HIP_LIB=$PWD/$lib python tools/experiments/time_conv64.py 2>&1 | grep "N=8\|N=4 in + split out\|N=4 in + f32 residual + split out + f32"
  done
done
